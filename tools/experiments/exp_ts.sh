#!/bin/bash
# exp_ts.sh -- Theil-Sen: bit-exact test, stage split of the kernel by probe builds (tools/ts_probe.cpp), old vs new classification
O=$PWD/gpurun_out/ts.txt; mkdir -p gpurun_out; : > $O
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "theil or awgn or impairment or all_modes" 2>&1 | tail -5 >> $O
cd tools
build() { # name flags
	hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -DVARIANT="\"$1\"" $2 ts_probe.cpp -o /tmp/tsp_$1 && /tmp/tsp_$1 | tail -1 >> $O
}
build r1_u16 '-DTS_SRC="_k_demod_r1.hip"'
build new_u8 ''
build new_u16_w4 '-DTS_U=16 -DTS_WAVES=4'
build new_u12 '-DTS_U=12'
build new_u4 '-DTS_U=4'
build new_skip_yint '-DTS_PROBE_SKIP_YINT'
build new_skip_list '-DTS_PROBE_SKIP_LIST'
build new_skip_main '-DTS_PROBE_SKIP_MAIN -DTS_PROBE_NO_FALLBACK'
build new_skip_main_list_yint '-DTS_PROBE_SKIP_MAIN -DTS_PROBE_SKIP_LIST -DTS_PROBE_SKIP_YINT'
cd ..
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "kernel_only", round(d["value_kernel_only"]), "polar", round(s["polar"],1), "ts", round(s["theilsen"],1), "fer", d["fer"], "ok", d["frames_ok"])'
echo -n "bench no-overlap: " >> $O
OFDMRX_NO_OVERLAP=1 timeout 300 python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
cat $O
