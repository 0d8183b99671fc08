#!/bin/bash
# exp_ts3.sh -- Theil-Sen with sign-history classification: bit-exact tests, stage split by probe builds, bench
O=$PWD/gpurun_out/ts3.txt; mkdir -p gpurun_out; : > $O
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "theil or awgn or impairment or all_modes or waterfall" 2>&1 | tail -5 >> $O
cd tools
build() { hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -DVARIANT="\"$1\"" $2 ts_probe.cpp -o /tmp/tsp_$1 && /tmp/tsp_$1 | tail -1 >> $O; }
build hist32 ''
build skip_yint '-DTS_PROBE_SKIP_YINT'
build skip_list '-DTS_PROBE_SKIP_LIST'
build skip_main '-DTS_PROBE_SKIP_MAIN -DTS_PROBE_NO_FALLBACK'
build skip_main_list_yint '-DTS_PROBE_SKIP_MAIN -DTS_PROBE_SKIP_LIST -DTS_PROBE_SKIP_YINT'
build w4 '-DTS_WAVES=4'
cd ..
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "kernel_only", round(d["value_kernel_only"]), "polar", round(s["polar"],1), "ts", round(s["theilsen"],1), "fer", d["fer"], "ok", d["frames_ok"])'
echo -n "bench no-overlap: " >> $O
OFDMRX_NO_OVERLAP=1 timeout 300 python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
for w in 11 12 13; do
echo -n "bench overlapped wpc $w: " >> $O
OFDMRX_POLAR_WPC=$w timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
cat $O
