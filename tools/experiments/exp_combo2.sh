#!/bin/bash
# exp_combo2.sh -- TS linear select + polar terminal rate-1 passes: tests, probe, wpc sweep
O=$PWD/gpurun_out/combo2.txt; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6 >> $O
cd tools
build() { hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -DVARIANT="\"$1\"" $2 ts_probe.cpp -o /tmp/tsp_$1 && /tmp/tsp_$1 | tail -1 >> $O; }
build linsel ''
build radixsel '-DTS_NO_LINEAR_SELECT'
cd ..
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "sync", round(s["sync"],1), "header", round(s["header"],1), "demod", round(s["demod"],1), "ts", round(s["theilsen"],1), "polar", round(s["polar"],1), "fer", d["fer"], "ok", d["frames_ok"])'
echo -n "one chunk alone: " >> $O
OFDMRX_NO_OVERLAP=1 timeout 300 python3 bench.py --frames 8192 --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
for w in 12 13 14 15 16; do
	echo -n "wpc $w: " >> $O
	OFDMRX_POLAR_WPC=$w timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
cat $O
