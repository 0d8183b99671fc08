#!/bin/bash
# exp_front_overlap.sh -- whole front inside the polar phase: tests, then schedule / wpc sweep
O=$PWD/gpurun_out/front_overlap.txt; mkdir -p gpurun_out; : > $O
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6 >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "sync", round(s["sync"],1), "header", round(s["header"],1), "demod", round(s["demod"],1), "ts", round(s["theilsen"],1), "polar", round(s["polar"],1), "fer", d["fer"], "ok", d["frames_ok"])'
V=$PWD/modem_amd/lib/variants
echo -n "one chunk alone (PER=8): " >> $O
OFDMRX_NO_OVERLAP=1 timeout 300 python3 bench.py --frames 8192 --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
echo -n "one chunk alone (PER=16): " >> $O
MODEM_AMD_LIB=$V/libofdmrx_per16.so OFDMRX_NO_OVERLAP=1 timeout 300 python3 bench.py --frames 8192 --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
for w in 10 11 12 13; do
	echo -n "front overlapped wpc $w: " >> $O
	OFDMRX_POLAR_WPC=$w timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
for w in 11 12; do
	echo -n "front exclusive wpc $w: " >> $O
	OFDMRX_FRONT_EXCLUSIVE=1 OFDMRX_POLAR_WPC=$w timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
echo -n "front exclusive wpc 12 PER=16: " >> $O
MODEM_AMD_LIB=$V/libofdmrx_per16.so OFDMRX_FRONT_EXCLUSIVE=1 timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
cat $O
