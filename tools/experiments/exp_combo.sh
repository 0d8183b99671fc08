#!/bin/bash
# exp_combo.sh -- full GPU suite on the current build, then schedule (front inside / outside the polar phase) x wpc, TS specialisation
O=$PWD/gpurun_out/combo.txt; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6 >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "sync", round(s["sync"],1), "header", round(s["header"],1), "demod", round(s["demod"],1), "ts", round(s["theilsen"],1), "polar", round(s["polar"],1), "fer", d["fer"], "ok", d["frames_ok"])'
V=$PWD/modem_amd/lib/variants
for lib in default per16 ts432; do
	L=$V/libofdmrx_$lib.so; [ $lib = default ] && L=$PWD/modem_amd/lib/libofdmrx.so
	echo -n "$lib one chunk alone: " >> $O
	MODEM_AMD_LIB=$L OFDMRX_NO_OVERLAP=1 timeout 300 python3 bench.py --frames 8192 --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
for w in 10 11 12 13 14; do
	echo -n "front overlapped wpc $w: " >> $O
	OFDMRX_POLAR_WPC=$w timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
for w in 11 12 13; do
	echo -n "front exclusive wpc $w: " >> $O
	OFDMRX_FRONT_EXCLUSIVE=1 OFDMRX_POLAR_WPC=$w timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
echo -n "ts432 front overlapped wpc 12: " >> $O
MODEM_AMD_LIB=$V/libofdmrx_ts432.so timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
cat $O
