#!/bin/bash
O=$PWD/gpurun_out/llr_back.txt; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
timeout 1800 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5 >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "demod", round(s["demod"],1), "ts", round(s["theilsen"],1), "llr", round(s["llr"],1), "polar", round(s["polar"],1), "finish", round(s["finish"],1), "fer", d["fer"], "ok", d["frames_ok"])'
for w in 10 11 12 13; do
	for rep in 1 2; do
	echo -n "llr on the back stream, wpc $w: " >> $O
	OFDMRX_POLAR_WPC=$w timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
	done
done
cat $O
