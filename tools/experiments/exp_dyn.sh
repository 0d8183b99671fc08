#!/bin/bash
O=$PWD/gpurun_out/dyn.txt; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "polar or waterfall or list_size or awgn or mixed_mode or chunk_pipeline or failure" 2>&1 | tail -4 >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "ts", round(s["theilsen"],1), "polar", round(s["polar"],1), "fer", d["fer"], "ok", d["frames_ok"])'
for w in 12 13 14 15; do
	for rep in 1 2; do
	echo -n "dynamic codewords, wpc $w: " >> $O
	OFDMRX_POLAR_WPC=$w timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
	done
done
cat $O
