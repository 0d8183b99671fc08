#!/bin/bash
O=$PWD/gpurun_out/lanes_dbg.txt; mkdir -p gpurun_out; : > $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); print("value", round(d["value"]), "kernel_only", round(d.get("value_kernel_only") or 0))'
for rep in 1 2; do
for env in "OFDMRX_LANES=2" "OFDMRX_LANES=0" "OFDMRX_LANES=0 OFDMRX_LANES_DBG=1" "OFDMRX_LANES=0 OFDMRX_LANES_DBG=3"; do
	echo -n "[$env]: " >> $O
	env $env timeout 300 python3 bench.py --steps 6 --warmup 3 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 --noise-db -20 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
done
cat $O
