#!/bin/bash
# wpc_sweep.sh -- the bench line for different numbers of resident polar decoders per CU under the overlapped schedule
O=gpurun_out/wpc.txt; mkdir -p gpurun_out; : > $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "polar", round(s["polar"],1), "ts", round(s["theilsen"],1), "fer", d["fer"])'
for w in ${WPCS:-8 10 12 14 16}; do
	echo -n "wpc $w: " >> $O
	OFDMRX_POLAR_WPC=$w python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 2>/dev/null | python3 -c "$pick" >> $O 2>&1
done
cat $O
