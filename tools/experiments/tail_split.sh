#!/bin/bash
# tail_split.sh -- the headline with the call's last chunk as two halves (default) against one chunk (OFDMRX_NO_TAIL_SPLIT=1), alternating
O=$PWD/gpurun_out/tail_split.txt; : > $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); print("value", round(d["value"]), "kernel_only", round(d["value_kernel_only"]), "fer", d["fer"])'
for rep in 1 2 3; do for e in "" "OFDMRX_NO_TAIL_SPLIT=1"; do
	echo -n "${e:-split}: " >> $O
	env $e timeout 300 python3 bench.py --steps 6 --warmup 2 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done; done
cat $O
