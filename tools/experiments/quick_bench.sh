#!/bin/bash
# quick_bench.sh [pytest -k expr] -- a parity subset, then the headline twice with the per-kernel times
O=$PWD/gpurun_out/${OUT:-quick_bench.txt}; : > $O
[ -n "$1" ] && timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$1" 2>&1 | tail -3 >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_launch_alone"]; print("value", round(d["value"]), "kernel_only", round(d["value_kernel_only"]), "alone:", {k: round(v,3) for k,v in s.items()})'
for x in "" ${EXTRA}; do
for i in 1 2; do echo -n "bench $x: " >> $O; timeout 300 python3 bench.py --steps 6 --warmup 2 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 $x 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1; done
done
cat $O
