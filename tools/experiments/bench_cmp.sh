#!/bin/bash
# bench_cmp.sh -- the bench line with the default flags, without the CPU / host legs and with the suite flags, twice each on one box (box-to-box spread of `value`: 1.71 - 1.77 M)
O=$PWD/gpurun_out/bench_cmp.txt; : > $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); print("value", round(d["value"]), "kernel_only", round(d["value_kernel_only"]), "ms_per_step", round(d["ms_per_step"],2), "steps", d["steps"], "warmup", d["warmup"])'
for rep in 1 2; do
echo -n "[default flags] " >> $O; python3 bench.py 2>/dev/null | tail -1 | python3 -c "$pick" >> $O
echo -n "[--cpu-frames 0 --host-frames 0] " >> $O; python3 bench.py --cpu-frames 0 --host-frames 0 2>/dev/null | tail -1 | python3 -c "$pick" >> $O
echo -n "[--steps 10 --warmup 2 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0] " >> $O; python3 bench.py --steps 10 --warmup 2 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 2>/dev/null | tail -1 | python3 -c "$pick" >> $O
done
cat $O
