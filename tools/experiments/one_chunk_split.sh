#!/bin/bash
O=$PWD/gpurun_out/one_chunk_split.txt; : > $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); print("value", round(d["value"]), "kernel_only", round(d["value_kernel_only"]), "ms_per_step", round(d["ms_per_step"],2), "chunk", d.get("chunk_frames"))'
for c in 0 4096 2048 2731; do
  echo -n "[--frames 8192 --chunk $c] " >> $O
  python3 bench.py --frames 8192 --chunk $c --steps 10 --warmup 2 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 2>/dev/null | tail -1 | python3 -c "$pick" >> $O 2>&1
done
for c in 0 4096; do
  echo -n "[--frames 16384 --chunk $c] " >> $O
  python3 bench.py --frames 16384 --chunk $c --steps 10 --warmup 2 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 2>/dev/null | tail -1 | python3 -c "$pick" >> $O 2>&1
done
cat $O
