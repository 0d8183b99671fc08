#!/bin/bash
# copy_engine_ab.sh -- how the per-chunk output copies of the device entry (pinned host outputs) travel: the runtime's default (copy kernels
# in the trace), SDMA off, blit threshold 0
O=$PWD/gpurun_out/copy_engine_ab.txt; : > $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); print("value", round(d["value"]), "kernel_only", round(d["value_kernel_only"]), "ms_per_step", round(d["ms_per_step"],2))'
for rep in 1 2; do
for cfg in "X=0" "HSA_ENABLE_SDMA=0" "GPU_FORCE_BLIT_COPY_SIZE=0" "HSA_ENABLE_SDMA=1 GPU_FORCE_BLIT_COPY_SIZE=0"; do
  echo -n "[$cfg] " >> $O
  env $cfg python3 bench.py --steps 10 --warmup 2 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done; done
cat $O
