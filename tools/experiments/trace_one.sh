#!/bin/bash
# kernel trace of one configuration, kernels back to back: ARGS="--mode 6" bash tools/experiments/trace_one.sh
R=$PWD; O=$R/gpurun_out/${OUT:-trace_one.txt}; mkdir -p $R/gpurun_out; : > $O
cd /tmp; export TMPDIR=/tmp; export OFDMRX_NO_OVERLAP=${NO_OVERLAP:-1}
rm -rf /tmp/prof_t
rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o trace -- python3 $R/bench.py --frames ${FRAMES:-8192} --steps 3 --warmup 1 --cpu-frames 0 --host-frames 0 --scl-steps 0 $ARGS > /dev/null 2>&1
python3 $R/profiles/summarize.py $(find /tmp/prof_t -name "*.db" | head -1) >> $O 2>&1
python3 $R/tools/timeline.py $(find /tmp/prof_t -name "*.db" | head -1) ${NLAST:-26} >> $O 2>&1
cat $O
