#!/bin/bash
O=$PWD/gpurun_out/ubench_issue.txt; mkdir -p gpurun_out
hipcc -w -O3 --offload-arch=gfx950 tools/ubench_issue.hip -o /tmp/ubench_issue && /tmp/ubench_issue > $O 2>&1
cat $O
tools/experiments/exp_demod.sh
