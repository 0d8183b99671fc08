#!/bin/bash
# ts_split.sh -- stage split of the Theil-Sen kernel by probe builds (tools/ts_probe.cpp: 51 200 synthetic rows of 432 points)
O=$PWD/gpurun_out/${OUT:-ts_split.txt}; mkdir -p gpurun_out; : > $O
cd tools
build() { hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -DVARIANT="\"$1\"" $2 ts_probe.cpp -o /tmp/tsp_$1 && /tmp/tsp_$1 | tail -1 >> $O; }
build full ''
build skip_yint '-DTS_PROBE_SKIP_YINT'
build skip_list '-DTS_PROBE_SKIP_LIST'
build skip_main '-DTS_PROBE_SKIP_MAIN -DTS_PROBE_NO_FALLBACK'
build skip_main_list_yint '-DTS_PROBE_SKIP_MAIN -DTS_PROBE_SKIP_LIST -DTS_PROBE_SKIP_YINT'
build no_linear_hist '-DTS_NO_LINEAR_HIST'
cd ..
cat $O
