#!/bin/bash
# ts_cols.sh -- Theil-Sen alone on 51 200 synthetic rows per row length: time (library build) and the search counters (probe build)
O=$PWD/gpurun_out/${OUT:-ts_cols.txt}; mkdir -p gpurun_out; : > $O
cd tools
F="-w -O3 -std=c++17 --offload-arch=gfx950 -I../modem_amd/csrc -mllvm -disable-machine-licm"
hipcc $F -DVARIANT='"time"' -DNO_COUNTERS ts_probe.cpp -o /tmp/tsp_time
hipcc $F -DVARIANT='"counters"' $CNTFLAGS ts_probe.cpp -o /tmp/tsp_cnt
for c in ${COLS:-256 360 384 400 432 512}; do
	for sg in ${SIGMAS:-0.1}; do
		timeout 120 /tmp/tsp_time $c $sg | tail -1 >> $O
		timeout 120 /tmp/tsp_cnt $c $sg | tail -2 >> $O
	done
done
cat $O
