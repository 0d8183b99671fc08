#!/bin/bash
# ts_trig.sh -- accuracy of the Theil-Sen kernel's own trig routines against double precision, then the tests and stage times
O=$PWD/gpurun_out/${OUT:-ts_trig.txt}; mkdir -p gpurun_out; : > $O
( cd tools && hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -I../modem_amd/csrc $TRIGFLAGS ts_trig_check.cpp -o /tmp/ts_trig_check && /tmp/ts_trig_check >> $O 2>&1 )
cat $O
