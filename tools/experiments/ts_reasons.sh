#!/bin/bash
# ts_reasons.sh -- how often real rows leave the rank search for the slow path (variant tsreason: PERFILE_k_theilsen="-mllvm -disable-machine-licm -DTS_REASON_LOG")
O=$PWD/gpurun_out/${OUT:-ts_reasons.txt}; mkdir -p gpurun_out; : > $O
export MODEM_AMD_LIB=$PWD/modem_amd/lib/variants/libofdmrx_tsreason.so
run() { echo "[$*] 8192 frames" >> $O; timeout 300 python3 bench.py --frames 8192 --steps 1 --warmup 0 --cpu-frames 0 --host-frames 0 --scl-steps 0 "$@" 2>&1 | grep "theil-sen so far" | tail -1 >> $O; }
run
run --noise-db -20
run --noise-db -14
run --impair
run --mode 7
run --mode 9
run --mode 13
run --mode 13 --noise-db -20
run --rate 48000
cat $O
