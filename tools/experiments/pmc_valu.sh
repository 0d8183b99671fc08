#!/bin/bash
# pmc_valu.sh -- VALU / SALU / LDS instruction counts per receive kernel for one 8192-frame chunk (kernels back to back)
R=$PWD; O=$R/gpurun_out/${OUT:-pmc_valu.txt}; mkdir -p $R/gpurun_out; : > $O
cd /tmp; export TMPDIR=/tmp; export OFDMRX_NO_OVERLAP=1
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
	d=/tmp/pmcv_$(echo $c | tr ' ' '_'); rm -rf $d
	rocprofv3 --pmc $c -d $d -o x -- python3 $R/bench.py --frames 8192 --steps 1 --warmup 0 --cpu-frames 0 --host-frames 0 --scl-steps 0 $ARGS > /dev/null 2>&1
	python3 $R/tools/pmc_kernel.py $(find $d -name "*.db" | head -1) rx:: | grep -v "k_tx\|k_awgn" >> $O 2>&1
done
cat $O
