#!/bin/bash
# pmc_lds.sh -- LDS activity / bank conflicts / waits per receive kernel for one 8192-frame chunk (kernels back to back)
R=$PWD; O=$R/gpurun_out/${OUT:-pmc_lds.txt}; mkdir -p $R/gpurun_out; : > $O
cd /tmp; export TMPDIR=/tmp; export OFDMRX_NO_OVERLAP=1
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_BUSY_CU_CYCLES SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
	d=/tmp/pmcl_$(echo $c | tr ' ' '_'); rm -rf $d
	rocprofv3 --pmc $c -d $d -o x -- python3 $R/bench.py --frames 8192 --steps 1 --warmup 0 --cpu-frames 0 --host-frames 0 --scl-steps 0 $ARGS > /dev/null 2>&1
	python3 $R/tools/pmc_kernel.py $(find $d -name "*.db" | head -1) rx:: | grep -v "k_tx\|k_awgn\|k_cert\|k_init\|k_polar\|k_finish\|_more\|k_back\|k_header\|k_theil" >> $O 2>&1
done
cat $O
