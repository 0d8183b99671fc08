#!/bin/bash
# polar_pmc.sh -- PMC passes over one 8192-frame chunk (kernels back to back): traffic and instruction mix of k_polar
R=$PWD; O=$R/gpurun_out/polar_pmc.txt; mkdir -p $R/gpurun_out; : > $O
cd /tmp; export TMPDIR=/tmp
export OFDMRX_NO_OVERLAP=1
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
	d=/tmp/pmc_$(echo $c | tr ' ' '_')
	rocprofv3 --pmc $c -d $d -o x -- python3 $R/bench.py --frames 8192 --steps 1 --warmup 0 --cpu-frames 0 > /dev/null 2>&1
	db=$(find $d -name "*.db" | head -1)
	python3 $R/tools/pmc_kernel.py $db k_polar >> $O 2>&1
done
cat $O
