#!/bin/bash
# shared_pmc.sh -- issue counters of the two kernels of the shared phase (one 8192-frame chunk, kernels back to back)
R=$PWD; O=$R/gpurun_out/shared_pmc.txt; mkdir -p $R/gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
cd /tmp; export TMPDIR=/tmp
export OFDMRX_NO_OVERLAP=1
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS SQ_WAIT_INST_ANY"; do
	d=/tmp/pmc_$(echo $c | tr ' ' '_')
	rocprofv3 --pmc $c -d $d -o x -- python3 $R/bench.py --frames 8192 --steps 1 --warmup 0 --cpu-frames 0 --host-frames 0 > /dev/null 2>&1
	db=$(find $d -name "*.db" | head -1)
	for k in k_polar k_theil_sen; do
		python3 $R/tools/pmc_kernel.py $db $k >> $O 2>&1
	done
done
cat $O
