#!/bin/bash
# ts_pmc.sh -- instruction counters of the Theil-Sen kernel alone (tools/ts_probe.cpp, 51 200 synthetic rows x 2 launches)
# usage: ts_pmc.sh name source [flags]
R=$PWD; O=$R/gpurun_out/ts_pmc_$1.txt; mkdir -p $R/gpurun_out; : > $O
( cd tools && hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -I../modem_amd/csrc -DVARIANT="\"$1\"" -DTS_SRC="\"$2\"" $3 ts_probe.cpp -o /tmp/tsp_$1 ) || exit 1
/tmp/tsp_$1 >> $O
cd /tmp; export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS SQ_WAIT_INST_ANY" "SQ_WAVES SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"; do
	d=/tmp/pmc_$1_$(echo $c | tr ' ' '_')
	rocprofv3 --pmc $c -d $d -o x -- /tmp/tsp_$1 > /dev/null 2>&1
	db=$(find $d -name "*.db" | head -1)
	python3 $R/tools/pmc_kernel.py $db k_theil_sen >> $O 2>&1
done
cat $O
