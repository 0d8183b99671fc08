#!/bin/bash
# front_pmc.sh -- PMC passes over one 8192-frame chunk (kernels back to back): issue / LDS / wait counters of the stream-A kernels
R=$PWD; O=$R/gpurun_out/front_pmc_${RATE:-8000}.txt; mkdir -p $R/gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
cd /tmp; export TMPDIR=/tmp
export OFDMRX_NO_OVERLAP=1
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAVES"; do
	d=/tmp/pmc_$(echo $c | tr ' ' '_')
	rocprofv3 --pmc $c -d $d -o x -- python3 $R/bench.py --rate ${RATE:-8000} --frames ${NF:-8192} --steps 1 --warmup 0 --cpu-frames 0 --host-frames 0 > /dev/null 2>&1
	db=$(find $d -name "*.db" | head -1)
	for k in k_demod k_sync k_header; do
		python3 $R/tools/pmc_kernel.py $db $k >> $O 2>&1
	done
done
cat $O
