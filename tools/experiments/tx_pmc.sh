#!/bin/bash
# tx_pmc.sh -- the transmitter alone: time per 8192 frames and the issue / LDS counters of its kernels
R=$PWD; O=$R/gpurun_out/${OUT:-tx_pmc.txt}; mkdir -p $R/gpurun_out; : > $O
python3 tools/tx_probe.py 8192 2>&1 | grep tx_encode >> $O
cd /tmp; export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE"; do
	d=/tmp/pmct_$(echo $c | tr ' ' '_'); rm -rf $d
	rocprofv3 --pmc $c -d $d -o x -- python3 $R/tools/tx_probe.py 8192 > /dev/null 2>&1
	python3 $R/tools/pmc_kernel.py $(find $d -name "*.db" | head -1) k_tx >> $O 2>&1
done
cat $O
