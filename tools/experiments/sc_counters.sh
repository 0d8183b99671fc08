#!/bin/bash
# sc_counters.sh TAG -- k_sc under the SQ / TCC counters at -20 dB and on the configs[3] chain (one 16384-frame call per pass, kernels back to back)
TAG=${1:-r06b}; R=$PWD; G=$R/gpurun_out; mkdir -p $G
S=$G/${TAG}_sc_counters.txt; : > $S
cd /tmp; export TMPDIR=/tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
B="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"
export OFDMRX_NO_OVERLAP=1 OFDMRX_NO_TAIL_SPLIT=1
BN="python3 $R/bench.py --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 --frames 16384 --steps 1 --warmup 0"
for mode in m20 chain; do
	X="--noise-db -20"; [ $mode = chain ] && X="--impair"
	i=0
	for c in "$A" "$B" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum"; do
		i=$((i + 1))
		echo "# [$mode] rocprofv3 --pmc $c -- python3 bench.py --frames 16384 --steps 1 --warmup 0 $X (kernels back to back)" >> $S
		d=/tmp/pmc_${mode}_$i
		rocprofv3 --pmc $c -d $d -o x -- $BN $X > /dev/null 2>&1
		python3 $R/tools/pmc_kernel.py $(find $d -name "*.db" | head -1) rx:: | grep -E "k_back|k_sc<|k_sc_finish" | sed "s/^/[$mode] /" >> $S 2>&1
	done
done
cat $S
