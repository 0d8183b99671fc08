#!/bin/bash
# polar_pmc.sh [variant ...] -- issue counters of k_polar (one 8192-frame chunk, kernels back to back) for the default library and variants
R=$PWD; O=$R/gpurun_out/${OUT:-polar_pmc.txt}; mkdir -p $R/gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
cd /tmp; export TMPDIR=/tmp
export OFDMRX_NO_OVERLAP=1
for v in default "$@"; do
	L=$R/modem_amd/lib/variants/libofdmrx_$v.so; [ $v = default ] && L=$R/modem_amd/lib/libofdmrx.so
	export MODEM_AMD_LIB=$L
	echo "== $v" >> $O
	for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_INSTS SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
		d=/tmp/pmc_${v}_$(echo $c | tr ' ' '_')
		rocprofv3 --pmc $c -d $d -o x -- python3 $R/bench.py --frames 8192 --steps 1 --warmup 0 --cpu-frames 0 --host-frames 0 > /dev/null 2>&1
		db=$(find $d -name "*.db" | head -1)
		python3 $R/tools/pmc_kernel.py $db k_polar >> $O 2>&1
	done
done
cat $O
