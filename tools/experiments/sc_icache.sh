#!/bin/bash
# sc_icache.sh -- does k_sc wait for its instructions?  SQC instruction-cache counters and SQ_IFETCH on the -20 dB run (16384 frames, kernels back to back)
R=$PWD; G=$R/gpurun_out; S=$G/${1:-r06}_sc_icache.txt; : > $S
cd /tmp; export TMPDIR=/tmp OFDMRX_NO_OVERLAP=1
BN="python3 $R/bench.py --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 --frames 16384 --steps 1 --warmup 0 --noise-db -20"
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQC_[A-Z_0-9]*ICACHE[A-Z_0-9]*|SQ_IFETCH[A-Z_]*|SQC_INST[A-Z_0-9]*" | sort -u | tr '\n' ' ' >> $S; echo >> $S
i=0
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ SQC_TC_STALL"; do
	i=$((i + 1)); d=/tmp/pmc_ic_$i
	echo "# rocprofv3 --pmc $c" >> $S
	rocprofv3 --pmc $c -d $d -o x -- $BN > /tmp/ic_$i.log 2>&1
	python3 $R/tools/pmc_kernel.py $(find $d -name "*.db" | head -1) rx:: 2>&1 | grep -E "k_sc<6>|k_theil_sen |k_back" >> $S
	grep -iE "error|invalid|not (found|supported)" /tmp/ic_$i.log | head -3 >> $S
done
cat $S
