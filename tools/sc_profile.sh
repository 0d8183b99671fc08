#!/bin/bash
# sc_profile.sh TAG [noise_db] -- kernel trace + counters of the pipeline at a noise level where the list-1 pass (k_sc) decides the frames
TAG=${1:-r05_sc}; DB=${2:--20}; R=$PWD; G=$R/gpurun_out; mkdir -p $G
S=$G/${TAG}_sc_summary.txt; : > $S
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 --noise-db $DB"
echo "# $TAG: rocprofv3 --kernel-trace --stats -- $B --steps 2 --warmup 1 (OFDMRX_NO_OVERLAP=1: kernels back to back)" >> $S
OFDMRX_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_sc -o trace -- $B --steps 2 --warmup 1 > $G/${TAG}_sc_bench_under_profiler.json 2>/dev/null
python3 $R/profiles/summarize.py $(find /tmp/prof_sc -name "*.db" | head -1) >> $S 2>&1
export OFDMRX_NO_OVERLAP=1
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS SQ_WAIT_INST_ANY"; do
	d=/tmp/pmc_sc_$(echo $c | tr ' ' '_')
	rocprofv3 --pmc $c -d $d -o x -- $B --frames 8192 --steps 1 --warmup 0 > /dev/null 2>&1
	python3 $R/tools/pmc_kernel.py $(find $d -name "*.db" | head -1) k_sc >> $S 2>&1
done
