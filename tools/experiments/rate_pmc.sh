#!/bin/bash
# rate_pmc.sh RATE CHANNELS -- instruction counters of every kernel of one 8192-frame chunk at a sample rate (kernels back to back)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; RATE=${1:-48000}; CH=${2:-2}
export OFDMRX_NO_OVERLAP=1
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
	d=/tmp/rp_$(echo $c | tr ' ' '_'); rm -rf $d
	rocprofv3 --pmc $c -d $d -o x -- python3 $R/bench.py --rate $RATE --channels $CH --frames 8192 --steps 1 --warmup 0 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 > /dev/null 2>&1
	python3 $R/tools/pmc_kernel.py $(find $d -name "*.db" | head -1) rx:: 2>&1 | grep -v "k_tx\|k_awgn\|k_queue\|k_sc\|k_polar\|k_finish\|k_init"
done
