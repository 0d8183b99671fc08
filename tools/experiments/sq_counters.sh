#!/bin/bash
# sq_counters.sh TAG -- what the SQ says about the issue-bound kernels (round-5 verdict, weak 8 / item 3), next to a calibration of
# the same counters on streams whose rate is known (tools/ubench_issue.hip):
#   passes A / B on the default path at -30 dB [cert: k_theil_sen, k_demod, k_back, k_sync] and at -20 dB [sc: k_sc];
#   TCC / TCP passes on the -20 dB run (is k_sc waiting for the memory system, and for what part of it);
#   pass A on the micro-benchmark's kernels.
# One 16384-frame call per pass, kernels back to back (OFDMRX_NO_OVERLAP=1).  Text only: gpurun_out/TAG_sq_counters.txt.
TAG=${1:-r06}; R=$PWD; G=$R/gpurun_out; mkdir -p $G
S=$G/${TAG}_sq_counters.txt; : > $S
make -C modem_amd/csrc -q all && echo "# library up to date with sources" >> $S || echo "# STALE LIBRARY" >> $S
cd /tmp; export TMPDIR=/tmp
hipcc -w --offload-arch=gfx950 -O3 $R/tools/ubench_issue.hip -o /tmp/ubench_issue || echo "# ubench did not build" >> $S
echo "# tools/ubench_issue.hip (no profiler)" >> $S
/tmp/ubench_issue >> $S 2>&1
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
B="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU"
echo "# pass A on the micro-benchmark: rocprofv3 --pmc $A -- /tmp/ubench_issue" >> $S
rocprofv3 --pmc $A -d /tmp/pmc_ub -o x -- /tmp/ubench_issue > /dev/null 2>&1
python3 $R/tools/pmc_kernel.py $(find /tmp/pmc_ub -name "*.db" | head -1) "k<" | sed "s/^/[ubench] /" >> $S 2>&1
export OFDMRX_NO_OVERLAP=1
BN="python3 $R/bench.py --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 --frames 16384 --steps 1 --warmup 0"
for mode in cert sc; do
	X=""; [ $mode = sc ] && X="--noise-db -20"
	i=0
	for c in "$A" "$B" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum TCC_CYCLE_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
		i=$((i + 1))
		[ $mode = cert ] && [ $i -gt 3 ] && continue          # (the memory-side passes: the -20 dB run only)
		echo "# [$mode] rocprofv3 --pmc $c -- python3 bench.py --frames 16384 --steps 1 --warmup 0 $X (kernels back to back)" >> $S
		d=/tmp/pmc_${mode}_$i
		rocprofv3 --pmc $c -d $d -o x -- $BN $X > /dev/null 2>&1
		python3 $R/tools/pmc_kernel.py $(find $d -name "*.db" | head -1) rx:: | grep -E "k_theil_sen |k_demod|k_back|k_sync<|k_sc<|k_sc_finish|k_header" | sed "s/^/[$mode] /" >> $S 2>&1
	done
done
tail -60 $S
