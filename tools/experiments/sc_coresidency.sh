#!/bin/bash
# sc_coresidency.sh -- can k_sc share a CU with the front stages?  Two handles on halves of a -20 dB batch (every frame goes through
# k_sc), each with OFDMRX_SC_WPC resident list-1 decoders per CU: with 10 (the default) one handle's k_sc takes every byte of LDS,
# with 4 - 7 the other handle's front kernels can be resident beside it.  GPU_MAX_HW_QUEUES: HIP maps streams onto that many
# hardware queues (default 4) and kernels of streams that share one run in order.
O=$PWD/gpurun_out/${OUT:-sc_coresidency.txt}; mkdir -p gpurun_out; : > $O
for q in ${QUEUES:-4 16}; do
for w in ${WPCS:-10 6 5}; do
	echo "== GPU_MAX_HW_QUEUES=$q OFDMRX_SC_WPC=$w" >> $O
	GPU_MAX_HW_QUEUES=$q OFDMRX_SC_WPC=$w PROBE_NOISE_DB=-20 timeout 600 python3 tools/two_handles_probe.py 2>&1 | grep handles >> $O
done
done
cat $O
