#!/bin/bash
# exp_polar_prof.sh -- phase breakdown of k_polar (POLAR_PROF build) at different resident decoders per CU
O=gpurun_out/polar_prof.txt; mkdir -p gpurun_out; : > $O
export MODEM_AMD_LIB=$PWD/modem_amd/lib/variants/libofdmrx_prof.so OFDMRX_NO_OVERLAP=1 OFDMRX_POLAR_FORCE_GRID=1
for w in ${WPCS:-1 3 8 13 16}; do
	echo "== wpc $w" >> $O
	OFDMRX_POLAR_WPC=$w python3 bench.py --frames 8192 --steps 1 --warmup 0 --cpu-frames 0 2>&1 | grep -E "POLAR_PROF|value" | cut -c1-600 >> $O
done
cat $O
