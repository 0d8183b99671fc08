#!/bin/bash
# polar_bw_probe.sh -- (1) FETCH_SIZE / WRITE_SIZE calibration in k_polar's access pattern, (2) k_polar alone at
# different numbers of resident decoders per CU.  Run on the GPU box from the repo root; writes gpurun_out/polar_bw.txt
R=$PWD; O=$R/gpurun_out/polar_bw.txt; mkdir -p $R/gpurun_out; : > $O
hipcc -w --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o /tmp/pmc_calib || exit 1
/tmp/pmc_calib >> $O 2>&1
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
	rocprofv3 --pmc $c -d /tmp/calib_$c -o x -- /tmp/pmc_calib > /dev/null 2>&1
	db=$(find /tmp/calib_$c -name "*.db" | head -1)
	python3 $R/tools/pmc_kernel.py $db calib >> $O 2>&1
done
cd $R
for w in 3 4 6 8 10 12 16 20; do
	echo "== resident decoders per CU: $w" >> $O
	OFDMRX_NO_OVERLAP=1 OFDMRX_POLAR_FORCE_GRID=1 OFDMRX_POLAR_WPC=$w python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('value', round(d['value']), 'polar ms/step', round(d['stage_ms_per_step']['polar'],1), 'fer', d['fer'])" >> $O 2>&1
done
cat $O
