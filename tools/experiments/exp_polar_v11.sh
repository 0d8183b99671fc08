#!/bin/bash
# exp_polar_v11.sh -- correctness of the rewritten k_polar (polar tests + waterfall sweep), then its phase breakdown and speed
O=gpurun_out/polar_v11.txt; mkdir -p gpurun_out; : > $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "polar or waterfall or list_size or awgn or mixed_mode" 2>&1 | tail -15 >> $O
export OFDMRX_NO_OVERLAP=1 OFDMRX_POLAR_FORCE_GRID=1
for w in ${WPCS:-3 13 16}; do
	echo "== wpc $w" >> $O
	MODEM_AMD_LIB=$PWD/modem_amd/lib/variants/libofdmrx_prof.so OFDMRX_POLAR_WPC=$w timeout 300 python3 bench.py --frames 8192 --steps 1 --warmup 0 --cpu-frames 0 --host-frames 0 2>&1 | grep -E "POLAR_PROF|Error|error" | cut -c1-400 >> $O
done
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "kernel_only", round(d["value_kernel_only"]), "polar", round(s["polar"],1), "ts", round(s["theilsen"],1), "fer", d["fer"], "ok", d["frames_ok"])'
for w in 10 13 16; do
	echo -n "alone wpc $w: " >> $O
	OFDMRX_POLAR_WPC=$w timeout 300 python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
unset OFDMRX_NO_OVERLAP OFDMRX_POLAR_FORCE_GRID
echo -n "overlapped default: " >> $O
timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
cat $O
