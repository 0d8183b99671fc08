#!/bin/bash
# exp_polar_v11b.sh -- full GPU suite on the rewritten k_polar, then variants: table prefetch (default), <= 96 VGPRs, nt stores
O=gpurun_out/polar_v11b.txt; mkdir -p gpurun_out; : > $O
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5 >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "kernel_only", round(d["value_kernel_only"]), "polar", round(s["polar"],1), "ts", round(s["theilsen"],1), "fer", d["fer"], "ok", d["frames_ok"])'
V=$PWD/modem_amd/lib/variants
echo "== phase profile, wpc 16" >> $O
MODEM_AMD_LIB=$V/libofdmrx_prof.so OFDMRX_NO_OVERLAP=1 OFDMRX_POLAR_FORCE_GRID=1 OFDMRX_POLAR_WPC=16 timeout 300 python3 bench.py --frames 8192 --steps 1 --warmup 0 --cpu-frames 0 --host-frames 0 2>&1 | grep -E "POLAR_PROF" | head -1 | cut -c1-300 >> $O
for lib in default lb5 nt; do
	L=$V/libofdmrx_$lib.so; [ $lib = default ] && L=$PWD/modem_amd/lib/libofdmrx.so
	for w in 13 16 20; do
		echo -n "$lib alone wpc $w: " >> $O
		MODEM_AMD_LIB=$L OFDMRX_NO_OVERLAP=1 OFDMRX_POLAR_FORCE_GRID=1 OFDMRX_POLAR_WPC=$w timeout 300 python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
	done
	for w in 8 12 13; do
		echo -n "$lib overlapped wpc $w: " >> $O
		MODEM_AMD_LIB=$L OFDMRX_POLAR_WPC=$w timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
	done
done
cat $O
