#!/bin/bash
# exp_api_nt.sh -- new API tests, then cache-policy variants of k_polar at 96 VGPRs (alone and overlapped)
O=gpurun_out/api_nt.txt; mkdir -p gpurun_out; : > $O
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "host_pointer or one_chunk or argument or cli or chunk_pipeline or device_pointer or skip" 2>&1 | tail -15 >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "kernel_only", round(d["value_kernel_only"]), "host", d["value_host"] and round(d["value_host"]), "polar", round(s["polar"],1), "ts", round(s["theilsen"],1), "fer", d["fer"], "ok", d["frames_ok"])'
V=$PWD/modem_amd/lib/variants
for lib in lb5 lb5nt1 lb5nt2 lb5nt3 lb5nt4; do
	L=$V/libofdmrx_$lib.so
	echo -n "$lib alone wpc 16: " >> $O
	MODEM_AMD_LIB=$L OFDMRX_NO_OVERLAP=1 OFDMRX_POLAR_FORCE_GRID=1 OFDMRX_POLAR_WPC=16 timeout 300 python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
	for w in 11 12 14; do
		echo -n "$lib overlapped wpc $w: " >> $O
		MODEM_AMD_LIB=$L OFDMRX_POLAR_WPC=$w timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
	done
done
echo "== full default line (host leg + cpu baseline)" >> $O
MODEM_AMD_LIB=$V/libofdmrx_lb5nt1.so OFDMRX_POLAR_WPC=12 timeout 600 python3 bench.py 2>&1 | tail -1 >> $O
cat $O
