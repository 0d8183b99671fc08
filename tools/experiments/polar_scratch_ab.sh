#!/bin/bash
# polar_scratch_ab.sh -- k_polar with 96 VGPRs (18 spilled, 76 B of scratch: the shipped build), 128 (6 spilled, 28 B) and 131 (none, three
# waves per SIMD = 12 decoders per CU): the list decoder forced for every frame, -16 dB and -30 dB, ms per 8192 codewords
O=$PWD/gpurun_out/${OUT:-polar_scratch_ab.txt}; mkdir -p gpurun_out; : > $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("frames/s", round(d["value_kernel_only"]), "k_polar ms per step", round(s["polar"],1), "fer", d["fer"])'
for db in -30 -16; do
for lib in default polar4 polar3; do
	L=$PWD/modem_amd/lib/variants/libofdmrx_$lib.so; [ $lib = default ] && L=$PWD/modem_amd/lib/libofdmrx.so
	for wpc in 16 12; do
		[ $lib = polar3 ] && [ $wpc = 16 ] && continue
		echo -n "$db dB $lib OFDMRX_POLAR_WPC=$wpc: " >> $O
		MODEM_AMD_LIB=$L OFDMRX_POLAR_WPC=$wpc OFDMRX_NO_CERT=1 OFDMRX_NO_SC=1 timeout 300 python3 bench.py --noise-db $db --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
	done
done
done
cat $O
