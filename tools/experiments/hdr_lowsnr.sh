#!/bin/bash
# header occupancy variants at the headline noise level and near the waterfall (certificate fails, order-3 search runs)
O=$PWD/gpurun_out/hdr_lowsnr.txt; mkdir -p gpurun_out; : > $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "sync", round(s["sync"],2), "header", round(s["header"],2), "demod", round(s["demod"],2), "ts", round(s["theilsen"],1), "polar", round(s["polar"],1), "fer", d["fer"], "ok", d["frames_ok"])'
V=$PWD/modem_amd/lib/variants
for lib in default $1; do
	L=$V/libofdmrx_$lib.so; [ $lib = default ] && L=$PWD/modem_amd/lib/libofdmrx.so
	for nz in -30 -16 -14.5; do
		echo -n "$lib noise $nz one chunk alone: " >> $O
		MODEM_AMD_LIB=$L OFDMRX_NO_OVERLAP=1 timeout 300 python3 bench.py --frames 8192 --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 --noise-db $nz 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
	done
	echo -n "$lib overlapped: " >> $O
	MODEM_AMD_LIB=$L timeout 300 python3 bench.py --steps 3 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
cat $O
