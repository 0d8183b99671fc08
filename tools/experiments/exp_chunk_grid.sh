#!/bin/bash
# exp_chunk_grid.sh -- chunk size as a multiple of the resident polar grid (wpc x 256 CUs): whole rounds of codewords per decoder
O=$PWD/gpurun_out/chunk_grid.txt; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "ts", round(s["theilsen"],1), "polar", round(s["polar"],1), "fer", d["fer"], "ok", d["frames_ok"])'
for cfg in "13 8192" "13 6656" "13 9984" "13 13312" "12 6144" "12 9216" "14 7168" "14 10752" "16 8192" "16 12288" "11 8448" "11 5632"; do
	set -- $cfg
	echo -n "wpc $1 chunk $2: " >> $O
	OFDMRX_POLAR_WPC=$1 timeout 300 python3 bench.py --chunk $2 --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
V=$PWD/modem_amd/lib/variants
for L in 10 11 12 13; do
	echo -n "nt level >= $L, wpc 13 chunk 8192: " >> $O
	MODEM_AMD_LIB=$V/libofdmrx_ntl$L.so timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
	echo -n "nt level >= $L, alone wpc 16: " >> $O
	MODEM_AMD_LIB=$V/libofdmrx_ntl$L.so OFDMRX_NO_OVERLAP=1 OFDMRX_POLAR_FORCE_GRID=1 OFDMRX_POLAR_WPC=16 timeout 300 python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
cat $O
