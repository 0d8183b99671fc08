#!/bin/bash
# rates_variants.sh "v1 v2" -- stage times (one chunk alone) at every sample rate for the default library and the named variants
O=$PWD/gpurun_out/${OUT:-rates_variants.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "sync", round(s["sync"],2), "header", round(s["header"],2), "demod", round(s["demod"],2), "ts", round(s["theilsen"],1), "polar", round(s["polar"],1), "fer", d["fer"], "ok", d["frames_ok"])'
V=$PWD/modem_amd/lib/variants
for lib in default $1; do
	L=$V/libofdmrx_$lib.so; [ $lib = default ] && L=$PWD/modem_amd/lib/libofdmrx.so
	for rate in ${RATES:-8000 16000 44100 48000}; do
		n=4096; [ $rate -le 16000 ] && n=8192
		echo -n "$lib $rate one chunk ($n) alone: " >> $O
		MODEM_AMD_LIB=$L OFDMRX_NO_OVERLAP=1 timeout 600 python3 bench.py --rate $rate --frames $n --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
	done
done
cat $O
