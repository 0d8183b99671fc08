#!/bin/bash
# variants.sh "name1 name2 ..." -- bench (one chunk alone + overlapped) of modem_amd/lib/variants/libofdmrx_<name>.so beside the default
# library, then the parity subset of the GPU suite for each variant.  OUT=<file under gpurun_out/>, K=<pytest -k expression>
O=$PWD/gpurun_out/${OUT:-variants.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "sync", round(s["sync"],2), "header", round(s["header"],2), "demod", round(s["demod"],2), "ts", round(s["theilsen"],1), "polar", round(s["polar"],1), "llr", round(s.get("llr",0),2), "finish", round(s.get("finish",0),2), "fer", d["fer"], "ok", d["frames_ok"])'
V=$PWD/modem_amd/lib/variants
for lib in default $1; do
	L=$V/libofdmrx_$lib.so; [ $lib = default ] && L=$PWD/modem_amd/lib/libofdmrx.so
	echo -n "$lib one chunk alone: " >> $O
	MODEM_AMD_LIB=$L OFDMRX_NO_OVERLAP=1 timeout 300 python3 bench.py --frames 8192 --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
	echo -n "$lib overlapped: " >> $O
	MODEM_AMD_LIB=$L timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
for lib in $1; do
	echo "$lib parity:" >> $O
	MODEM_AMD_LIB=$V/libofdmrx_$lib.so timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${K:-clean or awgn or impairment or all_modes or 8bit}" 2>&1 | tail -2 >> $O
done
cat $O
