#!/bin/bash
# stage times (one chunk alone) and the overlapped bench line at the other sample rates
O=$PWD/gpurun_out/${OUT:-rates.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "sync", round(s["sync"],2), "header", round(s["header"],2), "demod", round(s["demod"],2), "ts", round(s["theilsen"],1), "llr", round(s.get("llr",0),2), "polar", round(s["polar"],1), "finish", round(s.get("finish",0),2), "fer", d["fer"], "ok", d["frames_ok"])'
for rate in ${RATES:-16000 44100 48000}; do
	n=4096; [ $rate = 16000 ] && n=8192
	echo -n "$rate one chunk ($n) alone: " >> $O
	OFDMRX_NO_OVERLAP=1 timeout 600 python3 bench.py --rate $rate --frames $n --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
	echo -n "$rate overlapped (16384): " >> $O
	timeout 600 python3 bench.py --rate $rate --frames 16384 --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
cat $O
