#!/bin/bash
O=$PWD/gpurun_out/${OUT:-quick.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${K:-theil or osd or clean or awgn or failure or all_modes or other_rates_decode}" 2>&1 | tail -4 >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "sync", round(s["sync"],2), "header", round(s["header"],2), "demod", round(s["demod"],2), "ts", round(s["theilsen"],1), "polar", round(s["polar"],1), "fer", d["fer"], "ok", d["frames_ok"])'
echo -n "one chunk alone: " >> $O
OFDMRX_NO_OVERLAP=1 timeout 300 python3 bench.py --frames 8192 --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
for rep in 1 2; do
echo -n "overlapped: " >> $O
timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
cat $O
