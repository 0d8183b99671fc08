#!/bin/bash
# ab_bench.sh -- GPU suite, then the bench line with kernels back to back and with the default overlapped schedule
O=gpurun_out/ab.txt; mkdir -p gpurun_out; : > $O
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3 >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "polar", round(s["polar"],1), "ts", round(s["theilsen"],1), "fer", d["fer"], "ok", d["frames_ok"])'
OFDMRX_NO_OVERLAP=1 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 2>/dev/null | python3 -c "$pick" >> $O 2>&1
python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 2>/dev/null | python3 -c "$pick" >> $O 2>&1
cat $O
