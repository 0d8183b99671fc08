#!/bin/bash
# exp_split.sh -- front sub-batch split + nt level 10: full tests, stage times, then the other workloads (strong scaling line, configs[3], configs[4])
O=$PWD/gpurun_out/split.txt; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6 >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "sync", round(s["sync"],1), "header", round(s["header"],1), "demod", round(s["demod"],1), "ts", round(s["theilsen"],1), "polar", round(s["polar"],1), "fer", d["fer"], "ok", d["frames_ok"])'
for rep in 1 2; do
echo -n "default (split, nt>=10) overlapped: " >> $O
timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
echo -n "split off: " >> $O
OFDMRX_FRONT_SPLIT_OFF=1 timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
echo -n "nt>=9 overlapped: " >> $O
MODEM_AMD_LIB=$PWD/modem_amd/lib/variants/libofdmrx_ntl9.so timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
echo -n "one chunk alone: " >> $O
OFDMRX_NO_OVERLAP=1 timeout 300 python3 bench.py --frames 8192 --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
echo -n "strong scaling N=1: " >> $O
timeout 300 python3 bench.py --scaling strong --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
echo -n "8192 frames in one call (auto-split pipeline): " >> $O
timeout 300 python3 bench.py --frames 8192 --steps 4 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
echo -n "configs[3] --impair: " >> $O
timeout 300 python3 bench.py --impair --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
echo "configs[4] sweep driver, 3 levels x 65536 frames:" >> $O
timeout 600 python3 tools/ber_sweep.py --frames 65536 --levels -40 -30 -20 2>&1 | tail -4 >> $O
cat $O
