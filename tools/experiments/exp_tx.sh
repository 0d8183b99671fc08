#!/bin/bash
# exp_tx.sh -- transmitter with pre-formed rows + asymmetric one-chunk split: full tests, then timings
O=$PWD/gpurun_out/tx.txt; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6 >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "gen_s", round(d["input_generation_s"],2), "sync", round(s["sync"],1), "header", round(s["header"],1), "demod", round(s["demod"],1), "ts", round(s["theilsen"],1), "polar", round(s["polar"],1), "fer", d["fer"], "ok", d["frames_ok"])'
echo -n "default: " >> $O
timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
for n in 8192 4096 2048 16384; do
echo -n "$n frames in one call: " >> $O
timeout 300 python3 bench.py --frames $n --steps 4 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
echo -n "8192 frames, no overlap: " >> $O
OFDMRX_NO_OVERLAP=1 timeout 300 python3 bench.py --frames 8192 --steps 4 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
echo "configs[4] sweep driver, 4 levels x 65536 frames:" >> $O
timeout 600 python3 tools/ber_sweep.py --frames 65536 --levels -40 -35 -30 -20 2>&1 | tail -5 >> $O
cat $O
