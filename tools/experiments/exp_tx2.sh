#!/bin/bash
# exp_tx2.sh -- asynchronous transmitter with cached scratch: TX / split tests, the sweep driver, bench
O=$PWD/gpurun_out/tx2.txt; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6 >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "gen_s", round(d["input_generation_s"],2), "ts", round(s["theilsen"],1), "polar", round(s["polar"],1), "fer", d["fer"], "ok", d["frames_ok"])'
echo -n "default: " >> $O
timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
echo "configs[4] sweep driver, 5 levels x 65536 frames, batch 32768:" >> $O
timeout 600 python3 tools/ber_sweep.py --frames 65536 --levels -40 -35 -30 -25 -20 2>&1 | tail -6 >> $O
echo "configs[4] sweep driver, batch 16384:" >> $O
timeout 600 python3 tools/ber_sweep.py --frames 65536 --batch 16384 --levels -40 -30 -20 2>&1 | tail -4 >> $O
cat $O
