#!/bin/bash
O=$PWD/gpurun_out/two.txt; mkdir -p gpurun_out; : > $O
for w in 12 10 8; do echo "== wpc $w" >> $O; OFDMRX_POLAR_WPC=$w timeout 600 python3 tools/two_handles_probe.py 2>&1 | grep handles >> $O; done
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "ts", round(s["theilsen"],1), "polar", round(s["polar"],1), "fer", d["fer"], "ok", d["frames_ok"])'
for w in 10 11 12; do
	echo -n "bench wpc $w: " >> $O
	OFDMRX_POLAR_WPC=$w timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
cat $O
