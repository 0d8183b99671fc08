#!/bin/bash
# sc_runs.sh -- the list-1 pass in whole residencies (round 6): parity tests that touch the pass and the queue, then frames/s at the
# levels where it decides some / all / most frames and on the configs[3] chain.
O=$PWD/gpurun_out/${OUT:-sc_runs.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${K:-sc_ or queue or default_path or certificate or host or skip}" 2>&1 | tail -5 >> $O
for db in -20 -26 -25 -18.5; do
	timeout 300 python3 tools/dev_rate_probe.py $db 2>&1 | tail -2 >> $O
done
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "kernel_only", round(d.get("value_kernel_only") or 0), "routes", d.get("routes_rank0"), "fer", d["fer"], "sc ms", round(s.get("sc", 0), 1))'
for x in "--impair" "--noise-db -20" ""; do
	echo -n "bench $x: " >> $O
	timeout 300 python3 bench.py --steps 4 --warmup 2 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 $x 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
cat $O
