#!/bin/bash
# envs.sh "VAR=val VAR2=val" "..." -- the overlapped bench line under each set of environment knobs (first: none), twice each
O=$PWD/gpurun_out/${OUT:-envs.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "sync", round(s["sync"],2), "header", round(s["header"],2), "demod", round(s["demod"],2), "ts", round(s["theilsen"],1), "polar", round(s["polar"],1), "fer", d["fer"], "ok", d["frames_ok"])'
for e in "" "$@"; do
	for rep in 1 2; do
		echo -n "[$e] overlapped: " >> $O
		env $e timeout 300 python3 bench.py --steps 3 --warmup 1 --cpu-frames 0 --host-frames 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
	done
done
cat $O
