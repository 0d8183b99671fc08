#!/bin/bash
# stages.sh "ARGS1" "ARGS2" ... -- stage times of one 8192-frame chunk, kernels back to back, per bench argument set
O=$PWD/gpurun_out/${OUT:-stages.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), {k: round(v, 3) for k, v in s.items()}, "ok", d["frames_ok"])'
for a in "$@"; do
echo -n "[$a] " >> $O
OFDMRX_NO_OVERLAP=1 timeout 300 python3 bench.py --frames 8192 --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 --scl-steps 0 $a 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
cat $O
