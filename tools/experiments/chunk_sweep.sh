#!/bin/bash
# chunk_sweep.sh -- the bench line for different chunk sizes (frames per pipeline stage)
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "polar", round(s["polar"],1), "ts", round(s["theilsen"],1), "fer", d["fer"])'
for c in ${CHUNKS:-4096 6144 8192 9216 12288 16384}; do echo -n "chunk $c: "; python3 bench.py --chunk $c --steps 2 --warmup 1 --cpu-frames 0 2>/dev/null | python3 -c "$pick"; done
