#!/bin/bash
# variants_bench.sh -- bench lines of the non-headline instantiations (other rates, modes, the impaired channel)
O=gpurun_out/variants.txt; : > $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); print(round(d["value"]), "frames/s  fer", d["fer"], " ok", d["frames_ok"], "/", d["frames"])'
run() { echo -n "$*: " >> $O; python3 bench.py "$@" --steps 1 --warmup 1 --cpu-frames 0 2>/dev/null | python3 -c "$pick" >> $O 2>&1; }
run --impair
run --rate 16000 --frames 32768
run --rate 44100 --frames 16384
run --rate 48000 --frames 16384
run --mode 9
run --mode 10
run --mode 13
run --channels 1
cat $O
