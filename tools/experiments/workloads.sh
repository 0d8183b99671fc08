#!/bin/bash
# workloads.sh -- the bench line of every workload flavour DESIGN.md quotes (value = default path, scl = list decoder for every frame)
O=$PWD/gpurun_out/${OUT:-workloads.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); print("value", round(d["value"]), "kernel_only", round(d["value_kernel_only"]), "scl_forced", d["value_scl_forced"] and round(d["value_scl_forced"]), "list_decoded", d["list_decoded_frames_rank0"], "fer", d["fer"], "ok", d["frames_ok"], "of", d["frames"])'
run() { echo -n "[$*] " >> $O; timeout 600 python3 bench.py --steps 4 --warmup 2 --cpu-frames 0 --host-frames 0 "$@" 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1; }
run
run --impair
run --channels 1
run --noise-db -26
run --noise-db -24
run --noise-db -20
run --noise-db -16
run --list 4
run --scaling strong
run --frames 8192
run --rate 16000 --frames 32768
run --rate 44100 --frames 16384
run --rate 48000 --frames 16384
run --mode 9
run --mode 10
run --mode 13
cat $O
