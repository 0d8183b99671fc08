#!/bin/bash
# exp_variants.sh -- the non-headline instantiations on the current binary: stage times, one chunk alone and the default schedule
O=$PWD/gpurun_out/variants.txt; mkdir -p gpurun_out; : > $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), " ".join("%s %.1f" % (k, v) for k, v in s.items() if k != "total"), "fer", d["fer"], "ok", d["frames_ok"])'
run() { echo -n "$1: " >> $O; shift; timeout 600 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 "$@" 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1; }
run "list 4, 65536 frames" --list 4
OFDMRX_NO_OVERLAP=1 run "list 4, one chunk alone" --list 4 --frames 8192
run "16 kHz, 32768 frames" --rate 16000 --frames 32768
OFDMRX_NO_OVERLAP=1 run "16 kHz, one chunk alone" --rate 16000 --frames 8192
run "44.1 kHz, 16384 frames" --rate 44100 --frames 16384
OFDMRX_NO_OVERLAP=1 run "44.1 kHz, one chunk (4096) alone" --rate 44100 --frames 4096
run "48 kHz, 16384 frames" --rate 48000 --frames 16384
OFDMRX_NO_OVERLAP=1 run "48 kHz, one chunk (4096) alone" --rate 48000 --frames 4096
run "mode 9" --mode 9
run "mode 10" --mode 10
run "mode 13" --mode 13
run "mono clean (configs[1] flavour)" --channels 1
OFDMRX_NO_OVERLAP=1 run "mono clean, one chunk alone" --channels 1 --frames 8192
cat $O
