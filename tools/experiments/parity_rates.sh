#!/bin/bash
# parity_rates.sh -- GPU == oracle sweeps at the other sample rates (sync / header fields, payloads, flip counts)
O=$PWD/gpurun_out/parity_rates.txt; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
SWEEP_RATE=48000 timeout 900 python3 tests/parity_sweep.py 128 -20 -16 -15 >> $O 2>&1
SWEEP_RATE=44100 timeout 900 python3 tests/parity_sweep.py 128 -20 -16 -15 >> $O 2>&1
SWEEP_RATE=16000 SWEEP_MODE=12 timeout 900 python3 tests/parity_sweep.py 128 -24 -20 -19 >> $O 2>&1
SWEEP_RATE=16000 timeout 900 python3 tests/parity_sweep.py 128 -16 -15 -14.5 >> $O 2>&1
cat $O
