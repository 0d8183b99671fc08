#!/bin/bash
# validation_r05_rates_2ch.sh -- GPU == oracle on 2-channel input at 16 / 44.1 / 48 kHz (k_demod's prefetch / table changes of round 5), frame by frame
O=$PWD/gpurun_out/${OUT:-validation_r05_rates_2ch.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
export SWEEP_THREADS=128
SWEEP_RATE=16000 timeout 900 python3 tests/parity_sweep.py 384 -30 -22 -16 2>&1 | grep -v amdgpu.ids >> $O
SWEEP_RATE=44100 timeout 1200 python3 tests/parity_sweep.py 192 -30 -21 -16 2>&1 | grep -v amdgpu.ids >> $O
SWEEP_RATE=48000 timeout 1200 python3 tests/parity_sweep.py 192 -30 -21 -16 2>&1 | grep -v amdgpu.ids >> $O
SWEEP_RATE=44100 SWEEP_MODE=9 timeout 1200 python3 tests/parity_sweep.py 96 -30 -14 2>&1 | grep -v amdgpu.ids >> $O
SWEEP_RATE=48000 SWEEP_MODE=13 timeout 1200 python3 tests/parity_sweep.py 96 -30 -14 2>&1 | grep -v amdgpu.ids >> $O
cat $O
