#!/bin/bash
# validation_r05_mono_rates.sh -- GPU == oracle on MONO input at 16 / 44.1 / 48 kHz (the rebuilt front pass), frame by frame
O=$PWD/gpurun_out/${OUT:-validation_r05_mono_rates.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
export SWEEP_THREADS=128 SWEEP_CHANNELS=1
SWEEP_RATE=16000 SWEEP_DC=900 timeout 900 python3 tests/parity_sweep.py 384 -30 -22 -16 2>&1 | grep -v amdgpu.ids >> $O
SWEEP_RATE=44100 SWEEP_DC=-1200 timeout 1200 python3 tests/parity_sweep.py 192 -30 -21 -16 2>&1 | grep -v amdgpu.ids >> $O
SWEEP_RATE=48000 SWEEP_DC=2500 timeout 1200 python3 tests/parity_sweep.py 192 -30 -21 -16 2>&1 | grep -v amdgpu.ids >> $O
cat $O
