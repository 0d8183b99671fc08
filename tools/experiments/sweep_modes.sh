#!/bin/bash
# sweep_modes.sh -- tests/parity_sweep.py for the other code / modulation combinations and the highest rate
O=gpurun_out/r01_v9_parity_sweep_modes_rates.txt; : > $O
for m in 9 10 13; do SWEEP_MODE=$m python3 tests/parity_sweep.py 512 -20 -17 -16 -15 >> $O 2>&1; done
SWEEP_MODE=6 SWEEP_RATE=48000 python3 tests/parity_sweep.py 256 -20 -16 -15 >> $O 2>&1
SWEEP_MODE=12 SWEEP_RATE=16000 python3 tests/parity_sweep.py 256 -20 -16 -15 >> $O 2>&1
grep -E "^mode|mismatch" $O
