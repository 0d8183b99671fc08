#!/bin/bash
# exp_parity.sh -- the new full-size tests, the parity sweeps (waterfall; modes / rates) on the current binary, the sweep driver with reuse
O=$PWD/gpurun_out/${OUT:-parity.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size or one_chunk or ber_sweep" 2>&1 | tail -5 >> $O
echo "== levels where the syndrome certificate decides all / most / few frames (1024 frames per level)" >> $O
timeout 900 python3 tests/parity_sweep.py 1024 -30 -26 -24 >> $O 2>&1
echo "== waterfall parity sweep (tests/parity_sweep.py 1024 frames per level)" >> $O
timeout 1500 python3 tests/parity_sweep.py 1024 -20 -17 -16 -15.5 -15 -14.5 -14 >> $O 2>&1
echo "== modes / rates (256 frames per level)" >> $O
for m in 9 10 13; do SWEEP_MODE=$m timeout 900 python3 tests/parity_sweep.py 256 -22 -19 -18 -17 >> $O 2>&1; done
SWEEP_RATE=48000 timeout 900 python3 tests/parity_sweep.py 128 -20 -16 -15 >> $O 2>&1
SWEEP_RATE=44100 timeout 900 python3 tests/parity_sweep.py 128 -20 -16 -15 >> $O 2>&1
SWEEP_RATE=16000 SWEEP_MODE=12 timeout 900 python3 tests/parity_sweep.py 128 -24 -20 -19 >> $O 2>&1
echo "== configs[4] sweep driver, tx reuse 4" >> $O
timeout 600 python3 tools/ber_sweep.py --frames 131072 --tx-reuse 4 --levels -40 -30 -20 2>&1 | tail -4 >> $O
cat $O
