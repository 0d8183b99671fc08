#!/bin/bash
# validation_r06.sh -- GPU == oracle frame by frame where the list-1 pass decides, after its clean-node / half-array changes (round 6): the other 8PSK modes
# (mode 10: the second frozen table), the QPSK modes, mono input.  About 12 minutes of box time, most of it the oracle on the host cores.
O=$PWD/gpurun_out/${OUT:-validation_r06.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
export SWEEP_THREADS=128
echo "== 8PSK modes 7 / 10 / 11, 4096 frames each" >> $O
for m in 7 10 11; do SWEEP_MODE=$m timeout 900 python3 tests/parity_sweep.py 4096 -22 -19 -18.4 >> $O 2>&1; done
echo "== QPSK modes 8 / 9 / 12 / 13, 2048 frames each" >> $O
for m in 8 9 12 13; do SWEEP_MODE=$m timeout 900 python3 tests/parity_sweep.py 2048 -17 -14 >> $O 2>&1; done
echo "== mode 6, mono (DC offset -2500 LSB), 8192 frames each" >> $O
SWEEP_CHANNELS=1 SWEEP_DC=-2500 timeout 1500 python3 tests/parity_sweep.py 8192 -22 -19 >> $O 2>&1
tail -40 $O
