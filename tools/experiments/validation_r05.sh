#!/bin/bash
# validation_r05.sh -- GPU == oracle frame by frame at the noise levels where the list-1 pass (k_sc) decides, on the round's final binary:
# mode 6 analytic from the first raw bit errors to past the rule's reach, mono input, the QPSK modes down to THEIR reach, the second
# frozen table, 16 kHz; then the waterfall again (about 35 minutes of box time, most of it the oracle on the host cores)
O=$PWD/gpurun_out/${OUT:-validation_r05.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
export SWEEP_THREADS=128
echo "== mode 6, 2-channel, 16384 frames each: every frame has raw bit errors from -24 dB on; the rule holds to -18.4 dB, in part at -18.2, not at -18" >> $O
timeout 2400 python3 tests/parity_sweep.py 16384 -24 -21 -19 -18.4 -18.2 -18 >> $O 2>&1
echo "== mode 6, mono (DC offset -2500 LSB), 8192 frames each" >> $O
SWEEP_CHANNELS=1 SWEEP_DC=-2500 timeout 1500 python3 tests/parity_sweep.py 8192 -22 -19 >> $O 2>&1
echo "== QPSK modes 8 / 9 / 12 / 13 (the rule reaches -13.5 dB there), 4096 frames each" >> $O
for m in 8 9 12 13; do SWEEP_MODE=$m timeout 900 python3 tests/parity_sweep.py 4096 -17 -14 -13 >> $O 2>&1; done
echo "== 8PSK modes 7 / 10 / 11 (mode 10: the second frozen table), 4096 frames each" >> $O
for m in 7 10 11; do SWEEP_MODE=$m timeout 900 python3 tests/parity_sweep.py 4096 -21 -19 -18.3 >> $O 2>&1; done
echo "== 16 kHz, mode 6, 256 frames each" >> $O
SWEEP_RATE=16000 timeout 900 python3 tests/parity_sweep.py 256 -21 -19 >> $O 2>&1
echo "== the waterfall, 8192 frames each" >> $O
timeout 1500 python3 tests/parity_sweep.py 8192 -15.5 -15 -14.5 >> $O 2>&1
cat $O
