#!/bin/bash
# validation_r04.sh -- the large GPU == oracle sweeps and the mode table on the round's final binary (about half an hour of box time)
O=$PWD/gpurun_out/${OUT:-validation.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
echo "== certified levels, 8192 frames each (the queue holds a few stragglers per chunk at -27 dB)" >> $O
timeout 1500 python3 tests/parity_sweep.py 8192 -30 -27 >> $O 2>&1
echo "== mixed levels, 4096 frames each (the list decoder runs in full residencies across chunks)" >> $O
timeout 1500 python3 tests/parity_sweep.py 4096 -26 -25 -24 >> $O 2>&1
echo "== noisy rows, 4096 frames each" >> $O
timeout 1500 python3 tests/parity_sweep.py 4096 -16 -15 >> $O 2>&1
echo "== noisy headers, 2048 frames each" >> $O
timeout 1500 python3 tests/parity_sweep.py 2048 -15 -14.5 -14 -13 >> $O 2>&1
echo "== the many-row modes, 1024 frames each" >> $O
for m in 13 9 7 11; do SWEEP_MODE=$m timeout 900 python3 tests/parity_sweep.py 1024 -19 -17 >> $O 2>&1; done
echo "== mono input (the real part of the noisy stream + a DC offset of 700 LSB), 4096 / 2048 frames each, and 16 kHz mono, 192 each" >> $O
SWEEP_CHANNELS=1 timeout 1500 python3 tests/parity_sweep.py 4096 -30 -26 >> $O 2>&1
SWEEP_CHANNELS=1 SWEEP_DC=-4000 timeout 1500 python3 tests/parity_sweep.py 2048 -22 -18 -17 -16 >> $O 2>&1
SWEEP_CHANNELS=1 SWEEP_RATE=16000 timeout 900 python3 tests/parity_sweep.py 192 -24 -18 >> $O 2>&1
echo "== 16 kHz mode 6, 256 frames each" >> $O
SWEEP_RATE=16000 timeout 900 python3 tests/parity_sweep.py 256 -20 -16 -15 >> $O 2>&1
echo "== all eight modes, bench line" >> $O
OUT=modes_tmp.txt bash tools/experiments/modes_bench.sh > /dev/null 2>&1; cat gpurun_out/modes_tmp.txt >> $O
echo "== waterfall, configs[4] driver" >> $O
timeout 900 python3 tools/ber_sweep.py --frames 65536 --levels -17 -16 -15.5 -15 -14.5 -14 -13 2>&1 | tail -9 >> $O
cat $O
