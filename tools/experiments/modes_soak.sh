#!/bin/bash
# modes_soak.sh -- GPU == oracle frame by frame for every mode, 2-channel and mono input, two noise levels each (2048 frames per point)
O=$PWD/gpurun_out/${OUT:-modes_soak.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
for m in 6 7 8 9 10 11 12 13; do
  for ch in 2 1; do
    SWEEP_THREADS=128 SWEEP_MODE=$m SWEEP_CHANNELS=$ch SWEEP_DC=-2500 timeout 900 python3 tests/parity_sweep.py 2048 -24 -19 2>&1 | grep -v amdgpu.ids >> $O
  done
done
cat $O
