#!/bin/bash
# validation_r05_chain.sh -- GPU == oracle frame by frame on configs[3]'s impairment chain (multipath -> CFO -> SFO -> AWGN), where two
# thirds of the frames are decided by the list-1 pass and the rest by the syndrome certificate; then the chain at noisier levels
O=$PWD/gpurun_out/${OUT:-validation_r05_chain.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
export SWEEP_THREADS=128 SWEEP_CHAIN=1
timeout 2400 python3 tests/parity_sweep.py 16384 -30 -24 -20 -18 2>&1 | grep -v amdgpu.ids >> $O
echo "== the same chain, mono input (DC offset 1500 LSB), 4096 frames each" >> $O
SWEEP_CHANNELS=1 SWEEP_DC=1500 timeout 1200 python3 tests/parity_sweep.py 4096 -30 -22 2>&1 | grep -v amdgpu.ids >> $O
cat $O
