#!/bin/bash
# Theil-Sen stage per operation mode (one 8192-frame chunk, kernels back to back) + the tests that touch it
O=$PWD/gpurun_out/${OUT:-ts_modes.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${K:-theil or all_modes or mixed_mode or other_rates_decode}" 2>&1 | tail -3 >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), {k: round(v, 3) for k, v in s.items()}, "ok", d["frames_ok"])'
for m in ${MODES:-6 7 10 13}; do
echo -n "mode $m: " >> $O
OFDMRX_NO_OVERLAP=1 timeout 300 python3 bench.py --mode $m --frames 8192 --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 --scl-steps 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
cat $O
