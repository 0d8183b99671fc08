#!/bin/bash
# ts_rank.sh -- the rank-counting Theil-Sen against the pair-classifying one of round 2: parity tests, the kernel alone on 51 200
# synthetic rows (tools/ts_probe.cpp), and the bench alone / overlapped
O=$PWD/gpurun_out/${OUT:-ts_rank.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "theil" 2>&1 | tail -15 >> $O
( cd tools
for v in "rank|../modem_amd/csrc/k_theilsen.hip|" "pairs|../modem_amd/csrc/k_theilsen_pairs.hip|" $EXTRA; do
	IFS='|' read name src flags <<< "$v"
	hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -I../modem_amd/csrc -DVARIANT="\"$name\"" -DTS_SRC="\"$src\"" $flags ts_probe.cpp -o /tmp/tsp_$name && timeout 120 /tmp/tsp_$name | tail -1 >> $O
done )
K="clean or awgn or failure or all_modes or mixed_mode or config" OUT=ts_rank_quick.txt bash tools/experiments/quick.sh > /dev/null 2>&1
cat gpurun_out/ts_rank_quick.txt >> $O
cat $O
