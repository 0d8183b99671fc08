#!/bin/bash
# lanes_try.sh -- the two-lane device entry: its test, then the bench legs with one / two lanes and 4 / 8 hardware queues
O=$PWD/gpurun_out/${OUT:-lanes_try.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${K:-two_lanes}" 2>&1 | tail -5 >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); print("value", round(d["value"]), "kernel_only", round(d.get("value_kernel_only") or 0), "routes", d.get("routes_rank0"), "fer", d["fer"])'
for env in ${ENVS:-"OFDMRX_LANES=1 GPU_MAX_HW_QUEUES=4" "OFDMRX_LANES=2 GPU_MAX_HW_QUEUES=4" "OFDMRX_LANES=2 GPU_MAX_HW_QUEUES=8"}; do
	echo "== $env" >> $O
	for x in "" "--noise-db -20" "--impair" "--noise-db -26"; do
		echo -n "bench $x: " >> $O
		env $env timeout 300 python3 bench.py --steps 4 --warmup 2 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 $x 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
	done
done
cat $O
