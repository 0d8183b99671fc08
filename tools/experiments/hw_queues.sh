#!/bin/bash
# hw_queues.sh -- does the pipeline of ONE handle lose to HIP's stream -> hardware-queue mapping?  (GPU_MAX_HW_QUEUES, default 4:
# streams beyond that share a queue and their kernels run in order.)  The bench line's value / value_kernel_only / scl-forced at 4, 8, 16
# queues, the -20 dB and configs[3] workloads, and two handles on halves of the batch at -30 dB.
O=$PWD/gpurun_out/${OUT:-hw_queues.txt}; mkdir -p gpurun_out; : > $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); print("value", round(d["value"]), "kernel_only", round(d.get("value_kernel_only") or 0), "scl_forced", round((d.get("value_scl_forced") or 0)))'
for q in 4 8 16; do
	echo "== GPU_MAX_HW_QUEUES=$q" >> $O
	for x in "" "--noise-db -20" "--impair"; do
		echo -n "bench $x: " >> $O
		GPU_MAX_HW_QUEUES=$q timeout 300 python3 bench.py --steps 4 --warmup 2 --cpu-frames 0 --host-frames 0 --scl-steps 2 --leg-steps 0 $x 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
	done
	echo "two handles, -30 dB:" >> $O
	GPU_MAX_HW_QUEUES=$q PROBE_NOISE_DB=-30 timeout 600 python3 tools/two_handles_probe.py 2>&1 | grep handles >> $O
done
cat $O
