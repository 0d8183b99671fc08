#!/bin/bash
# lanes_ab.sh -- one / two lanes on the -20 dB workload (the call-by-call choice of profiles/r06_hw_queues_and_two_lanes.txt item 4 was OFDMRX_LANES=0 in the builds of that day), same box, alternating, `value` (pinned host outputs) and kernel-only
O=$PWD/gpurun_out/${OUT:-lanes_ab.txt}; mkdir -p gpurun_out; : > $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "kernel_only", round(d.get("value_kernel_only") or 0), {k: round(v,1) for k,v in s.items()})'
for rep in 1 2; do
for env in "OFDMRX_LANES=1" "OFDMRX_LANES=2"; do
	echo -n "[$env] ${X:---noise-db -20}: " >> $O
	env $env timeout 300 python3 bench.py --steps ${STEPS:-6} --warmup ${WARM:-3} --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 ${X:---noise-db -20} 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
done
cat $O
