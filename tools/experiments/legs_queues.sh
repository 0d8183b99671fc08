#!/bin/bash
# legs_queues.sh -- the default bench line's legs under 4 / 8 / 16 hardware queues
O=$PWD/gpurun_out/${OUT:-legs_queues.txt}; mkdir -p gpurun_out; : > $O
pick='import json,sys
d=json.loads(sys.stdin.readline())
print("value", round(d["value"]), "kernel_only", round(d["value_kernel_only"]))
for k in ("value_config3","value_noise_m20","value_config1_mono"):
    e=d[k]; print("  ", k, round(e["value"]), "one call of 8192:", round(e["value_one_call_of_8192_frames"]), "two lanes:", e["value_two_lanes"] and round(e["value_two_lanes"]["value"]), {a: round(b,1) for a,b in e["stage_ms_per_step"].items()})'
for q in ${QS:-4 8 16}; do
	echo "== GPU_MAX_HW_QUEUES=$q" >> $O
	GPU_MAX_HW_QUEUES=$q timeout 600 python3 bench.py --steps 6 --warmup 2 --cpu-frames 0 ${XARGS:---host-frames 0 --scl-steps 0} 2>/dev/null | tail -1 | python3 -c "$pick" >> $O 2>&1
done
cat $O
