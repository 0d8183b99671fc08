python3 bench.py --noise-db -20 --steps 3 --warmup 1 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'fer', d['fer'], 'listed', d['list_decoded_frames_rank0'], 'alone total', round(d['stage_ms_per_launch_alone']['total'],2))
"
