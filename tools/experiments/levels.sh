for db in "$@"; do python3 bench.py --noise-db $db --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$db dB value', round(d['value']), 'fer', d['fer'], 'routes', d['routes_rank0'], 'llr', round(d['stage_ms_per_launch_alone']['llr'],3), 'sc', round(d['stage_ms_per_launch_alone']['sc'],3))
"; done
