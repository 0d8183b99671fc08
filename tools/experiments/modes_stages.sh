#!/bin/bash
# modes_stages.sh [noise_db] -- per operation mode: the bench line's value, routes and stage times per launch (8 kHz, analytic)
DB=${1:--30}
for m in 6 7 8 9 10 11 12 13; do
python3 bench.py --mode $m --noise-db $DB --frames 16384 --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
st=d['stage_ms_per_launch_alone']
print('mode $m $DB dB: value', round(d['value']), 'fer', d['fer'], d['routes_rank0'], {k: round(v,3) for k,v in st.items() if v > 0.02})
"; done
