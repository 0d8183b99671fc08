#!/bin/bash
# rates_stages.sh -- per sample rate: the bench line's value and the stages' times per launch (kernels back to back), analytic and mono
for rate in 8000 16000 44100 48000; do for chn in 2 1; do
python3 bench.py --rate $rate --channels $chn --frames 16384 --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
st=d['stage_ms_per_launch_alone']
print('$rate Hz $chn ch: value', round(d['value']), 'frames/launch', d['roofline'].get('frames_per_launch'), {k: round(v,3) for k,v in st.items()})
"; done; done
