#!/bin/bash
# headline_ab.sh "lib lib ..." -- the headline workload (-30 dB) per variant library ("product" = the build): value and the stages' times, same call
for lib in $1; do
	[ "$lib" != product ] && export MODEM_AMD_LIB=$PWD/modem_amd/lib/variants/libofdmrx_$lib.so || unset MODEM_AMD_LIB
	python3 bench.py --steps 4 --warmup 2 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
st=d['stage_ms_per_launch_alone']
print('$lib: value', round(d['value']), 'kernel_only', round(d['value_kernel_only']), 'fer', d['fer'], {k: round(v,3) for k,v in st.items() if v > 0.02})
"; done
