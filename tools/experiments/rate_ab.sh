#!/bin/bash
# rate_ab.sh RATE CHANNELS "lib lib ..." -- the bench line's value and stage times at a sample rate, per variant library ("product" = the build)
RATE=$1; CH=$2; shift 2
for lib in $1; do
	[ "$lib" != product ] && export MODEM_AMD_LIB=$PWD/modem_amd/lib/variants/libofdmrx_$lib.so || unset MODEM_AMD_LIB
	python3 bench.py --rate $RATE --channels $CH --frames 16384 --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
st=d['stage_ms_per_launch_alone']
print('$lib $RATE Hz $CH ch: value', round(d['value']), 'fer', d['fer'], {k: round(v,3) for k,v in st.items() if k in ('sync','header','demod','total')})
"; done
