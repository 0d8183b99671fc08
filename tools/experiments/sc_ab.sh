#!/bin/bash
# sc_ab.sh "name[:ENV=val] ..." -- the legs that lean on k_sc (-20 dB, configs[3]) with modem_amd/lib/variants/libofdmrx_<name>.so beside the default library
O=$PWD/gpurun_out/${OUT:-sc_ab.txt}; : > $O
pick='import json,sys
d=json.loads(sys.stdin.readline())
o=[]
for k in ("value_config3", "value_noise_m20"):
    l = d.get(k)
    if isinstance(l, dict):
        t = l.get("value_two_lanes") or {}
        o.append("%s %d (two lanes %d) sc %.2f ms llr %.2f ms" % (k[6:], l["value"], t.get("value", 0), l["stage_ms_per_step"]["sc"], l["stage_ms_per_step"]["llr"]))
print("; ".join(o))'
for spec in default $1; do
	lib=${spec%%:*}; envs=""; [ "$spec" != "$lib" ] && envs=${spec#*:}
	L=$PWD/modem_amd/lib/variants/libofdmrx_$lib.so; [ $lib = default ] && L=$PWD/modem_amd/lib/libofdmrx.so
	echo -n "$spec: " >> $O
	env MODEM_AMD_LIB=$L $envs timeout 400 python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 3 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
cat $O
