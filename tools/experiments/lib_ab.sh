#!/bin/bash
# lib_ab.sh "name1 name2 ..." -- the headline bench (value, kernel-only, per-kernel times alone) of modem_amd/lib/variants/libofdmrx_<name>.so beside the default library, alternating
O=$PWD/gpurun_out/${OUT:-lib_ab.txt}; : > $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_launch_alone"]; print("value", round(d["value"]), "kernel_only", round(d["value_kernel_only"]), "fer", d["fer"], "alone:", {k: round(v,3) for k,v in s.items() if k in ("sync","header","demod","theilsen","llr","sc")})'
for rep in 1 2; do
for lib in default $1; do
	L=$PWD/modem_amd/lib/variants/libofdmrx_$lib.so; [ $lib = default ] && L=$PWD/modem_amd/lib/libofdmrx.so
	echo -n "$lib ${X}: " >> $O
	MODEM_AMD_LIB=$L timeout 300 python3 bench.py --steps 6 --warmup 2 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 $X 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
done
done
cat $O
