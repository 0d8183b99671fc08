#!/bin/bash
# modes_bench.sh -- the default bench line (65536 frames, pipeline) per operation mode; LIBS="a b" adds variant libraries
O=$PWD/gpurun_out/${OUT:-modes_bench.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "kernel_only", round(d["value_kernel_only"]), "ts", round(s["theilsen"], 2), "ok", d["frames_ok"], "of", d["frames"])'
for lib in default $LIBS; do
	L=$PWD/modem_amd/lib/variants/libofdmrx_$lib.so; [ $lib = default ] && L=$PWD/modem_amd/lib/libofdmrx.so
	for m in ${MODES:-6 7 8 9 10 11 12 13}; do
		echo -n "[$lib --mode $m] " >> $O
		MODEM_AMD_LIB=$L timeout 600 python3 bench.py --steps 4 --warmup 2 --cpu-frames 0 --host-frames 0 --scl-steps 0 --mode $m 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1
	done
done
cat $O
