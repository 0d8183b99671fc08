#!/bin/bash
# suite.sh -- the whole GPU test suite, then the bench line at a few noise levels (WORKLOADS) for the current build and VARIANTS
O=$PWD/gpurun_out/${OUT:-suite.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
[ -n "$NOTEST" ] || timeout 2400 python3 -m pytest tests -x -q -m gpu ${K:+-k "$K"} 2>&1 | tail -15 >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; a=d.get("stage_ms_per_launch_alone") or {}
print("value", round(d["value"]), "kernel_only", round(d["value_kernel_only"]), "scl_forced", d["value_scl_forced"] and round(d["value_scl_forced"]), "list_decoded", d["list_decoded_frames_rank0"], "fer", d["fer"], "ok", d["frames_ok"], "| ms/step", " ".join("%s %.1f" % (k, s[k]) for k in ("front","sync","header","demod","theilsen","llr","polar","finish")), "| alone", " ".join("%s %.2f" % (k, a[k]) for k in a))'
run() { echo -n "[${1:-current} | $2] " >> $O; ( [ -n "$1" ] && export MODEM_AMD_LIB=$PWD/modem_amd/lib/variants/libofdmrx_$1.so; timeout 600 python3 bench.py --steps ${STEPS:-4} --warmup 2 --cpu-frames 0 --host-frames 0 $2 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1 ); }
IFS=';' read -ra WL <<< "${WORKLOADS:-;--noise-db=-26;--noise-db=-20;--impair}"
for wl in "${WL[@]}"; do
	run "" "$wl"
	for v in $VARIANTS; do run $v "$wl"; done
done
cat $O
