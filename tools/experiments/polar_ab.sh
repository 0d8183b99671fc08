#!/bin/bash
# polar_ab.sh -- list-decoder parity tests on the current build, then the list-decoder-forced bench line of the current build
# against variant libraries (VARIANTS="r3 ..." = modem_amd/lib/variants/libofdmrx_<name>.so) on the same box
O=$PWD/gpurun_out/${OUT:-polar_ab.txt}; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
( [ -n "$TESTLIB" ] && export MODEM_AMD_LIB=$PWD/modem_amd/lib/variants/libofdmrx_$TESTLIB.so; timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${K:-polar or certificate or awgn or waterfall or list_size or impairment or all_modes}" 2>&1 | tail -6 >> $O )
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_step"]; print("value", round(d["value"]), "scl_forced", d["value_scl_forced"] and round(d["value_scl_forced"]), "polar_ms/step", round(s["polar"],1), "list_decoded", d["list_decoded_frames_rank0"], "fer", d["fer"], "ok", d["frames_ok"], "polar_ms/step scl", round(d["stage_ms_per_step_scl_forced"]["polar"],1), "same", "identical to the default path: True" in (d["value_scl_forced_definition"] or ""))'
# VARIANTS entries: name or name:decoders_per_cu
run() { echo -n "[${1:-current} wpc ${3:-default} | $2] " >> $O; ( [ -n "$1" ] && export MODEM_AMD_LIB=$PWD/modem_amd/lib/variants/libofdmrx_$1.so; [ -n "$3" ] && export OFDMRX_POLAR_WPC=$3; timeout 600 python3 bench.py --steps 3 --warmup 1 --cpu-frames 0 --host-frames 0 $2 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1 ); }
for wl in ${WORKLOADS:-"--noise-db=-20" "--impair" ""}; do
	run "" "$wl"
	for v in $VARIANTS; do run ${v%%:*} "$wl" $( [[ $v == *:* ]] && echo ${v##*:} ); done
done
cat $O
