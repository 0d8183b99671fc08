#!/bin/bash
# profile_round.sh TAG -- everything profiles/TAG_* holds, collected on the GPU box from the repo root:
#   kernel trace of the default bench line (overlapped schedule) and of the same with kernels back to back,
#   PMC passes (FETCH_SIZE, WRITE_SIZE, instruction counters: separate runs, one 8192-frame chunk, kernels back to back),
#   the FETCH_SIZE / WRITE_SIZE calibration in k_polar's access pattern (tools/pmc_calib.hip).
# Only text leaves the box: gpurun_out/TAG_summary.txt, TAG_bench_n1*.json, TAG_traffic.json
TAG=${1:-r01_vX}; R=$PWD; G=$R/gpurun_out; mkdir -p $G
S=$G/${TAG}_summary.txt; : > $S
cd /tmp; export TMPDIR=/tmp
echo "# $TAG: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0   (default: overlapped schedule)" >> $S
rocprofv3 --kernel-trace --stats -d /tmp/prof_e -o trace -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 > $G/${TAG}_bench_n1_under_profiler.json 2>/dev/null
python3 $R/profiles/summarize.py $(find /tmp/prof_e -name "*.db" | head -1) >> $S 2>&1
echo "# same with OFDMRX_NO_OVERLAP=1 (every kernel alone on the device)" >> $S
OFDMRX_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_n -o trace -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 > /dev/null 2>&1
python3 $R/profiles/summarize.py $(find /tmp/prof_n -name "*.db" | head -1) >> $S 2>&1
export OFDMRX_NO_OVERLAP=1
echo "# PMC passes: rocprofv3 --pmc <counters> -- python3 bench.py --frames 8192 --steps 1 --warmup 0 --cpu-frames 0 (OFDMRX_NO_OVERLAP=1)" >> $S
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
	d=/tmp/pmc_$(echo $c | tr ' ' '_')
	rocprofv3 --pmc $c -d $d -o x -- python3 $R/bench.py --frames 8192 --steps 1 --warmup 0 --cpu-frames 0 --host-frames 0 > /dev/null 2>&1
	python3 $R/tools/pmc_kernel.py $(find $d -name "*.db" | head -1) rx:: >> $S 2>&1
done
unset OFDMRX_NO_OVERLAP
echo "# calibration: tools/pmc_calib.hip, 2 GiB read / 2 GiB written / 2+2 GiB copied per kernel, one 256-byte row per wave instruction" >> $S
hipcc -w --offload-arch=gfx950 -O3 $R/tools/pmc_calib.hip -o /tmp/pmc_calib && /tmp/pmc_calib >> $S 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
	rocprofv3 --pmc $c -d /tmp/calib_$c -o x -- /tmp/pmc_calib > /dev/null 2>&1
	python3 $R/tools/pmc_kernel.py $(find /tmp/calib_$c -name "*.db" | head -1) calib >> $S 2>&1
done
cd $R
python3 bench.py > $G/${TAG}_bench_n1.json 2>/dev/null
OFDMRX_NO_OVERLAP=1 python3 bench.py --cpu-frames 0 --host-frames 0 > $G/${TAG}_bench_n1_no_overlap.json 2>/dev/null
R=$R python3 - "$S" "$G/${TAG}_traffic.json" <<'PY'
import hashlib, json, os, re, sys
txt = open(sys.argv[1]).read()
sha = hashlib.sha256(open(os.path.join(os.environ.get("R", "."), "modem_amd", "csrc", "k_polar.hip"), "rb").read()).hexdigest()[:16]
def grab(kern, ctr):
    m = re.search(r"^%s\s+%s\s+calls\s+(\d+)\s+sum\s+([0-9.]+)" % (re.escape(kern), ctr), txt, re.M)
    return float(m.group(2)) / int(m.group(1))
f, w = grab("void rx::k_polar<8>", "FETCH_SIZE"), grab("void rx::k_polar<8>", "WRITE_SIZE")
cf, cw = grab("calib_read", "FETCH_SIZE"), grab("calib_write", "WRITE_SIZE")
json.dump({"kernel": "rx::k_polar<8>", "frames_per_launch": 8192, "fetch_KiB": f, "write_KiB": w, "k_polar_src_sha": sha,
           "fetch_scale": 2097152.0 / cf, "write_scale": 2097152.0 / cw,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --frames 8192 --steps 1 --warmup 0 "
                     "--cpu-frames 0, kernels back to back; scales = true bytes / counted bytes of tools/pmc_calib.hip (2 GiB per kernel, "
                     "the decoder's access pattern) in the same session"}, open(sys.argv[2], "w"), indent=1)
PY
tail -5 $S; cat $G/${TAG}_traffic.json; cat $G/${TAG}_bench_n1.json | cut -c1-400
