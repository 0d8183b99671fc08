#!/bin/bash
# profile_round.sh TAG -- everything profiles/TAG_* holds, collected on the GPU box from the repo root:
#   kernel traces of the default bench line (pipeline), of the same with kernels back to back, and of the same with the list
#   decoder forced for every frame (OFDMRX_NO_CERT=1: no syndrome certificate);
#   PMC passes (FETCH_SIZE, WRITE_SIZE, instruction counters: separate runs, 8192-frame chunks, kernels back to back), in three modes:
#   the default path at -30 dB [cert], the list decoder forced [scl], the default path at -20 dB where the list-1 pass decides [sc];
#   the FETCH_SIZE / WRITE_SIZE calibration (tools/pmc_calib.hip); the issue-rate microbenchmark (tools/ubench_issue.hip).
# Only text leaves the box: gpurun_out/TAG_summary.txt, TAG_bench_n1*.json, TAG_issue_rates_ubench.txt, TAG_traffic.json (copy the
# last two to profiles/r06_issue_rates_ubench.txt / profiles/r06_traffic.json: bench.py reads the per-kernel HBM bytes, VALU
# instruction counts and the measured issue ceiling from the latter)
TAG=${1:-r06_vX}; R=$PWD; G=$R/gpurun_out; mkdir -p $G
S=$G/${TAG}_summary.txt; : > $S
make -C modem_amd/csrc -q all && echo "# library up to date with sources" >> $S || echo "# STALE LIBRARY" >> $S
cd /tmp; export TMPDIR=/tmp
export OFDMRX_NO_TAIL_SPLIT=1     # (every launch of the profiled runs a whole chunk of 8192 frames: the per-launch figures below divide by the launches)
B="python3 $R/bench.py --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0"
echo "# $TAG: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0   (default: pipeline, syndrome certificate on)" >> $S
rocprofv3 --kernel-trace --stats -d /tmp/prof_e -o trace -- $B --steps 2 --warmup 1 > $G/${TAG}_bench_n1_under_profiler.json 2>/dev/null
python3 $R/profiles/summarize.py $(find /tmp/prof_e -name "*.db" | head -1) >> $S 2>&1
echo "# same with OFDMRX_NO_OVERLAP=1 (every kernel alone on the device)" >> $S
OFDMRX_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_n -o trace -- $B --steps 2 --warmup 1 > /dev/null 2>&1
python3 $R/profiles/summarize.py $(find /tmp/prof_n -name "*.db" | head -1) >> $S 2>&1
echo "# OFDMRX_NO_CERT=1 OFDMRX_NO_SC=1 (list decoder for every frame), pipeline" >> $S
OFDMRX_NO_CERT=1 OFDMRX_NO_SC=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o trace -- $B --steps 2 --warmup 1 > /dev/null 2>&1
python3 $R/profiles/summarize.py $(find /tmp/prof_s -name "*.db" | head -1) >> $S 2>&1
echo "# --noise-db -20 (every frame has raw bit errors: the list-1 pass k_sc decides them), OFDMRX_NO_OVERLAP=1" >> $S
OFDMRX_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o trace -- $B --noise-db -20 --steps 2 --warmup 1 > /dev/null 2>&1
python3 $R/profiles/summarize.py $(find /tmp/prof_c -name "*.db" | head -1) >> $S 2>&1
echo "# OFDMRX_NO_CERT=1 OFDMRX_NO_SC=1 OFDMRX_NO_OVERLAP=1" >> $S
OFDMRX_NO_CERT=1 OFDMRX_NO_SC=1 OFDMRX_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_sn -o trace -- $B --steps 2 --warmup 1 > /dev/null 2>&1
python3 $R/profiles/summarize.py $(find /tmp/prof_sn -name "*.db" | head -1) >> $S 2>&1
export OFDMRX_NO_OVERLAP=1
# (16384 frames = two chunks of 8192 per call: a call of one chunk whose outputs go to the host would run as two halves in one of the three steps only)
for mode in cert scl sc; do
	unset OFDMRX_NO_CERT OFDMRX_NO_SC; X=""
	[ $mode = scl ] && export OFDMRX_NO_CERT=1 OFDMRX_NO_SC=1
	[ $mode = sc ] && X="--noise-db -20"
	echo "# PMC passes [$mode]: rocprofv3 --pmc <counters> -- python3 bench.py --frames 16384 --steps 1 --warmup 0 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 $X (OFDMRX_NO_OVERLAP=1$([ $mode = scl ] && echo ' OFDMRX_NO_CERT=1 OFDMRX_NO_SC=1'))" >> $S
	for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS SQ_WAIT_INST_ANY"; do
		d=/tmp/pmc_${mode}_$(echo $c | tr ' ' '_')
		rocprofv3 --pmc $c -d $d -o x -- $B --frames 16384 --steps 1 --warmup 0 $X > /dev/null 2>&1
		python3 $R/tools/pmc_kernel.py $(find $d -name "*.db" | head -1) rx:: | sed "s/^/[$mode] /" >> $S 2>&1
	done
done
unset OFDMRX_NO_CERT OFDMRX_NO_SC
echo "# mono input (--channels 1: clean 16-bit mono frames, configs[1] flavour), OFDMRX_NO_OVERLAP=1: kernel trace, then PMC passes of one 8192-frame chunk" >> $S
rocprofv3 --kernel-trace --stats -d /tmp/prof_m -o trace -- $B --channels 1 --steps 2 --warmup 1 > $G/${TAG}_bench_mono_under_profiler.json 2>/dev/null
python3 $R/profiles/summarize.py $(find /tmp/prof_m -name "*.db" | head -1) >> $S 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
	d=/tmp/pmc_mono_$(echo $c | tr ' ' '_')
	rocprofv3 --pmc $c -d $d -o x -- $B --channels 1 --frames 8192 --steps 1 --warmup 0 > /dev/null 2>&1
	python3 $R/tools/pmc_kernel.py $(find $d -name "*.db" | head -1) rx:: | sed "s/^/[mono] /" >> $S 2>&1
done
unset OFDMRX_NO_OVERLAP
echo "# calibration: tools/pmc_calib.hip, 2 GiB read / 2 GiB written / 2+2 GiB copied per kernel, one 256-byte row per wave instruction" >> $S
hipcc -w --offload-arch=gfx950 -O3 $R/tools/pmc_calib.hip -o /tmp/pmc_calib && /tmp/pmc_calib >> $S 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
	rocprofv3 --pmc $c -d /tmp/calib_$c -o x -- /tmp/pmc_calib > /dev/null 2>&1
	python3 $R/tools/pmc_kernel.py $(find /tmp/calib_$c -name "*.db" | head -1) calib >> $S 2>&1
done
echo "# issue rates: tools/ubench_issue.hip (ns per wave-instruction per SIMD, every CU busy; s_memtime against the 100 MHz clock)" >> $S
hipcc -w --offload-arch=gfx950 -O3 $R/tools/ubench_issue.hip -o /tmp/ubench_issue && /tmp/ubench_issue > $G/${TAG}_issue_rates_ubench.txt 2>&1
head -4 $G/${TAG}_issue_rates_ubench.txt >> $S
cd $R
unset OFDMRX_NO_TAIL_SPLIT
python3 bench.py > $G/${TAG}_bench_n1.json 2>/dev/null
R=$R python3 - "$S" "$G/${TAG}_traffic.json" "$G/${TAG}_issue_rates_ubench.txt" <<'PY'
import hashlib, json, os, re, sys
txt = open(sys.argv[1]).read()
root = os.environ.get("R", ".")
def sha(f):
    return hashlib.sha256(open(os.path.join(root, "modem_amd", "csrc", f), "rb").read()).hexdigest()[:16]
def grab(mode, kern, ctr):
    """per chunk: the sum over every kernel of the stage (the sync stage is several kernels, some launched twice per chunk) / the
    number of chunks (= the calls of the kernel launched least often: each stage has one that runs once per chunk)"""
    tot = 0.0; n = 0
    for m in re.finditer(r"^\[%s\] (\S.*?)\s+%s\s+calls\s+(\d+)\s+sum\s+([0-9.]+)" % (mode, ctr), txt, re.M):
        if kern in m.group(1):
            tot += float(m.group(3)); n = int(m.group(2)) if not n else min(n, int(m.group(2)))
    return tot / n if n else None
def calib(kern, ctr):
    m = re.search(r"^%s\s+%s\s+calls\s+(\d+)\s+sum\s+([0-9.]+)" % (re.escape(kern), ctr), txt, re.M)
    return float(m.group(2)) / int(m.group(1))
stages = {"sync": ("cert", "k_sync", "k_sync.hip"), "header": ("cert", "k_header", "k_header.hip"), "demod": ("cert", "k_demod", "k_demod.hip"),
          "theilsen": ("cert", "k_theil_sen", "k_theilsen.hip"), "llr": ("cert", "k_back", "k_finish.hip"), "finish": ("cert", "k_finish", "k_finish.hip"),
          "polar": ("scl", "k_polar", "k_polar.hip"), "sc": ("sc", "k_sc", "k_sc.hip")}
ub = open(sys.argv[3]).read() if len(sys.argv) > 3 and os.path.exists(sys.argv[3]) else ""
# the ceiling = the best in-kernel rate of plain fp32 FMAs with one workgroup per CU and 2 - 4 waves per SIMD (placement verified by the benchmark itself)
mu = min(((float(a), float(g)) for w, a, g in re.findall(r"^v_fma_f32 x8 independent\s+waves/SIMD=(\d)\s+ns/valu/SIMD: in-kernel ([0-9.]+).*?clock=([0-9.]+) GHz", ub, re.M) if 2 <= int(w) <= 4), default=None)
out = {"frames_per_launch": 8192, "valu_ns_per_inst_per_simd": mu[0] if mu else None, "sclk_GHz_measured": mu[1] if mu else None, "fetch_scale": 2097152.0 / calib("calib_read", "FETCH_SIZE"), "write_scale": 2097152.0 / calib("calib_write", "WRITE_SIZE"),
       "kernels": {},
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --frames 16384 --steps 1 --warmup 0 --cpu-frames 0 "
                 "--host-frames 0 --scl-steps 0 --leg-steps 0, kernels back to back (k_polar: with OFDMRX_NO_CERT=1; k_sc: with --noise-db -20); KiB per launch of 8192 frames; scales = true bytes / counted "
                 "bytes of tools/pmc_calib.hip (2 GiB per kernel, one dword per lane) in the same session; valu_insts = SQ_INSTS_VALU (wave instructions) per launch"}
for st, (mode, kern, src) in stages.items():
    f, w = grab(mode, kern, "FETCH_SIZE"), grab(mode, kern, "WRITE_SIZE")
    if f is not None and w is not None:
        out["kernels"][st] = {"kernel": kern, "fetch_KiB": f, "write_KiB": w, "valu_insts": grab(mode, kern, "SQ_INSTS_VALU"), "src_sha": sha(src)}
# mono input (configs[1]): the same stages from the [mono] passes (one 8192-frame chunk); `front` = k_mono_carries
out["kernels_mono"] = {}
for st, (kern, src) in {"front": ("k_mono_carries", "k_sync.hip"), "sync": ("k_sync", "k_sync.hip"), "header": ("k_header", "k_header.hip"),
                        "demod": ("k_demod", "k_demod.hip"), "theilsen": ("k_theil_sen", "k_theilsen.hip"), "llr": ("k_back", "k_finish.hip")}.items():
    f, w = grab("mono", kern, "FETCH_SIZE"), grab("mono", kern, "WRITE_SIZE")
    if f is not None and w is not None:
        out["kernels_mono"][st] = {"kernel": kern, "fetch_KiB": f, "write_KiB": w, "valu_insts": grab("mono", kern, "SQ_INSTS_VALU"), "src_sha": sha(src)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
PY
tail -5 $S; cat $G/${TAG}_traffic.json; cat $G/${TAG}_bench_n1.json | cut -c1-600
