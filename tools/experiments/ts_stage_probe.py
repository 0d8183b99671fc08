#!/usr/bin/env python3
"""ts_stage_probe.py -- per-stage SQ_INSTS_VALU of the rank kernel (k_theil_sen): builds variants of k_theilsen.hip that leave a row
after stage 1..5 (the product source carries no probe code: the early exits are patched into a copy here), runs one 8192-frame
chunk of the bench under rocprofv3 --pmc SQ_INSTS_VALU for each, and prints the cumulative and per-stage vector instructions per row.
Run on the GPU box from the repo root: python3 tools/experiments/ts_stage_probe.py > gpurun_out/<tag>_theil_sen_valu_by_stage.txt"""
import os, re, sqlite3, subprocess, sys, glob
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(R, "modem_amd", "csrc", "k_theilsen.hip")).read()

def patch(s):
    def once(old, new):
        assert s.count(old) == 1, old[:50]
        return s.replace(old, new)
    s = s.replace('namespace rx {\n', 'namespace rx {\n#ifndef TS_STOP\n#define TS_STOP 0\n#endif\n', 1)
    s = once('\tTS_SYNC();\n\tconst float2 sy = theil_sen_wave(s, cols, lane);\n\tif (lane == 0) {\n\t\tslope_all[(size_t)f * ROWS_MAX + j] = sy.x;',
             '\tTS_SYNC();\n\tif (TS_STOP == 1) { if (lane == 0) slope_all[(size_t)f * ROWS_MAX + j] = s.y[0]; return; }\n\tconst float2 sy = theil_sen_wave(s, cols, lane);\n\tif (lane == 0) {\n\t\tslope_all[(size_t)f * ROWS_MAX + j] = sy.x;')
    s = once('\t\t\t\tT = ts_uni(T);\n\t\t\t}\n\t\t}\n\t\tfor (;;) {', '\t\t\t\tT = ts_uni(T);\n\t\t\t}\n\t\t}\n\t\tif (TS_STOP == 2) return make_float2(T, ymin + ymax + sy + sxy);\n\t\tfor (;;) {')
    s = once('\t\t\t\tc_at_T = c_lt;\n', '\t\t\t\tc_at_T = c_lt;\n\t\t\t\tif (TS_STOP == 3 && it == 0) return make_float2(T, (float)(c_lt + c_le));\n')
    s = once('\t\tif (!done && !slow) {\n\t\t\t// ---- the pairs inside [Ta, Tb)', '\t\tif (TS_STOP == 4) return make_float2(T, (float)(ca + cb + c_at_T) + Ta + Tb + To_open);\n\t\tif (!done && !slow) {\n\t\t\t// ---- the pairs inside [Ta, Tb)')
    s = once('\t// ---- intercepts b = y - slope*x, median (sorted position n/2)\n', '\tif (TS_STOP == 5) return make_float2(slope, (float)k[0]);\n\t// ---- intercepts b = y - slope*x, median (sorted position n/2)\n')
    return s

os.makedirs("/tmp/ts_probe", exist_ok=True)
open("/tmp/ts_probe/k_theilsen_probe.hip", "w").write(patch(src))
names = {1: "phases (hard map, conj product, arc tangent) + row load", 2: "+ row statistics, least-squares slope", 3: "+ the first count (keys, 512-key sort with inversion count, uncertain pairs)",
         4: "+ the rest of the search (further counts, secant / open-bracket logic)", 5: "+ the bracket list (keys at the other end, window scan, exact divisions, 64-key sort)", 0: "+ the intercepts = the whole kernel"}
rows = 8192 * 50
cum = {}
for stop in (1, 2, 3, 4, 5, 0):
    env = dict(os.environ, PERFILE_k_theilsen="-DTS_STOP=%d" % stop, SRC_k_theilsen="../../../tmp/ts_probe/k_theilsen_probe.hip")
    # build_variant.sh resolves SRC_ relative to the repo root: give it a path that works from there
    env["SRC_k_theilsen"] = os.path.relpath("/tmp/ts_probe/k_theilsen_probe.hip", R)
    subprocess.check_call(["bash", os.path.join(R, "tools", "build_variant.sh"), "tsstop%d" % stop, ""], env=env, stdout=subprocess.DEVNULL)
    d = "/tmp/ts_probe/pmc%d" % stop
    subprocess.call(["rm", "-rf", d])
    env2 = dict(os.environ, MODEM_AMD_LIB=os.path.join(R, "modem_amd", "lib", "variants", "libofdmrx_tsstop%d.so" % stop), OFDMRX_NO_OVERLAP="1", TMPDIR="/tmp")
    subprocess.call(["rocprofv3", "--pmc", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "-d", d, "-o", "x", "--", "python3", os.path.join(R, "tools", "dev_rate_probe.py"), "-30", "8192"],
                    env=env2, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp")
    db = glob.glob(d + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    got = {}
    for k, cn, n, sm in c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        if "k_theil_sen" in k and "more" not in k:
            got[cn] = sm / n
    cum[stop] = got
    print("TS_STOP=%d  %-90s VALU / row %8.1f   SALU / row %7.1f" % (stop, names[stop], got["SQ_INSTS_VALU"] / rows, got["SQ_INSTS_SALU"] / rows), flush=True)
order = [1, 2, 3, 4, 5, 0]
print("\nper stage (differences), vector instructions per row:")
prev = 0.0
for stop in order:
    v = cum[stop]["SQ_INSTS_VALU"] / rows
    print("  %-95s %8.1f" % (names[stop].lstrip("+ "), v - prev))
    prev = v
