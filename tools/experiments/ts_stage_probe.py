#!/usr/bin/env python3
"""ts_stage_probe.py build | run -- where the rank kernel (k_theil_sen) spends its time, stage by stage.
build (here, no GPU): variants of k_theilsen.hip that leave a row after stage 1..5 (the product source carries no probe code: the
  early exits are patched into a copy), linked with the product's other objects into modem_amd/lib/variants/libofdmrx_tsstopN.so
  (git-ignored, travels with gpurun).
run (on the GPU box, from the repo root): one 8192-frame chunk of the bench under rocprofv3 --pmc for each variant; prints, cumulative
  and per stage: vector / scalar / LDS instructions per row, the kernel's time, and what the waves did with their cycles
  (SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY = waiting to issue / SQ_WAIT_ANY = parked at s_waitcnt; quad-cycles).
  python3 tools/experiments/ts_stage_probe.py run > gpurun_out/<tag>_theil_sen_by_stage.txt"""
import os, re, sqlite3, subprocess, sys, glob
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(R, "modem_amd", "csrc")
VAR = os.path.join(R, "modem_amd", "lib", "variants")

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

names = {1: "phases (hard map, conj product, arc tangent) + row load", 2: "+ row statistics, least-squares slope", 3: "+ the first count (keys, 512-key sort with inversion count, uncertain pairs)",
         4: "+ the rest of the search (further counts, secant / open-bracket logic)", 5: "+ the bracket list (keys at the other end, window scan, exact divisions, 64-key sort)", 0: "+ the intercepts = the whole kernel"}
order = [1, 2, 3, 4, 5, 0]
CTRS = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES"]

def build():
    src = open(os.path.join(CS, "k_theilsen.hip")).read()
    os.makedirs("/tmp/ts_probe", exist_ok=True)
    os.makedirs(VAR, exist_ok=True)
    open("/tmp/ts_probe/k_theilsen_probe.hip", "w").write(patch(src))
    subprocess.check_call(["make", "-C", CS, "-j8", "all"], stdout=subprocess.DEVNULL)
    objs = [os.path.join(CS, f) for f in os.listdir(CS) if f.endswith(".o") and f not in ("k_theilsen.o", "decode_main.o", "encode_main.o")]
    procs = []
    for stop in order:
        o = "/tmp/ts_probe/k_theilsen_%d.o" % stop
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-w", "-mllvm", "-disable-machine-licm",
                                       "-DTS_STOP=%d" % stop, "-I" + CS, "-c", "/tmp/ts_probe/k_theilsen_probe.hip", "-o", o]))
    for p in procs:
        assert p.wait() == 0
    for stop in order:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(VAR, "libofdmrx_tsstop%d.so" % stop),
                               "/tmp/ts_probe/k_theilsen_%d.o" % stop] + objs)
    print("built", VAR)

def run():
    rows = 8192 * 50
    cum = {}
    for stop in order:
        d = "/tmp/ts_probe/pmc%d" % stop
        subprocess.call(["rm", "-rf", d])
        env2 = dict(os.environ, MODEM_AMD_LIB=os.path.join(VAR, "libofdmrx_tsstop%d.so" % stop), OFDMRX_NO_OVERLAP="1", TMPDIR="/tmp")
        subprocess.call(["rocprofv3", "--pmc"] + CTRS + ["-d", d, "-o", "x", "--", "python3", os.path.join(R, "tools", "dev_rate_probe.py"), "-30", "8192"],
                        env=env2, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp")
        db = glob.glob(d + "/**/*.db", recursive=True)[0]
        c = sqlite3.connect(db)
        got = {}
        for k, cn, n, sm, du in c.execute("select kernel_name, counter_name, count(*), sum(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
            if "k_theil_sen" in k and "more" not in k:
                got[cn] = sm / n
                got["us"] = du / 1e3
        cum[stop] = got
    print("cumulative (the kernel leaves a row after the stage); per row: instructions; per launch of 8192 x 50 rows: time; quad-cycles of all waves in G")
    print("%-92s %8s %8s %7s %9s %8s %8s %8s %8s" % ("stage", "VALU", "SALU", "LDS", "us", "wave", "active", "w.issue", "w.cnt"))
    for stop in order:
        g = cum[stop]
        print("%-92s %8.1f %8.1f %7.1f %9.1f %8.3f %8.3f %8.3f %8.3f" % (names[stop], g["SQ_INSTS_VALU"] / rows, g["SQ_INSTS_SALU"] / rows, g["SQ_INSTS_LDS"] / rows, g["us"],
              g["SQ_WAVE_CYCLES"] / 1e9, g["SQ_ACTIVE_INST_ANY"] / 1e9, g["SQ_WAIT_INST_ANY"] / 1e9, g["SQ_WAIT_ANY"] / 1e9))
    print("\nper stage (differences):  instructions per row; us; cycles per instruction and wave (4 x wave quad-cycles / all instructions); share of the stage's wave cycles")
    print("%-92s %8s %8s %7s %9s %9s %8s %8s %8s" % ("stage", "VALU", "SALU", "LDS", "us", "cyc/inst", "active", "w.issue", "w.cnt"))
    prev = {k: 0.0 for k in CTRS + ["us"]}
    for stop in order:
        g = cum[stop]
        dv = {k: g[k] - prev[k] for k in prev}
        insts = dv["SQ_ACTIVE_INST_ANY"]        # one quad-cycle of "active" per instruction issued (calibrated on tools/ubench_issue.hip)
        wc = max(dv["SQ_WAVE_CYCLES"], 1.0)
        print("%-92s %8.1f %8.1f %7.1f %9.1f %9.2f %7.0f%% %7.0f%% %7.0f%%" % (names[stop].lstrip("+ "), dv["SQ_INSTS_VALU"] / rows, dv["SQ_INSTS_SALU"] / rows, dv["SQ_INSTS_LDS"] / rows,
              dv["us"], 4.0 * wc / max(insts, 1.0), 100 * dv["SQ_ACTIVE_INST_ANY"] / wc, 100 * dv["SQ_WAIT_INST_ANY"] / wc, 100 * dv["SQ_WAIT_ANY"] / wc))
        prev = dict(g)

if __name__ == "__main__":
    (build if sys.argv[1:] == ["build"] else run)()
