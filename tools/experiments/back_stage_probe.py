#!/usr/bin/env python3
"""back_stage_probe.py build | run -- where k_back spends its time on a certified frame, stage by stage (like ts_stage_probe.py: variants
of k_finish.hip that leave a frame after stage 1..4, patched into a copy; one 8192-frame chunk under rocprofv3 --pmc each)."""
import os, sqlite3, subprocess, sys, glob
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(R, "modem_amd", "csrc")
VAR = os.path.join(R, "modem_amd", "lib", "variants")
names = {1: "rows: rotation, SNR sums, sign bits (snr_rows)", 2: "+ systematic message gathered from the sign bits", 3: "+ u = x F in place, syndrome",
         4: "+ CRC-32", 0: "+ payload, record = the whole kernel"}
order = [1, 2, 3, 4, 0]
CTRS = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS"]

def patch(s):
    def once(old, new):
        assert s.count(old) == 1, old[:60]
        return s.replace(old, new)
    s = s.replace('namespace rx {\n', 'namespace rx {\n#ifndef BK_STOP\n#define BK_STOP 0\n#endif\n', 1)
    s = once('\todd |= !snr_ok;', '\todd |= !snr_ok;\n\tif (BK_STOP == 1) { if (tid == 0) res_all[f].bit_flips = (int)bits[0] + (int)prec[0]; return; }')
    s = once('\t\t// ---- 2. u = x F: at every level', '\t\tif (BK_STOP == 2) { __syncthreads(); if (tid == 0) res_all[f].bit_flips = mesg[0]; return; }\n\t\t// ---- 2. u = x F: at every level')
    s = once('\t\tbad = __syncthreads_or((syn != 0) | (odd ? 1 : 0));', '\t\tbad = __syncthreads_or((syn != 0) | (odd ? 1 : 0));\n\t\tif (BK_STOP == 3) { if (tid == 0) res_all[f].bit_flips = bad; return; }')
    s = once('\t\t\tbad = crc_sh != 0;\n\t\t}', '\t\t\tbad = crc_sh != 0;\n\t\t}\n\t\tif (BK_STOP == 4) { if (tid == 0) res_all[f].bit_flips = bad; return; }')
    return s

def build():
    src = open(os.path.join(CS, "k_finish.hip")).read()
    os.makedirs("/tmp/bk_probe", exist_ok=True)
    os.makedirs(VAR, exist_ok=True)
    open("/tmp/bk_probe/k_finish_probe.hip", "w").write(patch(src))
    subprocess.check_call(["make", "-C", CS, "-j8", "all"], stdout=subprocess.DEVNULL)
    objs = [os.path.join(CS, f) for f in os.listdir(CS) if f.endswith(".o") and f not in ("k_finish.o", "decode_main.o", "encode_main.o")]
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-w", "-DBK_STOP=%d" % st, "-I" + CS, "-c",
                               "/tmp/bk_probe/k_finish_probe.hip", "-o", "/tmp/bk_probe/k_finish_%d.o" % st]) for st in order]
    for p in procs:
        assert p.wait() == 0
    for st in order:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(VAR, "libofdmrx_bkstop%d.so" % st),
                               "/tmp/bk_probe/k_finish_%d.o" % st] + objs)
    print("built", VAR)

def run():
    cum = {}
    for st in order:
        d = "/tmp/bk_probe/pmc%d" % st
        subprocess.call(["rm", "-rf", d])
        env2 = dict(os.environ, MODEM_AMD_LIB=os.path.join(VAR, "libofdmrx_bkstop%d.so" % st), OFDMRX_NO_OVERLAP="1", TMPDIR="/tmp")
        subprocess.call(["rocprofv3", "--pmc"] + CTRS + ["-d", d, "-o", "x", "--", "python3", os.path.join(R, "tools", "dev_rate_probe.py"), "-30", "8192"],
                        env=env2, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp")
        db = glob.glob(d + "/**/*.db", recursive=True)[0]
        got = {}
        for k, cn, n, sm, du in sqlite3.connect(db).execute("select kernel_name, counter_name, count(*), sum(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
            if "k_back" in k:
                got[cn] = sm / n
                got["us"] = du / 1e3
        cum[st] = got
    print("per stage (differences between variants that leave a frame after the stage): instructions per frame; us per launch of 8192 frames; share of the stage's wave cycles")
    print("%-70s %8s %8s %8s %9s %8s %8s %8s %10s" % ("stage", "VALU", "SALU", "LDS", "us", "active", "w.issue", "parked", "LDS busy"))
    prev = {k: 0.0 for k in CTRS + ["us"]}
    for st in order:
        g = cum[st]
        dv = {k: g[k] - prev[k] for k in prev}
        wc = max(dv["SQ_WAVE_CYCLES"], 1.0)
        print("%-70s %8.0f %8.0f %8.0f %9.1f %7.0f%% %7.0f%% %7.0f%% %9.0f%%" % (names[st].lstrip("+ "), dv["SQ_INSTS_VALU"] / 8192, dv["SQ_INSTS_SALU"] / 8192, dv["SQ_INSTS_LDS"] / 8192, dv["us"],
              100 * dv["SQ_ACTIVE_INST_ANY"] / wc, 100 * dv["SQ_WAIT_INST_ANY"] / wc, 100 * dv["SQ_WAIT_ANY"] / wc,
              100 * 4 * dv["SQ_ACTIVE_INST_LDS"] / 256 / max(dv["us"] * 2300.0, 1.0)))
        prev = dict(g)

if __name__ == "__main__":
    (build if sys.argv[1:] == ["build"] else run)()
