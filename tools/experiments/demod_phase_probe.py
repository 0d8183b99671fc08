#!/usr/bin/env python3
"""demod_phase_probe.py build | run -- what each phase of k_demod's symbol loop costs: TIMING variants (wrong results on purpose: the kernel has
no data-dependent control flow) that leave out the loader's arithmetic (1), the transform (2), the carrier / division / store phase (4),
the barriers (8: all three; 16: the third only) - patched into a copy of k_demod.hip, one 8192-frame chunk each under rocprofv3."""
import os, sqlite3, subprocess, sys, glob
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(R, "modem_amd", "csrc")
VAR = os.path.join(R, "modem_amd", "lib", "variants")
masks = [0, 1, 2, 4, 3, 6, 7, 16, 8]
names = {0: "the product kernel", 1: "without the loader's arithmetic (NCO, radix-5, twiddles: zeros into the rows)", 2: "without the transform",
         4: "without the carrier phase (no division, no store)", 3: "without loader arithmetic and transform", 6: "without transform and carrier phase",
         7: "loads, row writes and barriers only", 16: "without the third barrier (races: timing only)", 8: "without any workgroup barrier (races: timing only)"}

def patch(s):
    def once(old, new):
        assert s.count(old) == 1, old[:60]
        return s.replace(old, new)
    s = s.replace('namespace rx {\n', 'namespace rx {\n#ifndef DM_SKIP\n#define DM_SKIP 0\n#endif\n', 1)
    s = once('\t\t\t\t\tBfly<R1>::run(v);\n', '\t\t\t\t\tif (!(DM_SKIP & 1)) Bfly<R1>::run(v);\n')
    s = once('\t\t\t\t\t\t\tv[a] = cmul(pre[q][a], (QA_LDS && a) ? cmul(p0, rotA[a]) : qa[a]);', '\t\t\t\t\t\t\tv[a] = (DM_SKIP & 1) ? pre[q][a] : cmul(pre[q][a], (QA_LDS && a) ? cmul(p0, rotA[a]) : qa[a]);')
    s = once('\t\t\t\t\t\trow[r * NS + npw] = cmul(v[r], DC::TWR_LDS', '\t\t\t\t\t\trow[r * NS + npw] = (DM_SKIP & 1) ? v[r] : cmul(v[r], DC::TWR_LDS')
    s = once('\t\t\t\tfft256_regs(row + wave * NS, twl, lane, swz256(lane));', '\t\t\t\tif (!(DM_SKIP & 2)) fft256_regs(row + wave * NS, twl, lane, swz256(lane));')
    s = once('\t\t\t\tif (coff[e] >= 0) {\n\t\t\t\t\tconst cf cur = cmul(row[coff[e]], w);', '\t\t\t\tif (coff[e] >= 0 && !(DM_SKIP & 4)) {\n\t\t\t\t\tconst cf cur = cmul(row[coff[e]], w);')
    # barriers of the symbol loop: after the rows are written, after the transform, after the carriers
    s = once('\t\t\tif (AHEAD)\n\t\t\t\tfetch(s + 1);\n\t\t\t__syncthreads();', '\t\t\tif (AHEAD)\n\t\t\t\tfetch(s + 1);\n\t\t\tif (!(DM_SKIP & 8)) __syncthreads();')
    s = once('\t\t\t\t\tmono_part1();                                 // the next symbol\'s span (its samples arrived during the transform)\n\t\t\t__syncthreads();',
             '\t\t\t\t\tmono_part1();                                 // the next symbol\'s span (its samples arrived during the transform)\n\t\t\tif (!(DM_SKIP & 8)) __syncthreads();')
    s = once('\t\t\t\t\t\tmono_raw(s + 2);\n\t\t\t\t}\n\t\t\t__syncthreads();', '\t\t\t\t\t\tmono_raw(s + 2);\n\t\t\t\t}\n\t\t\tif (!(DM_SKIP & 24)) __syncthreads();')
    return s

def build():
    src = open(os.path.join(CS, "k_demod.hip")).read()
    os.makedirs("/tmp/dm_probe", exist_ok=True)
    os.makedirs(VAR, exist_ok=True)
    open("/tmp/dm_probe/k_demod_probe.hip", "w").write(patch(src))
    subprocess.check_call(["make", "-C", CS, "-j8", "all"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = [os.path.join(CS, f) for f in os.listdir(CS) if f.endswith(".o") and f not in ("k_demod.o", "decode_main.o", "encode_main.o")]
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-w", "-fno-slp-vectorize", "-DDM_SKIP=%d" % m, "-I" + CS, "-c",
                               "/tmp/dm_probe/k_demod_probe.hip", "-o", "/tmp/dm_probe/k_demod_%d.o" % m]) for m in masks]
    for p in procs:
        assert p.wait() == 0
    for m in masks:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(VAR, "libofdmrx_dmskip%d.so" % m),
                               "/tmp/dm_probe/k_demod_%d.o" % m] + objs)
    print("built", VAR)

def run():
    print("%-92s %9s %9s %9s %9s" % ("variant", "us", "VALU/frm", "LDS/frm", "parked"))
    for m in masks:
        d = "/tmp/dm_probe/pmc%d" % m
        subprocess.call(["rm", "-rf", d])
        env2 = dict(os.environ, MODEM_AMD_LIB=os.path.join(VAR, "libofdmrx_dmskip%d.so" % m), OFDMRX_NO_OVERLAP="1", TMPDIR="/tmp")
        subprocess.call(["rocprofv3", "--pmc", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "-d", d, "-o", "x", "--", "python3", os.path.join(R, "tools", "dev_rate_probe.py"), "-30", "8192"],
                        env=env2, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp")
        db = glob.glob(d + "/**/*.db", recursive=True)[0]
        g = {}
        for k, cn, n, sm, du in sqlite3.connect(db).execute("select kernel_name, counter_name, count(*), sum(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
            if "k_demod" in k:
                g[cn] = sm / n
                g["us"] = du / 1e3
        print("%-92s %9.1f %9.0f %9.0f %8.0f%%" % (names[m], g["us"], g["SQ_INSTS_VALU"] / 8192, g["SQ_INSTS_LDS"] / 8192, 100 * g["SQ_WAIT_ANY"] / g["SQ_WAVE_CYCLES"]), flush=True)

if __name__ == "__main__":
    (build if sys.argv[1:] == ["build"] else run)()
