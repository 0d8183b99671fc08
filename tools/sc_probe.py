"""k_sc against the oracle's restatement of the sign-following path (orc_polar_sc_path), then the default pipeline at a noise
level where every frame has raw bit errors.  Run on the GPU box: python3 tools/sc_probe.py [frames_per_level]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import oracle_lib as O
import modem_amd

per = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rx = modem_amd.Receiver(device=0, chunk_frames=64)
fr = O.frozen(0)
base = O.encode_pcm(O.payload_for(3), channels=2)
llrs, want = [], []
for db in (-30, -24, -20, -19, -18, -16):
    for f in range(per):
        pcm = O.impair(base, noise_db=db, seed=7, frame=f + int(-db * 10))
        out, r, tb = O.decode(pcm, taps=True)
        llrs.append(tb.llr.copy())
        want.append((db,) + O.polar_sc_path(tb.llr, fr) + (float(tb.metric[0]), r.best_lane))
llr = np.stack(llrs)
llr[-1][5] = 0.0
want[-1] = (want[-1][0],) + O.polar_sc_path(llr[-1], fr) + want[-1][4:]
t0 = time.time()
cw, hd, M, F, ok = rx.sc_path(llr)
print("gpu sc_path: %.2f s for %d codewords" % (time.time() - t0, len(llr)))
bad = 0
for i, (db, c, m, f, m0, lane) in enumerate(want):
    same = (cw[i] == c).all() and (hd[i] == (llr[i] < 0)).all() and M[i] == m and F[i] == f and bool(ok[i]) == bool(f > m)
    print(db, "codeword", int((cw[i] != c).sum()), "hard", int((hd[i] != (llr[i] < 0)).sum()), "M", M[i], m, "fork", F[i], f, "ok", ok[i], "list metric0", m0, "OK" if same else "MISMATCH")
    bad += not same
print("mismatches:", bad)
