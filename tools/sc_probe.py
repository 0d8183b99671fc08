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

# the pipeline at -20 dB, 256 frames: outputs against a handle that list-decodes every frame
import torch
import modem_amd.ofdmrx as M
dev = torch.device("cuda:0")
n = 512
rxa = modem_amd.Receiver(device=0, chunk_frames=128)
rxs = modem_amd.Receiver(device=0, chunk_frames=128, scl_always=True)
spf = rxa.tx_frame_samples(6)
g = torch.Generator(device=dev); g.manual_seed(5)
d_pay = torch.randint(0, 256, (n, 5380), dtype=torch.uint8, device=dev, generator=g)
d_clean = torch.empty((n, spf, 2), dtype=torch.int16, device=dev)
rxa.tx_encode(d_pay.data_ptr(), n, d_clean.data_ptr()); rxa.synchronize()
d_in = torch.empty_like(d_clean)
for db in (-26.0, -20.0, -18.5, -17.0):
    rxa.awgn_tile(d_clean.data_ptr(), n, d_in.data_ptr(), n, spf, db, 3, 0); rxa.synchronize()
    outs = []
    for r in (rxa, rxs):
        d_out = torch.zeros((n, 5380), dtype=torch.uint8, device=dev)
        d_res = torch.zeros((n, M.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        r.decode_device(d_in.data_ptr(), M.FMT_S16, 2, spf, spf * 4, n, d_out.data_ptr(), d_res.data_ptr())
        r.synchronize()
        outs.append((d_out.cpu().numpy(), d_res.cpu().numpy().view(M.RESULT_DTYPE).reshape(-1)))
    (oa, ra), (ob, rb) = outs
    same = (oa == ob).all() and all(((ra[nm] == rb[nm]) | ((ra[nm] != ra[nm]) & (rb[nm] != rb[nm]))).all() for nm in ra.dtype.names)
    print(db, "sc_decided", rxa.sc_decided_frames(), "listed", rxa.list_decoded_frames(), "identical to the list decoder:", bool(same),
          "payload ok", int((oa == d_pay.cpu().numpy()).all(axis=1).sum()))
