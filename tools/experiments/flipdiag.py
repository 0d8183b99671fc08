#!/usr/bin/env python3
"""why does bit_flips differ for one noisy 48 kHz frame? compare the per-row slope / intercept / precision taps and the LLRs with the oracle's"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as O, modem_amd, modem_amd.ofdmrx as M
rate, n, li, db, fr = 48000, 64, 2, -8.0, 6
dev = torch.device("cuda:0"); stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
rx = modem_amd.Receiver(device=0, chunk_frames=96, stream=stream.cuda_stream, sample_rate=rate, keep_raw_cons=True)
spf = rx.tx_frame_samples(6)
g = torch.Generator(device=dev); g.manual_seed(1234 + li)
d_pay = torch.randint(0, 256, (n, 5380), dtype=torch.uint8, device=dev, generator=g)
d_in = torch.empty((n, spf, 2), dtype=torch.int16, device=dev)
rx.tx_encode(d_pay.data_ptr(), n, d_in.data_ptr(), mode=6)
rx.awgn_tile(d_in.data_ptr(), n, d_in.data_ptr(), n, spf, db, 99, li * n)
rx.synchronize()
pcm = np.ascontiguousarray(d_in[fr].cpu().numpy())
out, res = rx.decode(pcm[None])
oout, ores, tb = O.decode(pcm, taps=True, rate=rate)
print("flips gpu", int(res[0]["bit_flips"]), "oracle", ores.bit_flips)
sl, yi, pr = rx.tap("SLOPE", 0, rows=50), rx.tap("YINT", 0, rows=50), rx.tap("PRECISION", 0, rows=50)
for name, a, b in (("slope", sl, tb.slope[:50]), ("yint", yi, tb.yint[:50]), ("precision", pr, tb.precision[:50])):
    d = np.abs(a.astype(np.float64) - b)
    print(name, "max abs diff %.3g at row %d (value %.6g), rows differing at all: %d" % (d.max(), int(d.argmax()), b[int(d.argmax())], int((a != b).sum())))
llr = rx.tap("LLR", 0)[:64800]; ol = tb.llr[:64800]
dl = np.abs(llr - ol)
print("llr max abs diff %.3g, sign differences %d, of which |oracle llr| < 1e-3: %d" % (dl.max(), int((np.signbit(llr) != np.signbit(ol)).sum()),
      int(((np.signbit(llr) != np.signbit(ol)) & (np.abs(ol) < 1e-3)).sum())))
raw = rx.tap("CONS_RAW", 0, cons_cnt=21600); d = np.abs(raw - tb.cons_raw[:21600]); print("cons_raw max abs diff %.3g" % d.max())
rot = rx.tap("CONS_ROT", 0, cons_cnt=21600); d = np.abs(rot - tb.cons_rot[:21600]); print("cons_rot max abs diff %.3g at %d" % (d.max(), int(d.argmax()) // 432))
