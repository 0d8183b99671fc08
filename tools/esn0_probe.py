#!/usr/bin/env python3
"""esn0_probe.py -- the cumulative Es/N0 estimate of the last row (decode.cc:517) against who finishes the frame, per noise level"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import modem_amd, modem_amd.ofdmrx as M
dev = torch.device("cuda:0")
n = 4096
rx = modem_amd.Receiver(device=0, chunk_frames=4096, scl_always=False)
rk = modem_amd.Receiver(device=0, chunk_frames=64, keep_raw_cons=True)
spf = rx.tx_frame_samples(6)
g = torch.Generator(device=dev); g.manual_seed(1)
d_pay = torch.randint(0, 256, (n, 5380), dtype=torch.uint8, device=dev, generator=g)
d_clean = torch.empty((n, spf, 2), dtype=torch.int16, device=dev)
rx.tx_encode(d_pay.data_ptr(), n, d_clean.data_ptr())
d_imp = torch.empty_like(d_clean)
rx.channel(d_clean.data_ptr(), d_imp.data_ptr(), n, spf, cfo_hz=234.567, sfo_ppm=147.0, multipath=[(0, 1 + 0j), (5, 0.35 - 0.1j), (11, -0.2 + 0.2j), (23, 0.1 + 0.05j)])
d_in = torch.empty_like(d_clean)
d_out = torch.zeros((n, 5380), dtype=torch.uint8, device=dev)
d_res = torch.zeros((n, M.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
for base, name, levels in ((d_clean, "awgn", (-24, -20, -19, -18.6, -18.4, -18.2, -18.0, -17.5, -16)), (d_imp, "chain", (-30, -24, -21, -20, -19, -18))):
    for db in levels:
        rx.awgn_tile(base.data_ptr(), n, d_in.data_ptr(), n, spf, float(db), 7, 0)
        rx.decode_device(d_in.data_ptr(), M.FMT_S16, 2, spf, spf * 4, n, d_out.data_ptr(), d_res.data_ptr())
        rx.synchronize()
        res = d_res.cpu().numpy().view(M.RESULT_DTYPE).reshape(-1)
        e = res["esn0_db_last"]
        print("%s %6.1f dB: esn0_last mean %.3f min %.3f max %.3f | sc %d listed %d ok %d" % (name, db, e.mean(), e.min(), e.max(), rx.sc_decided_frames(), rx.list_decoded_frames(), int((res["status"] == 0).sum())), flush=True)
