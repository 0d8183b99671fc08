#!/usr/bin/env python3
"""flips_probe.py -- how many channel hard decisions disagree with the decoded codeword (decode.cc:546-555), per noise level:
the share of frames with none tells how often the raw hard decisions already ARE the codeword."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import modem_amd
import modem_amd.ofdmrx as M

dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
rx = modem_amd.Receiver(device=0, stream=stream.cuda_stream)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
spf = rx.tx_frame_samples(6)
d_pay = torch.randint(0, 256, (n, 5380), dtype=torch.uint8, device=dev)
d_clean = torch.empty((n, spf, 2), dtype=torch.int16, device=dev)
d_in = torch.empty_like(d_clean)
d_out = torch.zeros((n, 5380), dtype=torch.uint8, device=dev)
d_res = torch.zeros((n, M.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
rx.tx_encode(d_pay.data_ptr(), n, d_clean.data_ptr(), mode=6, freq_off=2000, call_sign="ANONYMOUS", channels=2)
for db in (-40, -30, -26, -24, -22, -20, -18, -16, -15):
    rx.awgn_tile(d_clean.data_ptr(), n, d_in.data_ptr(), n, spf, float(db), 99, 0)
    rx.decode_device(d_in.data_ptr(), M.FMT_S16, 2, spf, spf * 4, n, d_out.data_ptr(), d_res.data_ptr())
    torch.cuda.synchronize()
    res = d_res.cpu().numpy().view(M.RESULT_DTYPE).reshape(-1)
    ok = res["status"] == 0
    fl = res["bit_flips"][ok]
    print("noise %4d dB: decoded %5d of %d | bit_flips mean %.1f median %d max %d | frames with 0 flips %.3f, best_lane 0 in %.4f"
          % (db, ok.sum(), n, fl.mean() if fl.size else -1, np.median(fl) if fl.size else -1, fl.max() if fl.size else -1,
             (fl == 0).mean() if fl.size else 0, (res["best_lane"][ok] == 0).mean() if ok.any() else 0))
