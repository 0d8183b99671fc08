#!/usr/bin/env python3
"""dev_rate_probe.py [noise_db] [frames] -- frames/s of ofdmrx_decode_batch_device with the outputs left in HBM, and who finished the frames"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import modem_amd
import modem_amd.ofdmrx as M
db = float(sys.argv[1]) if len(sys.argv) > 1 else -20.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
rx = modem_amd.Receiver(device=0, stream=stream.cuda_stream)
spf = rx.tx_frame_samples(6)
g = torch.Generator(device=dev); g.manual_seed(1)
d_pay = torch.randint(0, 256, (n, 5380), dtype=torch.uint8, device=dev, generator=g)
d_in = torch.empty((n, spf, 2), dtype=torch.int16, device=dev)
rx.tx_encode(d_pay.data_ptr(), n, d_in.data_ptr())
rx.awgn_tile(d_in.data_ptr(), n, d_in.data_ptr(), n, spf, db, 7, 0)
rx.synchronize()
d_out = torch.zeros((n, 5380), dtype=torch.uint8, device=dev)
d_res = torch.zeros((n, M.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
for it in range(4):
    torch.cuda.synchronize()
    t = time.perf_counter()
    rx.decode_device(d_in.data_ptr(), M.FMT_S16, 2, spf, spf * 4, n, d_out.data_ptr(), d_res.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    ok = int((d_out == d_pay).all(dim=1).sum().item())
    print("%.1f dB: %.0f frames/s, payloads ok %d of %d, sc %d listed %d, sc stage %.1f ms" % (db, n / dt, ok, n, rx.sc_decided_frames(), rx.list_decoded_frames(), rx.sc_timing()[0]), flush=True)
