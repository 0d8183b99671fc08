#!/usr/bin/env python3
"""lanes_trace.py -- per call: wall time and whether the call was cut across two lanes, pinned host outputs, -20 dB"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import modem_amd
import modem_amd.ofdmrx as M
dev = torch.device("cuda:0")
n = 65536
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
rx = modem_amd.Receiver(device=0, stream=stream.cuda_stream)
spf = rx.tx_frame_samples(6)
g = torch.Generator(device=dev); g.manual_seed(1)
d_pay = torch.randint(0, 256, (n, 5380), dtype=torch.uint8, device=dev, generator=g)
d_in = torch.empty((n, spf, 2), dtype=torch.int16, device=dev)
rx.tx_encode(d_pay.data_ptr(), n, d_in.data_ptr())
rx.awgn_tile(d_in.data_ptr(), n, d_in.data_ptr(), n, spf, -20.0, 7, 0)
rx.synchronize()
h_out = torch.empty((n, 5380), dtype=torch.uint8, pin_memory=True)
h_res = torch.empty((n, M.RESULT_DTYPE.itemsize), dtype=torch.uint8, pin_memory=True)
d_out = torch.zeros((n, 5380), dtype=torch.uint8, device=dev)
d_res = torch.zeros((n, M.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
for mode in ("host", "hbm", "host"):
    for it in range(6):
        torch.cuda.synchronize()
        t = time.perf_counter()
        if mode == "host":
            rx.decode_device(d_in.data_ptr(), M.FMT_S16, 2, spf, spf * 4, n, h_out.data_ptr(), h_res.data_ptr())
        else:
            rx.decode_device(d_in.data_ptr(), M.FMT_S16, 2, spf, spf * 4, n, d_out.data_ptr(), d_res.data_ptr())
        t_enq = time.perf_counter() - t
        if os.environ.get("SYNC_VIA") == "timing":
            rx.timing()
        else:
            torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print("%s call %d: %.1f ms (enqueue %.1f ms)  %.0f frames/s  last chunk starts at %d" % (mode, it, dt * 1e3, t_enq * 1e3, n / dt, rx.last_chunk_first_frame()), flush=True)
