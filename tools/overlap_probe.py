#!/usr/bin/env python3
"""experiment: do two independent receive pipelines on two HIP streams overlap usefully on one GPU?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import modem_amd
import modem_amd.ofdmrx as M
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
s0 = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(s0)
rx0 = modem_amd.Receiver(device=0, stream=s0.cuda_stream, chunk_frames=chunk)
spf = rx0.tx_frame_samples(6)
d_pay = torch.randint(0, 256, (n, 5380), dtype=torch.uint8, device=dev)
d_in = torch.empty((n, spf, 2), dtype=torch.int16, device=dev)
rx0.tx_encode(d_pay.data_ptr(), n, d_in.data_ptr())
rx0.awgn_tile(d_in.data_ptr(), n, d_in.data_ptr(), n, spf, -30.0, 1, 0)
d_out = torch.zeros((n, 5380), dtype=torch.uint8, device=dev)
d_res = torch.zeros((n, M.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
rx0.synchronize()
def run_single():
    t = time.perf_counter()
    rx0.decode_device(d_in.data_ptr(), M.FMT_S16, 2, spf, spf * 4, n, d_out.data_ptr(), d_res.data_ptr())
    rx0.synchronize()
    return time.perf_counter() - t
run_single()
t1 = min(run_single() for _ in range(2))
print("single pipeline: %.1f ms  %.0f frames/s  ok=%s" % (t1 * 1e3, n / t1, bool((d_out == d_pay).all())))
s1 = torch.cuda.Stream(device=dev)
rx1 = modem_amd.Receiver(device=0, stream=s1.cuda_stream, chunk_frames=chunk)
h = n // 2
def run_dual():
    t = time.perf_counter()
    rx0.decode_device(d_in.data_ptr(), M.FMT_S16, 2, spf, spf * 4, h, d_out.data_ptr(), d_res.data_ptr())
    rx1.decode_device(d_in[h:].data_ptr(), M.FMT_S16, 2, spf, spf * 4, n - h, d_out[h:].data_ptr(), d_res[h:].data_ptr())
    rx0.synchronize(); rx1.synchronize()
    return time.perf_counter() - t
d_out.zero_()
run_dual()
t2 = min(run_dual() for _ in range(2))
print("two pipelines  : %.1f ms  %.0f frames/s  ok=%s" % (t2 * 1e3, n / t2, bool((d_out == d_pay).all())))
