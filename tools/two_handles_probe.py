#!/usr/bin/env python3
"""two_handles_probe.py -- does the machine have scheduling slack left?  The same 65536 frames decoded by ONE handle and by
TWO handles (halves of the batch, independent pipelines on their own streams, so the exclusive front phase of one runs
beside the shared phase of the other)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import modem_amd
import modem_amd.ofdmrx as M

dev = torch.device("cuda:0")
n = 65536
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
torch.cuda.set_stream(streams[0])
rx = [modem_amd.Receiver(device=0, stream=s.cuda_stream) for s in streams]
spf = rx[0].tx_frame_samples(6)
g = torch.Generator(device=dev); g.manual_seed(1)
d_pay = torch.randint(0, 256, (n, 5380), dtype=torch.uint8, device=dev, generator=g)
d_in = torch.empty((n, spf, 2), dtype=torch.int16, device=dev)
rx[0].tx_encode(d_pay.data_ptr(), n, d_in.data_ptr())
rx[0].awgn_tile(d_in.data_ptr(), n, d_in.data_ptr(), n, spf, float(os.environ.get("PROBE_NOISE_DB", "-30")), 7, 0)
rx[0].synchronize()
d_out = torch.zeros((n, 5380), dtype=torch.uint8, device=dev)
d_res = torch.zeros((n, M.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)

def run(parts):
    per = n // parts
    torch.cuda.synchronize()
    t = time.perf_counter()
    for q in range(parts):
        lo = q * per
        rx[q].decode_device(d_in[lo:].data_ptr(), M.FMT_S16, 2, spf, spf * 4, per, d_out[lo:].data_ptr(), d_res[lo:].data_ptr())
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t)

for parts in (1, 2, 1, 2, 3, 3):
    r = run(parts)
    ok = bool((d_out[: (n // parts) * parts] == d_pay[: (n // parts) * parts]).all())
    print("handles %d: %.0f frames/s, payloads ok %s" % (parts, r, ok), flush=True)
