#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry (ofdmrx_decode_batch): pageable numpy input, payloads back on the host"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import modem_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
fx = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "base_frames_2ch.npz"))
base, pay = fx["pcm"], fx["payload"]
batch = np.ascontiguousarray(base[np.arange(n) % base.shape[0]])
rx = modem_amd.Receiver(device=0)
out, res = rx.decode(batch[:256])
for rep in range(2):
    t = time.perf_counter()
    out, res = rx.decode(batch)
    dt = time.perf_counter() - t
    ok = int((res["status"] == 0).sum())
    print("host path: %d frames %.1f ms  %.0f frames/s  (%.1f GB/s in)  ok %d" % (n, dt * 1e3, n / dt, batch.nbytes / dt / 1e9, ok))
