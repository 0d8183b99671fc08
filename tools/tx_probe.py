#!/usr/bin/env python3
"""Transmitter alone: milliseconds per 8192 mode-6 frames of ofdmrx_tx_encode (random payloads resident in HBM -> int16 PCM in HBM)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import modem_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
rx = modem_amd.Receiver(device=0, stream=stream.cuda_stream)
spf = rx.tx_frame_samples(6)
pay = torch.randint(0, 256, (n, 5380), dtype=torch.uint8, device=dev)
pcm = torch.empty((n, spf, 2), dtype=torch.int16, device=dev)
for rep in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter()
    rx.tx_encode(pay.data_ptr(), n, pcm.data_ptr(), mode=6, freq_off=2000, call_sign="ANONYMOUS", channels=2)
    torch.cuda.synchronize()
    print("tx_encode %d frames: %.2f ms" % (n, 1e3 * (time.perf_counter() - t)), flush=True)
rx.close()
