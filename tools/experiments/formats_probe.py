"""Throughput of the receive path per INPUT FORMAT (int16 / float32 / 8-bit unsigned, 2-channel and mono) at one sample rate: the same
frames (device transmitter + AWGN at -30 dB), converted on the device, through ofdmrx_decode_batch_device; payloads against the
transmitted ones.  python3 tools/experiments/formats_probe.py [rate] [frames]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import modem_amd
import modem_amd.ofdmrx as M

rate = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
rx = modem_amd.Receiver(device=0, stream=stream.cuda_stream, sample_rate=rate)
spf = rx.tx_frame_samples(6)
g = torch.Generator(device=dev); g.manual_seed(3)
d_pay = torch.randint(0, 256, (n, 5380), dtype=torch.uint8, device=dev, generator=g)
d_s16 = torch.empty((n, spf, 2), dtype=torch.int16, device=dev)
rx.tx_encode(d_pay.data_ptr(), n, d_s16.data_ptr())
rx.awgn_tile(d_s16.data_ptr(), n, d_s16.data_ptr(), n, spf, -30.0, 5, 0)
rx.synchronize()
for ch in (2, 1):
    base = d_s16 if ch == 2 else d_s16[:, :, 0].contiguous()
    for name, fmt, conv in (("int16", M.FMT_S16, lambda t: t),
                            ("float32", M.FMT_F32, lambda t: (t.to(torch.float32) / 32767.0).contiguous()),
                            ("uint8", M.FMT_U8, lambda t: (torch.clamp(torch.round(t.to(torch.float32) / 32767.0 * 127.0), -128, 127) + 128).to(torch.uint8).contiguous())):
        d_in = conv(base)
        torch.cuda.synchronize()
        bps = d_in.element_size()
        d_out = torch.zeros((n, 5380), dtype=torch.uint8, device=dev)
        d_res = torch.zeros((n, M.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rx.decode_device(d_in.data_ptr(), fmt, ch, spf, spf * bps * ch, n, d_out.data_ptr(), d_res.data_ptr())
            rx.synchronize(); best = min(best, time.perf_counter() - t0)
        ok = int((d_out == d_pay).all(dim=1).sum().item())
        print("%d Hz %d ch %-8s %9.0f frames/s, payloads equal to the transmitted ones: %d of %d" % (rate, ch, name, n / best, ok, n), flush=True)
        del d_in
