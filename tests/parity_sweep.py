#!/usr/bin/env python3
"""GPU path vs the CPU oracle frame by frame near the waterfall, where every slow path of the list decoder runs
(failed node shortcuts, forks with path replacement, CRC failures): status, payload, winning lane, flips.
(test infrastructure: the only place outside tests/*.py, smoke() and bench.py's cpu_baseline that runs the oracle)
usage: python tests/parity_sweep.py [frames per level] [levels dB ...]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import oracle_lib as O
import modem_amd
import modem_amd.ofdmrx as M

mode = int(os.environ.get("SWEEP_MODE", "6"))
rate = int(os.environ.get("SWEEP_RATE", "8000"))
channels = int(os.environ.get("SWEEP_CHANNELS", "2"))   # 1: the real part of the noisy analytic stream + a DC offset (SWEEP_DC, LSB): mono input
dc = int(os.environ.get("SWEEP_DC", "700"))
chain = int(os.environ.get("SWEEP_CHAIN", "0"))         # 1: configs[3]'s chain in front of the noise - multipath (4 taps) -> CFO +234.567 Hz -> SFO +147 ppm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
levels = [float(x) for x in sys.argv[2:]] or [-17.0, -15.5, -15.0, -14.5]
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
rx = modem_amd.Receiver(device=0, chunk_frames=96, stream=stream.cuda_stream, sample_rate=rate)
spf = rx.tx_frame_samples(mode)
O.lib().orc_decode_rate   # (loads the library)
print("mode %d, %d Hz, %d channel%s%s" % (mode, rate, channels, "s" if channels == 2 else " (DC offset %d LSB)" % dc,
                                          ", multipath (4 taps) -> CFO +234.567 Hz -> SFO +147 ppm in front of the noise" if chain else ""), flush=True)
threads = min(os.cpu_count() or 1, int(os.environ.get("SWEEP_THREADS", "32")))
bad = 0
bad_decisions = 0
for li, db in enumerate(levels):
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + li)
    d_pay = torch.randint(0, 256, (n, 5380), dtype=torch.uint8, device=dev, generator=g)
    d_in = torch.empty((n, spf, 2), dtype=torch.int16, device=dev)
    rx.tx_encode(d_pay.data_ptr(), n, d_in.data_ptr(), mode=mode)
    if chain:
        d_imp = torch.empty_like(d_in)
        for lo in range(0, n, 8192):
            hi = min(lo + 8192, n)
            rx.channel(d_in[lo:hi].data_ptr(), d_imp[lo:hi].data_ptr(), hi - lo, spf, cfo_hz=234.567, sfo_ppm=147.0,
                       multipath=[(0, 1 + 0j), (5, 0.35 - 0.1j), (11, -0.2 + 0.2j), (23, 0.1 + 0.05j)])
        d_in = d_imp
    rx.awgn_tile(d_in.data_ptr(), n, d_in.data_ptr(), n, spf, db, 99, li * n)
    rx.synchronize()
    if channels == 1:
        d_in = torch.clamp(d_in[:, :, 0].to(torch.int32) + dc, -32768, 32767).to(torch.int16).contiguous()
        torch.cuda.synchronize()
    d_out = torch.zeros((n, 5380), dtype=torch.uint8, device=dev)
    d_res = torch.zeros((n, M.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    rx.decode_device(d_in.data_ptr(), M.FMT_S16, channels, spf, spf * 2 * channels, n, d_out.data_ptr(), d_res.data_ptr())
    rx.synchronize()
    out = d_out.cpu().numpy()
    res = d_res.cpu().numpy().view(M.RESULT_DTYPE).reshape(-1)
    n_list, n_sc = rx.list_decoded_frames(), rx.sc_decided_frames()
    pcm = np.ascontiguousarray(d_in.cpu().numpy())
    oout = np.zeros((n, 5380), np.uint8)
    ores = np.zeros(n * 56, np.uint8)
    t = time.perf_counter()
    if rate == 8000:
        O.lib().orc_decode_batch(O.ptr(pcm), O.FMT_S16, channels, spf, spf * 2 * channels, n, 8, O.ptr(oout), O.ptr(ores), threads)
    else:   # the batch helper is the 8 kHz instantiation: other rates frame by frame
        ov = ores.view(M.RESULT_DTYPE).reshape(-1)
        for f in range(n):
            o, r = O.decode(pcm[f] if channels == 2 else pcm[f][:, None], rate=rate)
            oout[f] = o
            for name in ov.dtype.names:
                ov[name][f] = getattr(r, name)
    dt = time.perf_counter() - t
    ores = ores.view(M.RESULT_DTYPE).reshape(-1)
    # bit_flips (decode.cc:546-555) counts sign(llr) != decoded bit: an LLR within the 1e-5 intermediate tolerance
    # of zero may carry either sign, so that diagnostic may differ by a count or two; everything decided must not
    same = (out == oout).all(axis=1) & (res["status"] == ores["status"]) & (res["best_lane"] == ores["best_lane"]) \
        & (np.abs(res["bit_flips"].astype(np.int64) - ores["bit_flips"]) <= 2) & (res["sc_start"] == ores["sc_start"]) \
        & (res["symbol_pos"] == ores["symbol_pos"]) & (res["oper_mode"] == ores["oper_mode"]) & (res["call_sign"] == ores["call_sign"])
    decided = (out == oout).all(axis=1) & (res["status"] == ores["status"]) & (res["best_lane"] == ores["best_lane"]) \
        & (res["sc_start"] == ores["sc_start"]) & (res["symbol_pos"] == ores["symbol_pos"]) & (res["oper_mode"] == ores["oper_mode"]) \
        & (res["call_sign"] == ores["call_sign"])                 # everything but the flip-count diagnostic
    worst_flips = int(np.abs(res["bit_flips"].astype(np.int64) - ores["bit_flips"]).max())
    bad_decisions += int((~decided).sum())
    flips_equal = int((res["bit_flips"] == ores["bit_flips"]).sum())
    nok = int((res["status"] == 0).sum())
    bad += int((~same).sum())
    print("%6.1f dB: %d frames, %d decoded, GPU == oracle (payload, status, lane, sync, header) on %d, flip count identical on %d, within 2 on %d, "
          "largest difference %d (oracle %.1f s on %d threads; GPU routes: list-1 pass %d, list decoder %d)"
          % (db, n, nok, int(decided.sum()), flips_equal, int(same.sum()), worst_flips, dt, threads, n_sc, n_list), flush=True)
    if not same.all():
        i = int(np.argmin(same))
        differ = [nm for nm in ("status", "best_lane", "sc_start", "symbol_pos", "oper_mode", "call_sign", "n_sync_rejects") if res[nm][i] != ores[nm][i]]
        if not (out[i] == oout[i]).all():
            differ.append("payload")
        print("   first mismatch frame %d: gpu status %d lane %d flips %d | oracle status %d lane %d flips %d | fields that differ: %s; "
              "cfo gpu %.9g oracle %.9g" % (i, res["status"][i], res["best_lane"][i], res["bit_flips"][i], ores["status"][i],
              ores["best_lane"][i], ores["bit_flips"][i], ", ".join(differ) or "none but the flip count", res["cfo_rad"][i], ores["cfo_rad"][i]))
    for i in np.nonzero(~decided)[0][:16]:                     # every frame that differs in something decided, field by field
        differ = [nm for nm in ("status", "best_lane", "sc_start", "symbol_pos", "oper_mode", "call_sign", "n_sync_rejects") if res[nm][i] != ores[nm][i]]
        if not (out[i] == oout[i]).all():
            differ.append("payload")
        print("   decided differently, frame %d: %s | gpu status %d lane %d sc_start %d symbol_pos %d | oracle status %d lane %d sc_start %d symbol_pos %d"
              % (i, ", ".join(differ), res["status"][i], res["best_lane"][i], res["sc_start"][i], res["symbol_pos"][i],
                 ores["status"][i], ores["best_lane"][i], ores["sc_start"][i], ores["symbol_pos"][i]), flush=True)
print("mismatches: %d in what is decided, %d more in the flip-count diagnostic (beyond +-2)" % (bad_decisions, bad - bad_decisions))
sys.exit(1 if bad else 0)
