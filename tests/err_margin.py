#!/usr/bin/env python3
"""diagnostic (test infrastructure): how far the fp32 intermediates of the HIP path sit from the oracle's, per rate and
mode, against the 1e-5 tolerance of north_star (max |difference| relative to the largest magnitude of the array)"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle_lib as O, modem_amd
def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
for rate in (8000, 16000, 44100, 48000):
    rx = modem_amd.Receiver(device=0, chunk_frames=4, keep_raw_cons=True, sample_rate=rate)
    for mode, ch, freq, noise in ((6, 2, 2000, -30), (12, 2, -1000, -28), (12, 2, 1600, -28), (13, 2, 1000, -28), (10, 1, 1700, None), (9, 2, 1500, -25)):
        m = O.Mode(); O.lib().orc_mode_lookup(mode, C.byref(m))
        p = O.payload_for(77 + mode)
        pcm = O.encode_pcm(p, channels=ch, freq_off=freq, mode=mode, rate=rate)
        if noise is not None:
            pcm = O.impair(pcm, noise_db=noise, cfo_hz=12.5, seed=rate, frame=mode, rate=rate)
        out, res = rx.decode(pcm[None])
        oo, orr, tb = O.decode(pcm, taps=True, rate=rate)
        print(rate, mode, ch, "raw %.2e rot %.2e llr %.2e" % (rel(rx.tap("CONS_RAW", 0, cons_cnt=m.cons_cnt), tb.cons_raw[:m.cons_cnt]),
              rel(rx.tap("CONS_ROT", 0, cons_cnt=m.cons_cnt), tb.cons_rot[:m.cons_cnt]), rel(rx.tap("LLR", 0)[:m.cons_bits], tb.llr[:m.cons_bits])), flush=True)
    rx.close()
