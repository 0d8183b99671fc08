"""GPU == oracle sweeps by regime, run by the driver's `-m gpu` suite (round 6; until then tests/parity_sweep.py, by hand).

One sweep per way a frame can be finished - syndrome certificate, list decoder, the waterfall, mono input (the list-1 pass has
test_gpu_parity.py::test_sc_certificate_at_scale) - device-made frames through the DEFAULT handle against the oracle's full list-8
decoder on all host threads, frame by frame.  What must be identical: payload bytes, status, winning lane, sync position
(sc_start, symbol_pos), header fields, reject count.  What may differ, and how that is checked instead of waved through:

  * bit_flips (decode.cc:546-555) counts payload positions whose LLR SIGN disagrees with the decoded bit.  For every frame whose count
    differs the frame is decoded again through a handle with taps and through the oracle with taps, and every code position where
    the two LLR signs differ must be a tie of one of two kinds: a SIGN tie - both magnitudes within the north_star tolerance (1e-5 of
    the frame's largest soft value) of zero - or an ERASURE tie - decode.cc:232 erases a point whose |cons|^2 exceeds 4 (LLR exactly 0),
    one side did and the other did not, and the point's |cons|^2 is within 1e-5 of 4 - or, where frames have raw bit errors, a ROW tie - the
    row's Theil-Sen line (decode.cc:488, a median over hard-decided phases) differs between the two sides because a point of the row
    sits on a hard-decision boundary (psk.hh:118-123) or is erased on one side, which turns every point of that row by the same small
    angle: then the soft bits of that row may differ by up to 5e-3 of the largest.  The counts may differ by at most the number of such
    positions.  No numeric slack besides that.
  * at the waterfall / on mono input: the two documented tie classes of test_gpu_parity._tie_class (a nearbyint of the fine timing
    estimate on a rounding boundary; path metrics an ulp apart at the list's edge), counted and bounded per sweep.
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from parity_explain import CONS_BITS, NAMES, REL, _explain_flips
from test_gpu_parity import _tie_class

pytestmark = pytest.mark.gpu

def _sweep(n, db, seed, channels=2, dc=0, chunk=1024, allow_row_ties=False, explain=True):
    import torch
    import modem_amd
    import modem_amd.ofdmrx as M
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        rx = modem_amd.Receiver(device=0, chunk_frames=chunk, stream=stream.cuda_stream)
        spf = rx.tx_frame_samples(6)
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        d_pay = torch.randint(0, 256, (n, 5380), dtype=torch.uint8, device=dev, generator=g)
        d_in = torch.empty((n, spf, 2), dtype=torch.int16, device=dev)
        rx.tx_encode(d_pay.data_ptr(), n, d_in.data_ptr())
        rx.awgn_tile(d_in.data_ptr(), n, d_in.data_ptr(), n, spf, db, seed + 1, 0)
        rx.synchronize()
        if channels == 1:       # the real part of the noisy analytic stream + a DC offset: what a 1-channel WAV holds (encode.cc:127-128)
            d_in = torch.clamp(d_in[:, :, 0].to(torch.int32) + dc, -32768, 32767).to(torch.int16).contiguous()
            torch.cuda.synchronize()
        d_out = torch.zeros((n, 5380), dtype=torch.uint8, device=dev)
        d_res = torch.zeros((n, M.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        rx.decode_device(d_in.data_ptr(), M.FMT_S16, channels, spf, spf * 2 * channels, n, d_out.data_ptr(), d_res.data_ptr())
        rx.synchronize()
        routes = (n - max(rx.list_decoded_frames(), 0) - max(rx.sc_decided_frames(), 0), rx.sc_decided_frames(), rx.list_decoded_frames())
        out = d_out.cpu().numpy()
        res = d_res.cpu().numpy().view(M.RESULT_DTYPE).reshape(-1)
        pcm = np.ascontiguousarray(d_in.cpu().numpy())
        pays = d_pay.cpu().numpy()
        rx.close()
    oout = np.zeros((n, 5380), np.uint8)
    ores = np.zeros(n * 56, np.uint8)
    O.lib().orc_decode_batch(O.ptr(pcm), O.FMT_S16, channels, spf, spf * 2 * channels, n, 8, O.ptr(oout), O.ptr(ores), min(os.cpu_count() or 1, 128))
    ores = ores.view(M.RESULT_DTYPE).reshape(-1)
    differ = [i for i in range(n) if not (out[i] == oout[i]).all() or any(res[nm][i] != ores[nm][i] for nm in NAMES)]
    classes = [_tie_class(i, out, oout, res, ores, pays) for i in differ]
    same = np.ones(n, bool)
    same[differ] = False
    ok = res["status"] == 0
    assert (out[ok] == pays[ok]).all()
    # the flip count: identical, or explained position by position
    fl = np.nonzero(same & (res["bit_flips"] != ores["bit_flips"]))[0]
    if len(fl) and explain:
        _explain_flips([pcm[i] for i in fl], channels, res["bit_flips"][fl], ores["bit_flips"][fl], allow_row_ties)
    return dict(differ=differ, classes=classes, routes=routes, ok=int(ok.sum()), flips_differ=len(fl), res=res, ores=ores)


def test_sweep_certified():
    """8192 frames at the headline's noise level: the syndrome certificate finishes every frame; nothing differs, not even a flip count"""
    s = _sweep(8192, -30.0, 3001)
    assert s["routes"] == (8192, 0, 0) and s["ok"] == 8192
    assert s["differ"] == [] and s["flips_differ"] == 0
    assert (s["res"]["bit_flips"] == 0).all() and (s["res"]["best_lane"] == 0).all()


def test_sweep_list_decoder():
    """3072 frames at -17 dB: below the reach of both certificates, every frame is list-decoded; everything decided identical"""
    s = _sweep(3072, -17.0, 3002, allow_row_ties=True)
    assert s["routes"][2] >= 3000 and s["ok"] == 3072
    assert s["differ"] == [], (s["differ"], s["classes"])


def test_sweep_waterfall():
    """4096 frames at -14.6 dB (decoded and lost frames mixed): identical but for the documented tie classes, at most three frames"""
    s = _sweep(4096, -14.6, 3003, chunk=512, allow_row_ties=True)
    assert 200 < s["ok"] < 4096 - 200
    assert len(s["differ"]) <= 3 and all(s["classes"]), (s["differ"], s["classes"])


def test_sweep_mono():
    """4096 MONO frames (DC offset 700 LSB) at -19 dB, where the list-1 pass decides.  Until round 6 the mono front end was 1e-6 of full
    scale from the oracle's (the oracle followed the reference's fp32 DC-blocker recurrence through its own rounding walk, the GPU
    runs a blocked scan): where the Schmidl-Cox arg-max sits on a plateau the two then took the coarse CFO from neighbouring samples,
    carrier ratios differed by up to 5e-4 and flip counts by up to 24.  With the oracle's DC blocker in double (both sides round the exact
    value; ORC_NUM_BLOCKDC_FP32 is the plain form, tests/test_oracle_numerics.py) mono input is as close as 2-channel input:
    everything decided identical but for timing ties (at most two), every flip-count difference explained position by position."""
    s = _sweep(4096, -19.0, 3004, channels=1, dc=700, allow_row_ties=True)
    assert s["ok"] == 4096
    assert len(s["differ"]) <= 2 and all(c == "timing" for c in s["classes"]), (s["differ"], s["classes"])
