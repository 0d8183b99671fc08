"""The oracle's numerics choices are decision-neutral (VERDICT r1 weak #1 / ADVICE medium).

The default restatement fixes a few fp32 evaluation orders the reference's -Ofast build leaves open, in forms a
wavefront can reproduce bit for bit: all-frozen nodes charged once in butterfly order, survivors stored in rank order,
sliding sums as differences of double prefix sums, a closed-form NCO, per-row double SNR sums.  oracle/ carries a switch
for the PLAIN form of each (modem_oracle.h ORC_NUM_*): leaf-by-leaf frozen penalties, survivors left in candidate
order, fp32 add-tree sliding sums, the recursive renormalised phasor, term-by-term fp32 SNR sums.  Over frames from
-20 dB down through the waterfall to -14 dB, plus the configs[3] impairment chain, every DECISION must be the same with
all switches thrown: payload bytes, status, sync position, header fields.  The index of the winning lane may differ
with the survivor order (the CRC-selected payload may not); diagnostics that are fp32 values may move in the last bits.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

ALL_PLAIN = 1 | 2 | 4 | 8 | 16
LEVELS = [-20.0, -18.0, -17.0, -16.0, -15.5, -15.0, -14.5, -14.0]
PER_LEVEL = 112          # 8 x 112 = 896 AWGN frames + 128 impaired frames = 1024
THREADS = min(os.cpu_count() or 1, 16)


def _set(flags):
    L = O.lib()
    L.orc_set_numerics.argtypes = [C.c_uint]
    L.orc_set_numerics(flags)


def _batch(pcm):
    n, spf = pcm.shape[0], pcm.shape[1]
    out = np.zeros((n, 5380), np.uint8)
    res = (O.Result * n)()
    O.lib().orc_decode_batch(O.ptr(pcm), O.FMT_S16, 2, spf, spf * 4, n, 8, O.ptr(out), C.cast(res, C.c_void_p), THREADS)
    fields = {f: np.array([getattr(r, f) for r in res]) for f, _ in O.Result._fields_}
    return out, fields


@pytest.fixture(scope="module")
def frames():
    clean = [O.encode_pcm(O.payload_for(500 + i), channels=2) for i in range(8)]
    pcm = np.zeros((len(LEVELS) * PER_LEVEL + 128, clean[0].shape[0], 2), np.int16)
    f = 0
    for li, db in enumerate(LEVELS):
        for q in range(PER_LEVEL):
            pcm[f] = O.impair(clean[q % 8], noise_db=db, seed=77, frame=f)
            f += 1
    taps = [(0, 1 + 0j), (5, 0.35 - 0.1j), (11, -0.2 + 0.2j), (23, 0.1 + 0.05j)]
    for q in range(128):     # README.md:49 chain: multipath | cfo | sfo | awgn, a few noise levels
        pcm[f] = O.impair(clean[q % 8], noise_db=[-30.0, -20.0, -17.0, -16.0][q % 4], cfo_hz=234.567, sfo_ppm=147.0,
                          multipath=taps, seed=78, frame=f)
        f += 1
    return pcm


@pytest.fixture(scope="module")
def baseline(frames):
    _set(0)
    return _batch(frames)


def _same_decisions(a, b, lanes_equal):
    (oa, ra), (ob, rb) = a, b
    assert (oa == ob).all(), "payload bytes differ in frames %s" % np.nonzero((oa != ob).any(axis=1))[0][:8]
    for f in ("status", "sc_start", "symbol_pos", "oper_mode", "call_sign", "n_sync_rejects"):
        assert (ra[f] == rb[f]).all(), (f, np.nonzero(ra[f] != rb[f])[0][:8])
    assert ((ra["best_lane"] >= 0) == (rb["best_lane"] >= 0)).all()
    if lanes_equal:
        assert (ra["best_lane"] == rb["best_lane"]).all()
    assert np.abs(ra["cfo_rad"] - rb["cfo_rad"]).max() <= 1e-5
    # the flip count compares LLR signs with decoded bits (decode.cc:546-555): an LLR within rounding of zero may flip
    assert np.abs(ra["bit_flips"].astype(np.int64) - rb["bit_flips"]).max() <= 4


def test_all_plain_forms_change_no_decision(frames, baseline):
    try:
        _set(ALL_PLAIN)
        plain = _batch(frames)
    finally:
        _set(0)
    _same_decisions(baseline, plain, lanes_equal=False)
    st = baseline[1]["status"]
    n_awgn = len(LEVELS) * PER_LEVEL
    # the sweep really crosses the waterfall: everything decodes at -20 dB, nothing at -14 dB
    assert (st[:PER_LEVEL] == 0).all() and (st[n_awgn - PER_LEVEL:n_awgn] != 0).all()
    assert (st[n_awgn:] == 0).sum() >= 96       # impaired frames at -30 / -20 / -17 dB decode
    # lane order is observable only through best_lane: report how often it moved
    moved = int((baseline[1]["best_lane"] != plain[1]["best_lane"]).sum())
    print("best_lane differs in %d of %d frames (survivor order only)" % (moved, frames.shape[0]))


@pytest.mark.parametrize("flag,lanes_equal", [(1, True), (2, False), (4, True), (8, True), (16, True)])
def test_each_plain_form_alone(frames, baseline, flag, lanes_equal):
    """one switch at a time on a subset around the waterfall (-16 ... -14.5 dB) and the impaired frames"""
    lo, hi = 3 * PER_LEVEL, 7 * PER_LEVEL
    idx = np.concatenate([np.arange(lo, hi, 4), np.arange(len(LEVELS) * PER_LEVEL, frames.shape[0], 4)])
    sub = np.ascontiguousarray(frames[idx])
    try:
        _set(flag)
        got = _batch(sub)
    finally:
        _set(0)
    ref = (baseline[0][idx], {k: v[idx] for k, v in baseline[1].items()})
    _same_decisions(ref, got, lanes_equal)


def test_blockdc_in_double_changes_no_decision_on_mono_input():
    """ORC_NUM_BLOCKDC_FP32 (round 6): the DC blocker of decode.cc:299 runs with its state in double and rounds every output once by
    default; the plain form is the fp32 recurrence.  96 MONO frames (the real part of the noisy analytic stream + a DC offset, what a
    1-channel WAV holds, encode.cc:127-128) from -24 dB through the waterfall: payload, status, sync position, header and lane are the
    same with either form, and the analytic signals differ by no more than the fp32 recurrence's own rounding walk (a few 1e-6)."""
    clean = [O.encode_pcm(O.payload_for(700 + i), channels=2) for i in range(4)]
    levels = [-24.0, -19.0, -17.0, -16.0, -15.0, -14.5]
    per = 16
    pcm = np.zeros((len(levels) * per, clean[0].shape[0], 1), np.int16)
    f = 0
    for db in levels:
        for q in range(per):
            z = O.impair(clean[q % 4], noise_db=db, seed=79, frame=f)
            pcm[f, :, 0] = np.clip(z[:, 0].astype(np.int32) + (700 if q % 2 else -2500), -32768, 32767)
            f += 1

    def run():
        n, spf = pcm.shape[0], pcm.shape[1]
        out = np.zeros((n, 5380), np.uint8)
        res = (O.Result * n)()
        O.lib().orc_decode_batch(O.ptr(pcm), O.FMT_S16, 1, spf, spf * 2, n, 8, O.ptr(out), C.cast(res, C.c_void_p), THREADS)
        return out, {fl: np.array([getattr(r, fl) for r in res]) for fl, _ in O.Result._fields_}

    def analytic(frame):
        z = np.zeros((frame.shape[0], 2), np.float32)
        O.lib().orc_front_end(O.ptr(np.ascontiguousarray(frame)), O.FMT_S16, 1, frame.shape[0], O.ptr(z))
        return z

    try:
        _set(0)
        base, z0 = run(), analytic(pcm[3])
        _set(32)
        plain, z1 = run(), analytic(pcm[3])
    finally:
        _set(0)
    _same_decisions(base, plain, lanes_equal=True)
    st = base[1]["status"]
    assert (st[:per] == 0).all() and (st[-per:] != 0).sum() >= per // 2     # from all decoded to mostly lost
    scale = np.abs(z0).max()
    assert 1e-8 * scale < np.abs(z0 - z1).max() <= 2e-5 * scale
