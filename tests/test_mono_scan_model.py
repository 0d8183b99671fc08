"""CPU model of the blocked DC-blocker recurrence the mono front end runs on the GPU (modem_amd/csrc/mono_front.h, dev_common.h:
WScan) against the serial recurrence of the reference (decode.cc:294-301: BlockDC) as the oracle restates it (oracle/dsp.c).
No GPU: numpy emulations of the data-parallel-primitive steps, so that the algebra - the low-pass form of the DC blocker, the
weights of the wave scan, the fold of the kept states - is checked wherever the CPU suite runs."""
import numpy as np


def serial_dc_block(x, a, b):
    """y[n] = b (x[n] - x[n-1]) + a y[n-1], x[-1] = y[-1] = 0 (oracle/dsp.c, in double here)"""
    y = np.zeros_like(x)
    x1 = y1 = 0.0
    for n, x0 in enumerate(x):
        y1 = b * (x0 - x1) + a * y1
        x1 = x0
        y[n] = y1
    return y


def wscan(e, A):
    """dev_common.h WScan on one wave: v[l] = sum_{j <= l} A^(l - j) e[j] by four row_shr steps inside the rows of 16 lanes, then
    row_bcast:15 (rows 1 and 3 take lane 15 of the row before) and row_bcast:31 (rows 2 and 3 take lane 31)"""
    v = np.array(e, dtype=np.float64)
    lane = np.arange(64)
    for k, d in enumerate((1, 2, 4, 8)):
        src = np.where((lane & 15) >= d, np.roll(v, d), 0.0)           # row_shr:d, lanes without a source get 0
        v = v + A ** d * src
    w16 = A ** ((lane & 15) + 1)
    b15 = np.where((lane >> 4) & 1, v[(lane & ~15) - 1], 0.0)         # row_bcast:15, row mask 0xa
    v = v + w16 * b15
    w32 = A ** ((lane & 31) + 1)
    b31 = np.where(lane >= 32, v[31], 0.0)                             # row_bcast:31, row mask 0xc
    return v + w32 * b31


def test_weighted_wave_scan():
    rng = np.random.default_rng(1)
    for A in (0.9993 ** 8, 0.9993 ** 5, 0.5):
        e = rng.normal(size=64)
        ref = np.zeros(64)
        acc = 0.0
        for l in range(64):
            acc = A * acc + e[l]
            ref[l] = acc
        assert np.allclose(wscan(e, A), ref, rtol=1e-12, atol=1e-12)


def blocked_dc_block(x, a, b, per=8, ck=64):
    """mono_front.h: s[n] = a s[n-1] + g x[n], y[n] = b x[n] - s[n-1]; the kept state after every 64th sample (k_mono_carries), a
    span of 64 threads x `per` samples from the kept state before it (MonoCover::span)"""
    n = len(x)
    g = b * (1.0 - a)
    # k_mono_carries: the state after every ck-th sample
    s = 0.0
    kept = []
    for i in range(n):
        s = a * s + g * x[i]
        if (i + 1) % ck == 0:
            kept.append(s)
    y = np.zeros(n)
    span = 64 * per
    for start in range(0, n, span):                                    # spans start on kept states (multiples of 64)
        S = 0.0 if start == 0 else kept[start // ck - 1]
        xs = np.zeros(span)
        m = min(span, n - start)
        xs[:m] = x[start:start + m]
        chunks = xs.reshape(64, per)
        sl = np.zeros((64, per))                                       # thread chunks from a zero state
        for t in range(64):
            acc = 0.0
            for i in range(per):
                acc = a * acc + g * chunks[t, i]
                sl[t, i] = acc
        v = wscan(sl[:, -1], a ** per)                                 # ends of the chunks, scanned over the wave
        prev = np.concatenate(([0.0], v[:-1]))                         # wave_shr:1
        cin = prev + a ** (per * np.arange(64)) * S                    # the state entering each thread
        for t in range(64):
            sprev = cin[t]
            for i in range(per):
                k = start + t * per + i
                if k < n:
                    y[k] = b * chunks[t, i] - sprev
                sprev = sl[t, i] + a ** (i + 1) * cin[t]
    return y


def test_low_pass_form_and_blocked_scan_equal_the_serial_recurrence():
    rng = np.random.default_rng(7)
    s = 2.0 * (1280 + 160)
    a = (s - 1.0) / s                                                  # BlockDC::samples(2 (symbol_len + guard_len)), decode.cc:386
    b = (1.0 + a) / 2.0
    x = rng.normal(0.0, 0.2, 5000) + 0.4                               # a signal on a DC offset
    x[3000:3300] = 0.0
    ref = serial_dc_block(x, a, b)
    for per in (8, 5):
        got = blocked_dc_block(x, a, b, per=per)
        assert np.abs(got - ref).max() < 1e-12
