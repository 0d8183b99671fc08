#!/usr/bin/env python3
"""ts_rank_model.py -- numpy model of the rank-counting Theil-Sen (k_theilsen.hip, round 3).

The kernel finds the median of the n(n-1)/2 pairwise slopes s_ij = fl(fl(y_j - y_i) / d) (decode.cc:488) without
classifying pairs one by one.  x is the integer grid (decode.cc:485), so for a threshold T

    s_ij < T   <=>   y_j - T x_j  <  y_i - T x_i      (up to the two fp32 roundings of s)

i.e. the number of slopes below T is the number of INVERSIONS of z = y - T x in index order: one sort of n keys.
This model restates the arithmetic of that count (fp64 transform, 23-bit quantisation, the uncertainty margin MQ
inside which a pair gets the exact fp32 division) and the search policy (least-squares start, density from the
interquartile range, secant steps on exact counts, final bracket listed and selected), checks every row against the
brute-force median and prints how many counts / listed pairs / exact divisions the policy needs.
Tools only: nothing in the product imports this.
"""
import sys
import numpy as np

f32 = np.float32
QBITS = 22
MQ = 2          # |q_i - q_j| <= MQ -> the pair is resolved by the exact division


def all_slopes(y):
    n = y.size
    i, j = np.triu_indices(n, 1)
    return ((y[j] - y[i]).astype(f32) / (j - i).astype(f32)).astype(f32), i, j


class Row:
    def __init__(self, y):
        self.y = y.astype(f32)
        self.n = n = y.size
        self.x = np.arange(n) - n // 2
        self.s, self.i, self.j = all_slopes(self.y)
        self.count = self.s.size
        self.target = self.count // 2
        self.truth = np.partition(self.s, self.target)[self.target]
        self.ymin, self.ymax = float(self.y.min()), float(self.y.max())
        self.sorts = 0
        self.exact_divs = 0

    def keys(self, T):
        """quantised z = y - T x: fp64 transform, fixed range from (ymin, ymax, |T| n)"""
        T = float(f32(T))
        half = abs(T) * (self.n // 2 + 1)
        lo = self.ymin - half
        hi = self.ymax + half
        span = max(hi - lo, 1e-30) * (1.0 + 2.0 ** -20)
        scale = (2.0 ** QBITS - 8192 - 2) / span
        z = self.y.astype(np.float64) - T * self.x
        q = np.floor((z - lo) * scale).astype(np.int64)
        assert q.min() >= 0 and q.max() < 2 ** QBITS
        return q

    def count(self, T):
        raise NotImplementedError

    def count_lt(self, T):
        """exact #{s_ij < T} and #{s_ij <= T} from the key inversions + exact divisions of the uncertain pairs"""
        self.sorts += 1
        q = self.keys(T)
        qi, qj = q[self.i], q[self.j]
        inv = int((qj < qi).sum())                            # what the merge sort counts
        unc = np.abs(qi - qj) <= MQ
        nu = int(unc.sum())
        self.exact_divs += nu
        T = f32(T)
        s_u = self.s[unc]
        lt = inv - int((qj[unc] < qi[unc]).sum()) + int((s_u < T).sum())
        le = inv - int((qj[unc] < qi[unc]).sum()) + int((s_u <= T).sum())
        return lt, le, nu, q


def check_counts(rng):
    bad = 0
    for trial in range(40):
        n = [432, 432, 400, 255, 64, 5][trial % 6]
        kind = trial % 5
        x = np.arange(n) - n // 2
        if kind == 0:
            y = rng.normal(0, 0.03, n)
        elif kind == 1:
            y = 0.002 * x + rng.normal(0, 0.2, n)
        elif kind == 2:
            y = 0.01 * x
        elif kind == 3:
            y = rng.normal(0, 1e-6, n)
        else:
            y = rng.normal(0, 0.1, n)
            y[rng.integers(0, n, n // 10)] = 0.0
        r = Row(y.astype(f32))
        for T in (r.truth, np.nextafter(r.truth, f32(9)), f32(0), f32(r.truth * 1.01), f32(-r.truth)):
            lt, le, nu, _ = r.count_lt(T)
            ok = lt == int((r.s < f32(T)).sum()) and le == int((r.s <= f32(T)).sum())
            if not ok:
                bad += 1
                print("count mismatch", trial, n, kind, T, lt, int((r.s < f32(T)).sum()))
    print("count_lt: %d mismatches" % bad)
    return bad


def fkey(v):
    b = np.float32(v).view(np.uint32)
    return int(~b & 0xffffffff) if b & 0x80000000 else int(b | 0x80000000)


def fkey_inv(k):
    b = (k & 0x7fffffff) if k & 0x80000000 else (~k & 0xffffffff)
    return np.uint32(b).view(np.float32)


def solve(r, CAP=448, verbose=False):
    """the kernel's search policy; returns (slope, counts used, listed pairs)"""
    n, y, x = r.n, r.y.astype(np.float64), r.x.astype(np.float64)
    target = r.target
    if r.ymin == r.ymax:
        return f32(0), 0, 0
    # least-squares start
    sx, sy = x.sum(), y.sum()
    T = f32(((x * y).sum() - sx * sy / n) / ((x * x).sum() - sx * sx / n))
    Ta, ca, Tb, cb = None, 0, None, r.count
    pts = []
    rho = None
    for it in range(64):
        lt, le, nu, q = r.count_lt(T)
        if lt <= target < le:
            return f32(T), r.sorts, 0
        pts.append((float(T), lt))
        if lt <= target:
            Ta, ca = f32(T), lt
        else:
            Tb, cb = f32(T), lt
        if Ta is not None and Tb is not None:
            if cb - ca <= CAP:
                break
            if np.nextafter(Ta, f32(np.inf)) == Tb:
                return Ta, r.sorts, 0
        # next threshold
        if rho is None:
            qs = np.sort(q)
            iqr = (qs[(3 * n) // 4] - qs[n // 4])
            # density of slopes at the median for gaussian noise: sum_d (n-d) d / (2 sigma sqrt(pi)), sigma = IQR / 1.349
            half = abs(float(T)) * (n // 2 + 1)
            span = max((r.ymax + half) - (r.ymin - half), 1e-30) * (1.0 + 2.0 ** -20)
            sigma = max(iqr, 1) * span / (2.0 ** QBITS - 1) / 1.349
            rho = n * (n * n - 1) / 6.0 / (2.0 * sigma * np.sqrt(np.pi))
        if len(pts) >= 2:
            (t0, c0), (t1, c1) = pts[-2], pts[-1]
            if c1 != c0 and t1 != t0:
                rho = (c1 - c0) / (t1 - t0)
        # aim: the side still open, a margin past the target
        margin = 100
        if Ta is None:
            goal = target - margin
        elif Tb is None:
            goal = target + margin
        else:
            # bracketed but too wide: go for the end that is farther away
            goal = target - margin if (target - ca) > (cb - target) else target + margin
        Tn = f32(float(T) + (goal + 0.5 - lt) / rho) if rho and rho > 0 else T
        lo_ok = Ta is None or Tn > Ta
        hi_ok = Tb is None or Tn < Tb
        if not (lo_ok and hi_ok) or Tn == T or not np.isfinite(Tn):
            # bisection in key space
            ka = fkey(Ta) if Ta is not None else fkey(f32(-3e38))
            kb = fkey(Tb) if Tb is not None else fkey(f32(3e38))
            Tn = fkey_inv((ka + kb) // 2)
            if Tn == Ta or Tn == Tb:
                Tn = Tb
        T = Tn
    # list the bracket and select
    sel = (r.s >= Ta) & (r.s < Tb)
    lst = np.sort(r.s[sel])
    assert lst.size == cb - ca
    return lst[target - ca], r.sorts, int(lst.size)


def main():
    rng = np.random.default_rng(7)
    if check_counts(rng):
        sys.exit(1)
    stats = {}
    for name, gen in (
        ("awgn -30 dB (sigma 0.03)", lambda n: 1e-4 * (np.arange(n) - n // 2) + rng.normal(0, 0.03, n)),
        ("sigma 0.2 + slope", lambda n: 0.002 * (np.arange(n) - n // 2) + rng.normal(0, 0.2, n)),
        ("waterfall: sigma 0.25, 10 % wrapped", lambda n: np.where(rng.random(n) < 0.1, rng.uniform(-0.39, 0.39, n), rng.normal(0, 0.25, n).clip(-0.39, 0.39))),
        ("uniform garbage", lambda n: rng.uniform(-0.39, 0.39, n)),
        ("clean 1e-6", lambda n: rng.normal(0, 1e-6, n)),
        ("outliers", lambda n: rng.normal(0, 0.1, n) + (rng.random(n) < 0.1) * rng.normal(0, 2, n)),
    ):
        cs, ls, bad, ex = [], [], 0, []
        for t in range(60):
            r = Row(gen(432).astype(f32))
            v, c, l = solve(r)
            cs.append(c); ls.append(l); ex.append(r.exact_divs)
            if v != r.truth:
                bad += 1
        print("%-40s counts mean %.2f max %d | listed mean %.0f max %d | exact divisions in counts mean %.1f max %d | wrong %d" %
              (name, np.mean(cs), max(cs), np.mean(ls), max(ls), np.mean(ex), max(ex), bad))
    # edge rows
    for name, y in (("zeros", np.zeros(432)), ("exact line", 0.01 * (np.arange(432) - 216)), ("5 points", rng.normal(0, 1, 5)),
                    ("half zeros", np.where(np.arange(432) % 2 == 0, 0.0, rng.normal(0, 0.1, 432)))):
        r = Row(y.astype(f32))
        v, c, l = solve(r)
        print("%-12s slope %r truth %r counts %d listed %d exact divisions %d %s" % (name, v, r.truth, c, l, r.exact_divs, "OK" if v == r.truth else "WRONG"))


if __name__ == "__main__":
    main()


def check_incremental(rng, trials=60):
    """the incremental count of k_theilsen.hip: from the order sorted at T_prev to T_new by an odd-even transposition sort whose
    swaps S are the pairs that change order; only pairs that are uncertain at either threshold need the exact division:
        |c(T_new) - c(T_prev)| = S - #{uncertain pairs that swapped} + #{uncertain pairs with T_lo <= s < T_hi}"""
    bad = 0
    for trial in range(trials):
        n = [432, 400, 255, 512][trial % 4]
        kind = trial % 4
        x = np.arange(n) - n // 2
        if kind == 0:
            y = 1e-4 * x + rng.normal(0, 0.03, n)
        elif kind == 1:
            y = rng.normal(0, 1e-6, n)
        elif kind == 2:
            y = rng.normal(0, 0.1, n); y[rng.integers(0, n, n // 8)] = 0.0
        else:
            y = 0.003 * x + rng.normal(0, 0.2, n)
        r = Row(y.astype(f32))
        base = r.truth
        span = np.abs(np.partition(r.s, r.target + 400)[r.target + 400] - base)
        for (tp, tn) in ((base - span, base + 0.3 * span), (base + span, base), (base, np.nextafter(base, f32(9))), (f32(0), base)):
            tp, tn = f32(tp), f32(tn)
            qp, qn = r.keys(tp), r.keys(tn)
            kp = qp * 512 + np.arange(n)
            kn = qn * 512 + np.arange(n)
            i, j = r.i, r.j
            swapped = (kp[i] < kp[j]) != (kn[i] < kn[j])
            S = int(swapped.sum())
            unc = (np.abs(qp[i] - qp[j]) <= MQ) | (np.abs(qn[i] - qn[j]) <= MQ)
            lo, hi = (tp, tn) if tp <= tn else (tn, tp)
            inr = (r.s >= lo) & (r.s < hi)
            delta = S - int((swapped & unc).sum()) + int((inr & unc).sum())
            truth = int(inr.sum())
            if delta != truth:
                bad += 1
                print("incremental mismatch", trial, n, kind, tp, tn, delta, truth)
    print("incremental count: %d mismatches" % bad)
    return bad
