"""The clean-node shortcut of k_sc (round 6) against the oracle's leaf-by-leaf walk, on the CPU.

k_sc does not walk a node whose input hard decisions are already a codeword of the node's sub-code (k_sc.hip: "CLEAN nodes"): it returns
them, adds no penalty, and takes the node's share of min_fork from a level-by-level evaluation of MAGNITUDES (|f| = min, |g| = sum).
Here the same rule, restated in numpy on small codes (2^9 .. 2^11 leaves, random frozen sets, codewords with a few to many raw errors,
nodes tested at every size from 2 leaves up), must give the oracle's sign-following path (oracle/polar.c: orc_polar_sc_path - a plain
walk with the list decoder's arithmetic) bit for bit: the re-encoded codeword, the fp32 path metric M*, the fp32 min_fork.  The lower bound
the kernel uses above 4096 leaves (the smallest magnitude of the node's ARRAY) must never exceed the exact figure."""
import numpy as np

import oracle_lib as O

F32 = np.float32
RATE0_MIN, RATE0_MAX = 1, 7                       # oracle/polar.c: the all-frozen nodes whose penalties are summed butterfly-wise


def _transform(bits):
    """u = x F (the polar transform over GF(2)); its own inverse"""
    u = bits.copy()
    n, t = u.size, 1
    while t < n:
        v = u.reshape(-1, 2 * t)
        v[:, :t] ^= v[:, t:]
        t *= 2
    return u


def _leaf_magnitudes(mag):
    """the magnitudes of a clean node's leaf LLRs: at every level the left child takes min(|a|, |b|), the right one fl(|a| + |b|)"""
    a = mag.astype(F32).copy()
    t = a.size // 2
    while t >= 1:
        v = a.reshape(-1, 2 * t)
        lo, hi = np.minimum(v[:, :t], v[:, t:]), (v[:, :t] + v[:, t:]).astype(F32)
        v[:, :t], v[:, t:] = lo, hi
        t //= 2
    return a


class _Walk:
    def __init__(self, frozen, clean_from):
        self.fz, self.M, self.fork, self.clean_from = frozen, F32(0), F32(np.inf), clean_from
        self.skipped = self.bound_ok = 0

    def node(self, lam, idx):
        n = lam.size
        fz = self.fz[idx:idx + n]
        if n == 1:
            if fz[0]:
                if lam[0] < 0:
                    self.M = F32(self.M - lam[0])
                return np.zeros(1, np.uint8)
            self.fork = min(self.fork, F32(self.M + abs(lam[0])))
            return np.array([lam[0] < 0], np.uint8)
        m = n.bit_length() - 1
        if RATE0_MIN <= m <= RATE0_MAX and fz.all():
            p = np.where(lam < 0, -lam, F32(0)).astype(F32)
            h = n // 2
            while h >= 1:
                p[:h] = p[:h] + p[h:2 * h]
                h //= 2
            self.M = F32(self.M + p[0])
            return np.zeros(n, np.uint8)
        if n >= self.clean_from and (lam != 0).all():
            hard = (lam < 0).astype(np.uint8)
            if not (_transform(hard) & fz).any():                 # clean: nothing below is walked
                self.skipped += n
                if not fz.all():
                    leaf = _leaf_magnitudes(np.abs(lam))
                    exact = leaf[~fz.astype(bool)].min()
                    assert np.abs(lam).min() <= exact             # the bound used above 4096 leaves is a lower bound
                    self.fork = min(self.fork, F32(self.M + exact))
                return hard
        h = n // 2
        a, b = lam[:h], lam[h:]
        f = (np.sign(a) * np.sign(b) * np.minimum(np.abs(a), np.abs(b))).astype(F32)
        left = self.node(f, idx)
        g = (np.where(left == 1, -a, a) + b).astype(F32)
        right = self.node(g, idx + h)
        return np.concatenate([left ^ right, right])


def _case(rng, level, frozen_share, errors, sigma):
    n = 1 << level
    # a frozen set with the shape of a polar code's: the share decreasing with the index (plus noise)
    score = np.array([bin(i).count("1") for i in range(n)]) + rng.normal(0, 0.8, n)
    fz = np.zeros(n, np.uint8)
    fz[np.argsort(score)[:int(frozen_share * n)]] = 1
    u = rng.integers(0, 2, n).astype(np.uint8) & (1 - fz)
    x = _transform(u)
    llr = ((1.0 - 2.0 * x) * np.abs(rng.normal(4.0, sigma, n))).astype(F32)
    llr[llr == 0] = F32(0.5)
    for p in rng.choice(n, errors, replace=False):
        llr[p] = F32(-llr[p] * rng.uniform(0.05, 0.6))
    words = np.zeros(n // 32, np.uint32)
    for i in np.nonzero(fz)[0]:
        words[i // 32] |= np.uint32(1 << (i % 32))
    return fz, words, llr


def test_clean_node_rule_equals_the_leaf_walk():
    rng = np.random.default_rng(606)
    skipped = total = decided = 0
    for trial in range(60):
        level = int(rng.integers(9, 12))
        fz, words, llr = _case(rng, level, rng.uniform(0.25, 0.6), int(rng.choice([0, 1, 2, 5, 20, 80])), rng.uniform(0.5, 2.5))
        if trial % 7 == 0:
            llr[int(rng.integers(0, llr.size))] = F32(0)          # a zero input: nodes that hold it are walked
        code, M, fork = O.polar_sc_path(llr, words, level)
        for clean_from in (2, 64):                                # every node size / the kernel's smallest (64 leaves)
            w = _Walk(fz, clean_from)
            got = w.node(llr, 0)
            assert (got == code).all(), (trial, clean_from)
            assert w.M == M and (w.fork == fork or (np.isnan(fork) and np.isnan(w.fork))), (trial, clean_from, w.M, M, w.fork, fork)
        skipped += w.skipped
        total += llr.size
        decided += int(fork > M)
    assert skipped > total // 3 and decided > 20                  # (the shortcut was taken, and the rule held, often enough to mean something)
