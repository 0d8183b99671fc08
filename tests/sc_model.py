"""numpy model of the SC-dominance certificate -- TEST INFRASTRUCTURE.

The list decoder of decode.cc:530 (restated in oracle/polar.c) follows, among its L paths, the one that takes the sign of
its LLR at every information leaf: P*.  sc_path() decodes P* alone with the same fp32 arithmetic (min-sum f, g = b +- a,
frozen penalties max(0, -llr) added leaf by leaf, an aligned all-frozen node of 2..128 leaves in butterfly order at once)
and returns with it
    M*        its path metric,
    min_fork  min over the information leaves i of fl(M*(i) + |llr_i|)  (what the candidate that leaves P* at i costs).
Rule: min_fork > M*  =>  P* is lane 0 of the list decoder at every fork and at the end, for every list size:
every other candidate descends from a first deviation at some leaf i (or from a placeholder path that starts at 1000 and
makes P*'s decisions: never cheaper than P*, and P* wins ties by its index), carries at least fl(M*(i) + |llr_i|) for ever
(fp32 sums of non-negative terms are monotone), so P* has the strictly smallest metric whenever candidates are ranked.
The syndrome certificate of round 3 is the case M* = 0.
"""
import numpy as np

f32 = np.float32


def frozen_bits(words, n):
    return ((np.asarray(words, dtype=np.uint32)[:, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(-1)[:n].astype(bool)


def sc_path(llr, frozen, rate0_max=7):
    """llr [N] float32, frozen [N] bool -> (codeword bits [N] uint8, M*, min_fork), all arithmetic in float32"""
    llr = np.asarray(llr, dtype=f32)
    state = {"M": f32(0), "fork": f32(np.inf)}

    def node(a, idx):
        n = a.size
        fz = frozen[idx:idx + n]
        if n == 1:
            v = a[0]
            if fz[0]:
                if v < 0:
                    state["M"] = f32(state["M"] - v)
                return np.zeros(1, np.uint8)
            cand = f32(state["M"] + np.abs(v))
            if not cand >= state["fork"]:
                state["fork"] = cand
            return np.array([1 if v < 0 else 0], np.uint8)
        if fz.all() and n <= (1 << rate0_max):
            p = np.where(a < 0, -a, f32(0)).astype(f32)
            h = n // 2
            while h >= 1:
                p = (p[:h] + p[h:2 * h]).astype(f32)
                h //= 2
            state["M"] = f32(state["M"] + p[0])
            return np.zeros(n, np.uint8)
        h = n // 2
        lo, hi = a[:h], a[h:]
        f = (np.minimum(np.abs(lo), np.abs(hi)) * np.where((lo < 0) != (hi < 0), f32(-1), f32(1))).astype(f32)
        ul = node(f, idx)
        g = np.where(ul == 1, hi - lo, lo + hi).astype(f32)
        ur = node(g, idx + h)
        return np.concatenate([ul ^ ur, ur])

    code = node(llr, 0)
    return code, state["M"], state["fork"]


def bec_frozen(level, k_info, p=0.5):
    """a frozen set of the usual shape: the BEC construction, the k_info most reliable positions carry information"""
    z = np.array([p], dtype=np.float64)
    for _ in range(level):
        z = np.stack([2 * z - z * z, z * z], axis=1).reshape(-1)
    order = np.argsort(z, kind="stable")
    fz = np.ones(1 << level, bool)
    fz[order[:k_info]] = False
    return fz


def pack_frozen(fz):
    return np.packbits(fz.astype(np.uint8), bitorder="little").view("<u4").astype(np.uint32)
