#!/usr/bin/env python3
"""Level-store traffic model of the SCL decoder (N = 65536, L = 8) for the mode-6 frozen table.

Counts bytes per tree level under the rules of k_polar.hip (every array of a level >= TOP lives in HBM: written once
when produced, read once by the g step of its right child; a fused pass of up to three levels re-reads its lowest
level when the next pass continues from it), and evaluates the variants DESIGN.md 4c discusses.  CPU only; the frozen
table comes from the oracle (test infrastructure) - this is a planning tool, not product code.
usage: python tests/polar_traffic_model.py   (under tests/ because it asks the oracle for the table)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as O

N = 65536
fz = np.ctypeslib.as_array(O.lib().orc_frozen_get(0), shape=(2048,))
frozen = ((fz[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(np.uint8).reshape(-1)
cum = np.concatenate([[0], np.cumsum(frozen)])


def kind(t, m):
    """0 = all frozen, 1 = all information, 2 = mixed, for the node of 2^m leaves at t"""
    c = cum[t + (1 << m)] - cum[t]
    return 0 if c == (1 << m) else (1 if c == 0 else 2)


def model(top=8, rate0_max=7, rate1_max=11, frozen_left_fused=False, rate1_fused=False, verbose=False):
    """bytes per codeword by category.  Arrays of levels < top are on chip (free)."""
    W = np.zeros(17)
    R = np.zeros(17)
    notes = {}

    def full(m):
        return 32.0 * (1 << m)   # 8 paths x 4 B per position

    def visit(t, m, on_spine_left, compact):
        """decode node (t, m); its LLR array exists (stored unless noted).  compact: array identical for all paths"""
        k = kind(t, m)
        if k == 0 and m <= rate0_max:
            # decided on its own array: one read if the array is in HBM (m >= top)
            if m >= top:
                R[m] += full(m) / (8 if compact else 1)
            return
        if k == 1 and m <= rate1_max and t != 0:
            if m >= top and not rate1_fused:
                R[m] += 2 * full(m)      # mu sweep + sign sweep
            return
        if m < top or m == 0:
            return
        # left child: f from this array (produced in registers by the pass that made this array: no read), stored
        lk = kind(t, m - 1)
        lcompact = compact
        if m - 1 >= top:
            if frozen_left_fused and lk == 0:
                pass                      # penalties summed on the fly, array never stored
            else:
                W[m - 1] += full(m - 1) / (8 if lcompact else 1)
        visit(t, m - 1, on_spine_left, lcompact)
        # right child: g reads this array (both halves)
        if m >= top:
            if frozen_left_fused and lk == 0:
                pass                      # right child was formed in the same pass as the penalties (u = 0)
            else:
                R[m] += full(m) / (8 if compact else 1)
        if m - 1 >= top:
            rk = kind(t + (1 << (m - 1)), m - 1)
            if rate1_fused and rk == 1 and m - 1 <= rate1_max:
                pass                      # signs / min |.| taken from registers, never stored
            else:
                W[m - 1] += full(m - 1)
        visit(t + (1 << (m - 1)), m - 1, False, False)

    # root: level 16 = channel LLRs (compact, not counted as level store)
    visit(0, 16, True, True)
    return W, R


def report(name, **kw):
    W, R = model(**kw)
    w, r = W[:16].sum(), R[:16].sum()
    print("%-58s W %6.2f MB  R %6.2f MB  total %6.2f MB" % (name, w / 1e6, r / 1e6, (w + r) / 1e6))
    return W, R


if __name__ == "__main__":
    print("frozen leaves:", int(frozen.sum()), "of", N)
    for m in range(8, 16):
        ks = [kind(t, m) for t in range(0, N, 1 << m)]
        print("level %2d: %4d nodes, all-frozen %3d, all-info %3d" % (m, len(ks), ks.count(0), ks.count(1)))
    report("ideal (every level >= 8 array W once + R once), no shortcuts", rate0_max=0, rate1_max=0)
    report("current rules: rate-0 <= 128 leaves, rate-1 <= 2048", rate0_max=7, rate1_max=11)
    report("  + level 8 on chip", top=9)
    report("  + levels 8, 9 on chip", top=10)
    report("  + rate-0 nodes up to 4096 leaves on their own array", rate0_max=12)
    report("  + rate-0 up to 4096, left-frozen fused (never stored)", rate0_max=12, frozen_left_fused=True)
    report("  + rate-0 up to 4096, fused, rate-1 fused", rate0_max=12, frozen_left_fused=True, rate1_fused=True)
    report("  + all of it and level 8 on chip", top=9, rate0_max=12, frozen_left_fused=True, rate1_fused=True)
    report("  + all of it and levels 8, 9 on chip", top=10, rate0_max=12, frozen_left_fused=True, rate1_fused=True)
    report("  + rate-1 up to 4096 too", top=9, rate0_max=12, rate1_max=12, frozen_left_fused=True, rate1_fused=True)
    W, R = model()
    for m in range(8, 16):
        print("  level %2d: W %5.2f  R %5.2f MB" % (m, W[m] / 1e6, R[m] / 1e6))
