#!/usr/bin/env python3
"""tests/ts_soak.py [rows_per_length] (test infrastructure: it calls the oracle, so it lives under tests/) -- the rank-counting Theil-Sen kernel against the oracle's nth_element on many random rows of every
row length of the mode table: gaussian, heavy-tailed, clustered and quantised residuals (the shapes that make the search's density
estimate miss, so that its retry and closed-bracket paths run).  Prints the number of rows whose slope or intercept differ."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import modem_amd
import oracle_lib as O

def make_rows(rng, cols, n):
    x = np.arange(cols, dtype=np.float64) - cols // 2
    rows = []
    for r in range(n):
        kind = r % 6
        sigma = 10 ** rng.uniform(-3, -0.5)
        base = rng.normal(0, 1e-3) * x + rng.normal(0, 0.05)
        if kind == 0:
            y = base + rng.normal(0, sigma, cols)
        elif kind == 1:
            y = base + rng.standard_t(1.5, cols) * sigma * 0.3                  # heavy tails
        elif kind == 2:
            y = base + np.where(rng.random(cols) < 0.3, rng.uniform(-0.39, 0.39, cols), rng.normal(0, sigma * 0.2, cols))
        elif kind == 3:
            y = np.round(base + rng.normal(0, sigma, cols), 2 + int(rng.integers(0, 2)))   # quantised: many tied slopes
        elif kind == 4:
            y = base + rng.normal(0, sigma, cols) * (1 + 5 * (np.arange(cols) > cols * rng.random()))   # heteroscedastic
        else:
            y = base + rng.normal(0, sigma, cols); y[rng.integers(0, cols, int(rng.integers(1, cols // 2)))] = 0.0
        rows.append(y)
    return np.stack(rows).astype(np.float32)

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    rx = modem_amd.Receiver(device=0)
    rng = np.random.default_rng(2024)
    bad = total = 0
    t0 = time.time()
    for cols in (432, 400, 360, 512, 384, 256):
        rows = make_rows(rng, cols, n)
        s, yi = rx.theil_sen(rows)
        for r in range(rows.shape[0]):
            os_, oy = O.theil_sen(rows[r])
            ok = s[r] == np.float32(os_) and yi[r] == np.float32(oy)
            if not ok and bad < 10:
                print("MISMATCH cols %d row %d kind %d: gpu %r %r oracle %r %r" % (cols, r, r % 6, s[r], yi[r], os_, oy), flush=True)
            bad += not ok
            total += 1
    print("%d rows, %d mismatches (%.0f s)" % (total, bad, time.time() - t0))
    rx.close()
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
