"""Which cheap slope estimate lands closest (in ranks of the 93 096 pairwise slopes) to the Theil-Sen median?  numpy only, no GPU.
The rank kernel needs a count within 30 ranks of the median to skip its second count; none of these gets there (least squares:
median miss 400 ranks; the median over all pairs at least 288 columns apart: 340).  python3 tools/experiments/ts_first_threshold_study.py"""
import numpy as np
rng = np.random.default_rng(1)
n = 432
x = np.arange(n) - n // 2
iu = np.triu_indices(n, 1)
dx = (x[iu[1]] - x[iu[0]]).astype(np.float64)
target = len(dx) // 2
def rankdist(sl, est):
    return abs(int((sl < est).sum()) - target)
res = {k: [] for k in ("ls", "med216", "med_long", "ls_trim", "med144x2", "hl")}
for trial in range(60):
    sigma = 10 ** rng.uniform(-3, -0.5)
    kind = trial % 3
    y = 1e-3 * rng.normal() * x + rng.normal(0, 0.05)
    if kind == 0: y = y + rng.normal(0, sigma, n)
    elif kind == 1: y = y + rng.standard_t(3, n) * sigma
    else: y = y + np.where(rng.random(n) < 0.1, rng.uniform(-0.39, 0.39, n), rng.normal(0, sigma, n))
    sl = (y[iu[1]] - y[iu[0]]) / dx
    ls = np.polyfit(x, y, 1)[0]
    res["ls"].append(rankdist(sl, ls))
    res["med216"].append(rankdist(sl, np.median((y[216:] - y[:216]) / 216.0)))
    # median over all pairs with baseline >= 288 (10k pairs) - the ideal "long baseline" estimator
    m = dx >= 288
    res["med_long"].append(rankdist(sl, np.median(sl[m])))
    # LS after trimming the 10 % largest residuals of a first LS
    p = np.polyfit(x, y, 1); r = y - np.polyval(p, x); keep = np.abs(r) <= np.quantile(np.abs(r), 0.9)
    res["ls_trim"].append(rankdist(sl, np.polyfit(x[keep], y[keep], 1)[0]))
    # median of the slopes of baselines 144 and 288 (two sets of 288 / 144 pairs)
    s2 = np.concatenate([(y[144:] - y[:-144]) / 144.0, (y[288:] - y[:-288]) / 288.0])
    res["med144x2"].append(rankdist(sl, np.median(s2)))
    # weighted median proxy: median of slopes between block means (blocks of 8: 54 points -> 1431 pairs)
    yb = y.reshape(54, 8).mean(axis=1); xb = x.reshape(54, 8).mean(axis=1)
    ib = np.triu_indices(54, 1)
    res["hl"].append(rankdist(sl, np.median((yb[ib[1]] - yb[ib[0]]) / (xb[ib[1]] - xb[ib[0]]))))
for k, v in res.items():
    v = np.array(v)
    print("%-9s median |rank miss| %6.0f   mean %6.0f   within 30: %4.0f %%   within 100: %4.0f %%" % (k, np.median(v), v.mean(), 100 * (v <= 30).mean(), 100 * (v <= 100).mean()))
