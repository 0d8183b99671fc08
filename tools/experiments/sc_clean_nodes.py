"""How often is a node of the sign-following path's tree CLEAN - the hard decisions of its input LLRs already a codeword of its sub-code, no
zero among them - so that successive cancellation returns those hard decisions without walking the node, and how much certificate is lost if the
smallest input magnitude stands in for the smallest leaf magnitude (a lower bound)?  CPU only (the oracle), run from the repo root:
    python3 tools/experiments/sc_clean_nodes.py [frames] [noise_db | chain]
"""
import sys
import numpy as np
sys.path.insert(0, "tests")
import oracle_lib as O

def tree(llr, code, frz):
    """arrays of every level in parallel form (the partial sums are the path's own): yields (m, lam[nodes, 2^m], x[nodes, 2^m])"""
    lam = llr.reshape(1, -1).astype(np.float32)
    x = code.reshape(1, -1).astype(np.uint8)
    out = []
    for m in range(16, -1, -1):
        out.append((m, lam, x))
        if m == 0:
            break
        n = 1 << (m - 1)
        a, b = lam[:, :n], lam[:, n:]
        xl, xh = x[:, :n] ^ x[:, n:], x[:, n:]
        f = (np.sign(a) * np.sign(b) * np.minimum(np.abs(a), np.abs(b))).astype(np.float32)
        g = (np.where(xl == 1, -a, a) + b).astype(np.float32)
        lam = np.stack([f, g], 1).reshape(-1, n)
        x = np.stack([xl, xh], 1).reshape(-1, n)
    return out

TOP = 15
VERBOSE = len(sys.argv) > 3
def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    what = sys.argv[2] if len(sys.argv) > 2 else "-20"
    fr = O.frozen(0)
    frz = ((fr[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).reshape(-1).astype(bool)
    tot = {}
    weak_ok = weak2_ok = exact_ok = 0
    saved = []
    for i in range(n):
        pcm = O.encode_pcm(O.payload_for(i), bits=16, channels=2)
        if what == "chain":
            pcm = O.impair(pcm, noise_db=-30.0, cfo_hz=234.567, sfo_ppm=147.0, multipath=[(0, 1 + 0j), (5, 0.35 - 0.1j), (11, -0.2 + 0.2j), (23, 0.1 + 0.05j)], seed=5, frame=i)
        else:
            pcm = O.impair(pcm, noise_db=float(what), seed=5, frame=i)
        out, res, tb = O.decode(pcm, taps=True)
        llr = tb.llr.copy()
        code, M, fork = O.polar_sc_path(llr)
        exact_ok += fork > M
        lev = tree(llr, code, frz)
        leaf_l = lev[16][1].reshape(-1)
        pen = np.where(frz, np.maximum(-leaf_l, 0), 0).astype(np.float64)
        Mpre = np.concatenate([[0], np.cumsum(pen)])           # metric before leaf i
        covered = np.zeros(65536, bool)
        fork_lb = fork_lb2 = np.inf
        by_level = {m: (lam, x) for m, lam, x in lev}
        walked = 0                                             # leaves still walked
        for m, lam, x in lev:
            if m > TOP or m < 6:
                continue
            nodes = lam.shape[0]
            sz = 1 << m
            hard = (lam < 0).astype(np.uint8)
            clean = (hard == x).all(1) & (lam != 0).all(1)
            parent_cov = covered.reshape(nodes, sz).all(1)
            fresh = clean & ~parent_cov
            t = tot.setdefault(m, [0, 0])
            t[0] += int((~parent_cov).sum()); t[1] += int(fresh.sum())
            for k in np.nonzero(fresh)[0]:
                covered[k * sz:(k + 1) * sz] = True
                inf = ~frz[k * sz:(k + 1) * sz]
                if inf.any():
                    ex = np.abs(leaf_l[k * sz:(k + 1) * sz][inf]).min()
                    # what the kernel takes: the exact smallest leaf magnitude up to 4096 leaves, the smallest input magnitude above
                    if m >= 14:                                 # the kernel's policy: only where skipping saves level-store traffic
                        l12 = np.abs(by_level[12][1][k * (sz >> 12):(k + 1) * (sz >> 12)])
                        has = [(~frz[(k * (sz >> 12) + t) * 4096:(k * (sz >> 12) + t + 1) * 4096]).any() for t in range(sz >> 12)]
                        b2 = min(l12[t].min() for t in range(sz >> 12) if has[t])
                        fork_lb2 = min(fork_lb2, Mpre[k * sz] + b2)
                        fork_lb = min(fork_lb, Mpre[k * sz] + np.abs(lam[k]).min())
                    else:
                        fork_lb = min(fork_lb, Mpre[k * sz] + ex); fork_lb2 = min(fork_lb2, Mpre[k * sz] + ex)
                    if m >= 12 and VERBOSE:
                        print(f"   clean 2^{m} node {k}: M_start {Mpre[k * sz]:.1f}  min|input| {np.abs(lam[k]).min():.1f}  exact min info leaf {ex:.1f}  (needs > {Mpre[-1] - Mpre[k * sz]:.1f})")
        rest = ~covered & ~frz
        if rest.any():
            fork_lb = min(fork_lb, (Mpre[:-1][rest] + np.abs(leaf_l[rest])).min()); fork_lb2 = min(fork_lb2, fork_lb)
        weak_ok += fork_lb > Mpre[-1] * (1 + 1e-5); weak2_ok += fork_lb2 > Mpre[-1] * (1 + 1e-5)
        saved.append(covered.mean())
        print(f"frame {i}: raw errors {int(((llr < 0) != (code == 1)).sum())}  M* {M:.4g} min_fork {fork:.4g} weak bound {fork_lb:.4g}  leaves under a clean node {covered.mean():.3f}", flush=True)
    print("level: nodes met (parent not clean) / clean")
    for m in sorted(tot, reverse=True):
        print(f"  2^{m}: {tot[m][0]} / {tot[m][1]}  ({tot[m][1] / max(tot[m][0], 1):.2f})")
    print(f"certificate: exact {exact_ok}/{n}, with the input-magnitude bound at nodes of 16384 / 32768 leaves {weak_ok}/{n}, with the per-4096 bound there {weak2_ok}/{n}; leaves skipped {np.mean(saved):.3f}")

main()
