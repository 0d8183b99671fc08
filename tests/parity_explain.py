"""Explaining a flip-count difference position by position (test infrastructure shared by test_gpu_parity.py and test_gpu_sweeps.py).

bit_flips (decode.cc:546-555) counts payload positions whose LLR SIGN disagrees with the decoded bit.  Where the GPU's count differs
from the oracle's the frame is decoded again through a handle with taps and through the oracle with taps, and every code position
where the two LLR signs differ must be a tie of a known kind (sign / erasure / row tie: DESIGN.md section 3); the counts may differ by
at most the number of such positions."""
import numpy as np

import oracle_lib as O

REL = 1e-5   # north_star tolerance on fp32 intermediates
NAMES = ("status", "best_lane", "sc_start", "symbol_pos", "oper_mode", "call_sign", "n_sync_rejects")
CONS_BITS = 64800    # mode 6 (decode.cc:310): positions beyond are lengthen()'s 9000 (decode.cc:252)


def _explain_flips(pcm_frames, channels, gpu_flips, orc_flips, allow_row_ties=False):
    """every frame of pcm_frames has a flip count that differs: show that sign ties of LLRs within the tolerance account for it"""
    import modem_amd
    dbg = modem_amd.Receiver(device=0, chunk_frames=64, keep_raw_cons=True)
    try:
        for lo in range(0, len(pcm_frames), 64):
            part = pcm_frames[lo:lo + 64]
            batch = np.stack([p if channels == 2 else p[:, None] for p in part])
            out, res = dbg.decode(batch)
            for k, p in enumerate(part):
                g = dbg.tap("LLR", k)[:CONS_BITS]
                gc = dbg.tap("CONS_RAW", k).astype(np.float64)
                oo, orr, tb = O.decode(p if channels == 2 else p[:, None], taps=True)
                o = np.asarray(tb.llr, np.float32)[:CONS_BITS]
                oc = np.asarray(tb.cons_raw, np.float64)[:CONS_BITS // 3]
                assert int(res["bit_flips"][k]) == int(gpu_flips[lo + k]), "the debug handle's LLRs are the default path's"
                assert int(orr.bit_flips) == int(orc_flips[lo + k])
                tol = REL * max(float(np.abs(o).max()), 1e-30)
                differ = np.nonzero((g < 0) != (o < 0))[0]
                sign_tie = (np.abs(g[differ]) <= tol) & (np.abs(o[differ]) <= tol)
                pt = differ // 3                                  # mode 6: three soft bits per point (psk.hh:125-130)
                pw_g, pw_o = (gc[pt] ** 2).sum(axis=1), (oc[pt] ** 2).sum(axis=1)
                erased_g, erased_o = pw_g == 0.0, pw_o == 0.0
                erasure_tie = (erased_g != erased_o) & (np.abs(np.where(erased_g, pw_o, pw_g) - 4.0) <= 4.0 * 10 * REL)
                row_tie = np.zeros(len(differ), bool)
                if allow_row_ties:
                    row = pt // 432                               # mode 6: 432 points per row (decode.cc:306)
                    moved = (np.abs(dbg.tap("YINT", k) - tb.yint[:50]) > 1e-6) | (np.abs(dbg.tap("SLOPE", k) - tb.slope[:50]) > 1e-8)
                    row_tie = moved[row] & (np.abs(g[differ]) <= 500 * tol) & (np.abs(o[differ]) <= 500 * tol)
                fine = sign_tie | erasure_tie | row_tie
                assert fine.all(), \
                    "an LLR sign differs beyond a tie: positions %s gpu %s oracle %s |cons|^2 %s / %s" % (
                        differ[~fine][:4], g[differ][~fine][:4], o[differ][~fine][:4], pw_g[~fine][:4], pw_o[~fine][:4])
                assert abs(int(gpu_flips[lo + k]) - int(orc_flips[lo + k])) <= len(differ)
    finally:
        dbg.close()


