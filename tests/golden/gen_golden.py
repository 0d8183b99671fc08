#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from REAL reference material.

Runs only in the build container (needs /root/reference).  Sources:
  * psk.hh          -> psk_vectors.json   (hard / soft / map of PhaseShiftKeying<8|4>,
                       evaluated through oracle/_ref/libref_psk.so which #includes the real header)
  * polar_tables.hh -> polar_tables.json  (SHA-256 + popcount + first/last landmarks of both masks)
The fixtures are DATA (inputs + expected outputs); no reference source text is stored.
"""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_psk.so"))
    ref.ref_frozen.restype = C.POINTER(C.c_uint32)
    f3 = C.c_float * 3
    f2 = C.c_float * 2
    rng = np.random.Generator(np.random.PCG64(20211))
    pts = rng.normal(0, 0.8, size=(192, 2)).astype(np.float32)
    # edge cases: axes, diagonals (|re| == |im| ties), zero, negative zero
    edges = [(0, 0), (-0.0, 0.0), (1, 0), (0, 1), (-1, 0), (0, -1), (0.5, 0.5), (-0.5, 0.5), (0.5, -0.5),
             (-0.5, -0.5), (0.92387953, 0.38268343), (0.38268343, 0.92387953), (1e-30, -1e-30), (3, -4)]
    pts = np.concatenate([np.asarray(edges, dtype=np.float32), pts])
    precs = rng.uniform(0.5, 200.0, size=len(pts)).astype(np.float32)
    out = {"psk8": [], "psk4": [], "map8": [], "map4": []}
    for (re, im), p in zip(pts, precs):
        for order, hard, soft, nb in ((8, ref.ref_psk8_hard, ref.ref_psk8_soft, 3), (4, ref.ref_psk4_hard, ref.ref_psk4_soft, 2)):
            hb, sb = f3(), f3()
            hard(hb, C.c_float(re), C.c_float(im))
            soft(sb, C.c_float(re), C.c_float(im), C.c_float(p))
            out["psk%d" % order].append({
                "re": float(re).hex(), "im": float(im).hex(), "precision": float(p).hex(),
                "hard": [float(hb[i]) for i in range(nb)], "soft": [float(sb[i]).hex() for i in range(nb)]})
    for bits in range(8):
        b = f3(*[1.0 - 2.0 * ((bits >> i) & 1) for i in range(3)])
        o = f2()
        ref.ref_psk8_map(o, b)
        out["map8"].append({"b": list(b), "re": float(o[0]).hex(), "im": float(o[1]).hex()})
    for bits in range(4):
        b = f3(*[1.0 - 2.0 * ((bits >> i) & 1) for i in range(2)], 0.0)
        o = f2()
        ref.ref_psk4_map(o, b)
        out["map4"].append({"b": list(b)[:2], "re": float(o[0]).hex(), "im": float(o[1]).hex()})
    with open(os.path.join(HERE, "psk_vectors.json"), "w") as f:
        json.dump(out, f, indent=0)

    tabs = {}
    for t, name in ((0, "frozen_64800_43072"), (1, "frozen_64512_43072")):
        w = np.ctypeslib.as_array(ref.ref_frozen(t), shape=(2048,)).astype("<u4")
        bits = np.unpackbits(w.view(np.uint8), bitorder="little")
        unf = np.flatnonzero(bits == 0)
        frz = np.flatnonzero(bits == 1)
        tabs[name] = {
            "sha256_le_u32": hashlib.sha256(w.tobytes()).hexdigest(),
            "frozen_count": int(bits.sum()), "unfrozen_count": int((bits == 0).sum()),
            "first_unfrozen": int(unf[0]), "last_frozen": int(frz[-1]),
            "words_first8": [int(x) for x in w[:8]], "words_last8": [int(x) for x in w[-8:]],
            "word_1000": int(w[1000]), "word_1500": int(w[1500]),
        }
    with open(os.path.join(HERE, "polar_tables.json"), "w") as f:
        json.dump(tabs, f, indent=1)
    print("wrote psk_vectors.json (%d points) and polar_tables.json" % len(pts))


if __name__ == "__main__":
    sys.exit(main())
