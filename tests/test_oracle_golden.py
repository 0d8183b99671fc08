"""Oracle vs the golden vectors taken from the real reference headers (psk.hh, polar_tables.hh)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))


class CF(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


def _psk(order):
    L = O.lib()
    hard = getattr(L, "orc_psk%d_hard" % order)
    soft = getattr(L, "orc_psk%d_soft" % order)
    mp = getattr(L, "orc_psk%d_map" % order)
    hard.argtypes = [C.c_void_p, CF]
    soft.argtypes = [C.c_void_p, CF, C.c_float]
    mp.argtypes = [C.c_void_p]
    mp.restype = CF
    return hard, soft, mp


@pytest.mark.parametrize("order", [8, 4])
def test_psk_matches_reference_vectors(order):
    """psk.hh:49-88 / 90-140, bit-exact against vectors produced by the real header"""
    vec = json.load(open(os.path.join(HERE, "golden", "psk_vectors.json")))
    hard, soft, mp = _psk(order)
    nb = 3 if order == 8 else 2
    for v in vec["psk%d" % order]:
        c = CF(float.fromhex(v["re"]), float.fromhex(v["im"]))
        hb = (C.c_float * 3)()
        sb = (C.c_float * 3)()
        hard(hb, c)
        soft(sb, c, float.fromhex(v["precision"]))
        assert [hb[i] for i in range(nb)] == v["hard"]
        assert [float(sb[i]).hex() for i in range(nb)] == v["soft"]
    for v in vec["map%d" % order]:
        b = (C.c_float * 3)(*v["b"])
        r = mp(b)
        assert float(r.re).hex() == v["re"] and float(r.im).hex() == v["im"]


@pytest.mark.parametrize("table,name", [(0, "frozen_64800_43072"), (1, "frozen_64512_43072")])
def test_frozen_table_matches_reference_hash(table, name):
    """freezer.cc:14-32 recipe reproduces polar_tables.hh bit for bit (pinned by SHA-256)"""
    g = json.load(open(os.path.join(HERE, "golden", "polar_tables.json")))[name]
    w = O.frozen(table).astype("<u4")
    assert hashlib.sha256(w.tobytes()).hexdigest() == g["sha256_le_u32"]
    bits = np.unpackbits(w.view(np.uint8), bitorder="little")
    assert int(bits.sum()) == g["frozen_count"]
    assert int(np.flatnonzero(bits == 0)[0]) == g["first_unfrozen"]
    assert int(np.flatnonzero(bits == 1)[-1]) == g["last_frozen"]
    assert [int(x) for x in w[:8]] == g["words_first8"] and int(w[1000]) == g["word_1000"]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="real reference only in the build container")
def test_psk_against_live_reference_build():
    """oracle/_ref (the real psk.hh compiled from /root/reference) on 20k random points"""
    O.build()
    ref = C.CDLL(os.path.join(O.ORACLE_DIR, "_ref", "libref_psk.so"))
    rng = np.random.default_rng(7)
    pts = rng.normal(0, 1, size=(20000, 2)).astype(np.float32)
    pr = rng.uniform(0.1, 500, size=20000).astype(np.float32)
    for order in (8, 4):
        hard, soft, _ = _psk(order)
        rh = getattr(ref, "ref_psk%d_hard" % order)
        rs = getattr(ref, "ref_psk%d_soft" % order)
        for (re, im), p in zip(pts[:4000], pr):
            a, b, c, d = ((C.c_float * 3)() for _ in range(4))
            hard(a, CF(re, im)); rh(b, C.c_float(re), C.c_float(im))
            soft(c, CF(re, im), p); rs(d, C.c_float(re), C.c_float(im), C.c_float(p))
            assert list(a) == list(b) and [x.hex() for x in c] == [x.hex() for x in d]
    ref.ref_frozen.restype = C.POINTER(C.c_uint32)
    for t in (0, 1):
        assert (np.ctypeslib.as_array(ref.ref_frozen(t), shape=(2048,)) == O.frozen(t)).all()
