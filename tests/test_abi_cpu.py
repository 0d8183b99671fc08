"""CPU-side checks of the drop-in boundary: libofdmrx.so builds for gfx950, loads, exports every
symbol include/ofdmrx.h declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import modem_amd
    modem_amd.build()
    return modem_amd.load_library()


def _declared():
    src = open(os.path.join(ROOT, "include", "ofdmrx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ofdmrx_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_all_exported(lib):
    import modem_amd.ofdmrx as M
    names = _declared()
    assert len(names) >= 15 and sorted(M.EXPORTS) == names
    for n in names:
        assert hasattr(lib, n), n
    assert lib.ofdmrx_abi_version() == 1


def test_header_compiles_as_plain_c(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "ofdmrx.h"\nint main(void){ofdmrx_config c; ofdmrx_frame_result r; (void)c; (void)r; '
                 'ofdmrx_attempt a; (void)a; return sizeof(ofdmrx_frame_result) == 56 && sizeof(ofdmrx_attempt) == 24 ? 0 : 1;}\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0


def test_result_struct_layout_matches_numpy_and_oracle():
    import modem_amd.ofdmrx as M
    import oracle_lib as O
    assert C.sizeof(M.FrameResult) == M.RESULT_DTYPE.itemsize == 56 == C.sizeof(O.Result)
    for (name, _), (oname, _) in zip(M.FrameResult._fields_, O.Result._fields_):
        assert name == oname
        assert getattr(M.FrameResult, name).offset == getattr(O.Result, name).offset == M.RESULT_DTYPE.fields[name][1]


def test_no_gpu_means_loud_failure_not_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import modem_amd
    with pytest.raises(modem_amd.OfdmRxError, match="no usable HIP device"):
        modem_amd.Receiver()
    assert lib.ofdmrx_strerror(-4).decode().startswith("no usable HIP device")


def test_argument_validation_without_device(lib):
    # NULL handle / bad arguments are API errors (negative), never crashes
    assert lib.ofdmrx_decode_batch(None, None, 0, 1, 10, 20, 1, None, None, None) == -1
    assert lib.ofdmrx_chunk_frames(None) == -1
    bad = (C.c_int32 * 16)()
    h = C.c_void_p()
    import modem_amd.ofdmrx as M
    cfg = M.Config(2, 8000, 8, 0, 0, 0, 1, 0, None)      # wrong ABI version
    assert lib.ofdmrx_create(C.byref(cfg), C.byref(h)) == -1
    cfg = M.Config(1, 22050, 8, 0, 0, 0, 1, 0, None)     # decode.cc:603-605 "Unsupported sample rate."
    assert lib.ofdmrx_create(C.byref(cfg), C.byref(h)) == -5
    cfg = M.Config(1, 48000, 16, 0, 0, 0, 1, 0, None)    # list sizes: 8 (AVX2 build) or 4 (decode.cc:164-169), nothing else
    assert lib.ofdmrx_create(C.byref(cfg), C.byref(h)) == -5
    # frame lengths: 2 x rate silence + (rows + 5) symbols of symbol_len + guard_len (encode.cc:423,441)
    lib.ofdmrx_frame_samples.restype = C.c_long
    assert lib.ofdmrx_frame_samples(8000, 6) == 95200 and lib.ofdmrx_tx_frame_samples(6) == 95200
    assert lib.ofdmrx_frame_samples(48000, 6) == 571200 and lib.ofdmrx_frame_samples(44100, 13) == 1128078
    assert lib.ofdmrx_frame_samples(22050, 6) == -1 and lib.ofdmrx_frame_samples(8000, 5) == -1


def test_product_does_not_reference_the_oracle():
    """the product path must never route through oracle/: no include, no link, no dlopen"""
    csrc = os.path.join(ROOT, "modem_amd")
    for dp, _, fs in os.walk(csrc):
        for f in fs:
            if f.endswith((".hip", ".cpp", ".h", ".py", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "modem_oracle" not in txt and "oracle_lib" not in txt and "liboracle" not in txt, f
    out = subprocess.run(["ldd", os.path.join(csrc, "lib", "libofdmrx.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_host_tables_match_oracle_constants(lib):
    """the product builds its own frozen mask / BCH generator (tables.cpp); they must equal the
    oracle's, which is pinned to the reference's polar_tables.hh by SHA-256"""
    import oracle_lib as O
    exe_src = r'''
#include "tables.h"
#include <cstdio>
int main(){ rx::HostTables t; rx::build_tables(t, 48000);
  fwrite(t.frozen.data(),4,2048,stdout); fwrite(t.genmat_bits.data(),4,71*8,stdout);
  fwrite(t.scramble.data(),1,5380,stdout); fwrite(t.crc32_tab.data(),4,256,stdout);
  fwrite(&t.front.reco,4,1,stdout); fwrite(t.front.imco,4,32,stdout);
  fwrite(t.node_lev.data(),1,8192,stdout); return 0; }
'''
    import tempfile
    d = tempfile.mkdtemp()
    open(os.path.join(d, "m.cpp"), "w").write(exe_src)
    csrc = os.path.join(ROOT, "modem_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-I", csrc, os.path.join(d, "m.cpp"),
                           os.path.join(csrc, "tables.cpp"), "-o", os.path.join(d, "m")])
    raw = subprocess.check_output([os.path.join(d, "m")])
    fr = np.frombuffer(raw[:8192], np.uint32)
    assert (fr == O.frozen(0)).all()
    gm = np.frombuffer(raw[8192:8192 + 71 * 32], np.uint32).reshape(71, 8)
    g = np.zeros((71, 255), np.int8)
    O.lib().orc_bch_genmat(O.ptr(g))
    bits = np.unpackbits(gm.view(np.uint8), bitorder="little").reshape(71, 256)[:, :255]
    assert (bits == g).all()
    scr = np.frombuffer(raw[8192 + 71 * 32:8192 + 71 * 32 + 5380], np.uint8).copy()
    z = np.zeros(5380, np.uint8)
    O.lib().orc_scramble(O.ptr(z), 5380)
    assert (scr == z).all()
    tab = np.frombuffer(raw[8192 + 71 * 32 + 5380:8192 + 71 * 32 + 5380 + 1024], np.uint32)
    # Hilbert<cmplx, 125> taps of the 48 kHz instantiation (decode.cc:172,193)
    hil = np.frombuffer(raw[8192 + 71 * 32 + 5380 + 1024:8192 + 71 * 32 + 5380 + 1024 + 33 * 4], np.float32)
    # uniform-node table of the list decoder: every entry names an aligned node that really is all frozen (low
    # nibble, <= 128 leaves) / all information (high nibble, <= 2048 leaves), and it is the largest such node
    nl = np.frombuffer(raw[8192 + 71 * 32 + 5380 + 1024 + 33 * 4:], np.uint8)
    assert nl.size == 8192
    fz = ((fr[:, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(-1).astype(np.int8)
    for t8 in range(8192):
        t0 = 8 * t8
        for kind, lev, cap in ((1, int(nl[t8]) & 15, 7), (0, int(nl[t8]) >> 4, 11)):
            if lev:
                assert 3 <= lev <= cap and t0 % (1 << lev) == 0 and (fz[t0:t0 + (1 << lev)] == kind).all()
            nxt = max(lev, 2) + 1
            if nxt <= cap and t0 % (1 << nxt) == 0 and not (nxt > 7 and t0 == 0):
                assert not (fz[t0:t0 + (1 << nxt)] == kind).all()
    assert (nl >> 4).max() == 11 and (nl & 15).max() == 7
    reco, imco = np.zeros(1, np.float32), np.zeros(32, np.float32)
    O.lib().orc_hilbert_coeffs_n(125, O.ptr(reco), O.ptr(imco))
    assert hil[0] == reco[0] and (hil[1:33] == imco).all() and imco[30] != 0 and imco[31] == 0
    data = O.payload_for(1)
    crc = 0
    for b in data[:64]:
        crc = (crc >> 8) ^ int(tab[(crc ^ int(b)) & 255])
    assert crc == O.lib().orc_crc32_bytes(0xD419CC15, O.ptr(data), 64)


def test_round6_host_tables_gather_and_clean_node_masks():
    """tables.cpp, round 6.  info_compress: the records message_gather (dev_common.h) takes the systematic message out of a codeword with -
    per code word its mask of unfrozen positions, five "compress" move masks, the message bit its first unfrozen position is and how many it
    holds: replayed here on random codewords, both frozen tables, the result must be the codeword at the unfrozen positions in ascending order
    (decode.cc:254-261).  frozen_t: the frozen bits of a 4096-leaf sub-tree the way lane `lane` of k_sc holds it (bit x = leaf
    s * 4096 + x * 64 + position, position = lane ^ 3 where lane & 4)."""
    import oracle_lib as O
    import tempfile
    exe_src = r'''
#include "tables.h"
#include <cstdio>
int main(){ rx::HostTables t; rx::build_tables(t, 8000);
  fwrite(t.info_compress.data(),4,2*2048*8,stdout); fwrite(t.frozen_t.data(),4,2*16*64*2,stdout); return 0; }
'''
    d = tempfile.mkdtemp()
    open(os.path.join(d, "m.cpp"), "w").write(exe_src)
    csrc = os.path.join(ROOT, "modem_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-I", csrc, os.path.join(d, "m.cpp"),
                           os.path.join(csrc, "tables.cpp"), "-o", os.path.join(d, "m")])
    raw = subprocess.check_output([os.path.join(d, "m")])
    rec = np.frombuffer(raw[:2 * 2048 * 8 * 4], np.uint32).reshape(2, 2048, 8)
    ft = np.frombuffer(raw[2 * 2048 * 8 * 4:], np.uint32).reshape(2, 16, 64, 2)
    rng = np.random.default_rng(66)
    for table in (0, 1):
        fr = O.frozen(table)
        fz = ((fr[:, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(-1).astype(bool)
        code = rng.integers(0, 2 ** 32, 2048, dtype=np.uint64).astype(np.uint32)
        bits = ((code[:, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(-1).astype(np.uint8)
        want = bits[~fz]
        mesg = np.zeros(want.size + 64, np.uint8)
        for w in range(2048):
            mv, m, oc = rec[table, w, :5], int(rec[table, w, 5]), int(rec[table, w, 6])
            assert m == (~int(fr[w])) & 0xffffffff
            x = int(code[w]) & m
            for i in range(5):
                t = x & int(mv[i])
                x = (x ^ t) | (t >> (1 << i))
            off, cnt = oc & 0xffff, oc >> 16
            assert cnt == bin(m).count("1") and x < (1 << cnt)
            for b in range(cnt):
                mesg[off + b] |= (x >> b) & 1
        assert (mesg[:want.size] == want).all() and not mesg[want.size:].any()
        for s_ in range(16):
            for lane in range(64):
                q = lane ^ (3 if lane & 4 else 0)
                word = int(ft[table, s_, lane, 0]) | (int(ft[table, s_, lane, 1]) << 32)
                for x in (0, 1, 31, 32, 63, int(rng.integers(0, 64))):
                    assert ((word >> x) & 1) == int(fz[s_ * 4096 + x * 64 + q])


def test_sc_kernel_pattern_list_covers_both_frozen_tables():
    """k_sc.hip compiles the mixed 16-leaf blocks of the list-1 pass for their frozen pattern (SC_PATTERNS16, straight-line code) and
    walks any other pattern generically: the list must be exactly the set of mixed 16-leaf patterns the sign-following decoder
    meets in the two tables of the reference (frozen_64800_43072, frozen_64512_43072) - top-down, a node that is all frozen (up to
    128 leaves) or all information is decided as a whole and never descended into"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    src = open(os.path.join(ROOT, "modem_amd", "csrc", "k_sc.hip")).read()
    m = re.search(r"#define SC_PATTERNS16\(X\)(.*?)\n__device__", src, re.S)
    listed = sorted(int(x, 16) for x in re.findall(r"X\(0x([0-9a-fA-F]+)u\)", m.group(1)))
    assert len(listed) == len(set(listed))
    met = set()
    for table in (0, 1):
        w = O.frozen(table)
        fz = ((w[:, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(-1).astype(np.uint8)

        def visit(mm, idx):
            s = int(fz[idx:idx + (1 << mm)].sum())
            if (s == (1 << mm) and mm <= 7) or s == 0:
                return
            if mm == 4:
                met.add(sum(int(b) << i for i, b in enumerate(fz[idx:idx + 16])))
                return
            visit(mm - 1, idx)
            visit(mm - 1, idx + (1 << (mm - 1)))
        visit(16, 0)
    assert sorted(met) == listed, (sorted(hex(x) for x in met), [hex(x) for x in listed])
