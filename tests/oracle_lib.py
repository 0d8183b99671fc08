"""ctypes binding of oracle/libmodem_oracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference (decode.cc / encode.cc);
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libmodem_oracle.so")

FMT_S16, FMT_U8, FMT_F32 = 0, 1, 2
DATA_BYTES = 5380
CODE_LEN = 65536
STATUS = ["OK", "NO_SYNC", "OSD_ERROR", "HEADER_CRC", "BAD_MODE", "BAD_CALLSIGN", "PAYLOAD_CRC"]


class Result(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("symbol_pos", C.c_int32), ("sc_start", C.c_int64),
        ("cfo_rad", C.c_float), ("cfo_fine", C.c_float), ("sfo_slope", C.c_float),
        ("oper_mode", C.c_int32), ("call_sign", C.c_uint64), ("best_lane", C.c_int32),
        ("bit_flips", C.c_int32), ("esn0_db_last", C.c_float), ("n_sync_rejects", C.c_int32),
    ]


class Taps(C.Structure):
    _fields_ = [
        ("hdr_soft", C.c_void_p), ("cons_raw", C.c_void_p), ("cons_rot", C.c_void_p),
        ("slope", C.c_void_p), ("yint", C.c_void_p), ("precision", C.c_void_p),
        ("llr", C.c_void_p), ("metric", C.c_void_p), ("lane_mesg", C.c_void_p),
    ]


class Mode(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("oper_mode", "cons_cols", "cons_rows", "mod_bits", "cons_bits", "mesg_bits",
                 "cons_cnt", "table", "band_width")]


def build(force=False):
    """compile the oracle (and oracle/_ref when /root/reference is present)"""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-j8", "all"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and not os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libref_psk.so")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_encode_pcm.restype = C.c_size_t
        L.orc_encode_pcm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.orc_encode_pcm_rate.restype = C.c_size_t
        L.orc_encode_pcm_rate.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.orc_frame_samples.restype = C.c_size_t
        L.orc_frame_samples.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_decode_rate.restype = C.c_int
        L.orc_decode_rate.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.POINTER(Result), C.POINTER(Taps)]
        L.orc_decode.restype = C.c_int
        L.orc_decode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.POINTER(Result), C.POINTER(Taps)]
        L.orc_decode_cf.restype = C.c_int
        L.orc_decode_cf.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.POINTER(Result), C.POINTER(Taps)]
        L.orc_decode_batch.restype = C.c_int
        L.orc_decode_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_int]
        L.orc_decode_batch_sc.restype = C.c_int
        L.orc_decode_batch_sc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_polar_list_decode.restype = C.c_int
        L.orc_polar_list_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_polar_lane_mesg.restype = C.c_int
        L.orc_polar_lane_mesg.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_polar_sc_path.restype = C.c_int
        L.orc_polar_sc_path.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_frozen_get.restype = C.POINTER(C.c_uint32)
        L.orc_frozen_get.argtypes = [C.c_int]
        L.orc_fft.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_front_end.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p]
        L.orc_theil_sen.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_osd_decode.restype = C.c_int
        L.orc_osd_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_bch_genmat.argtypes = [C.c_void_p]
        L.orc_bch_encode.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_crc16_u64.restype = C.c_uint16
        L.orc_crc16_u64.argtypes = [C.c_uint16, C.c_uint64]
        L.orc_crc32_bytes.restype = C.c_uint32
        L.orc_crc32_bytes.argtypes = [C.c_uint32, C.c_void_p, C.c_int]
        L.orc_base37_encode.restype = C.c_longlong
        L.orc_base37_encode.argtypes = [C.c_char_p]
        L.orc_scramble.argtypes = [C.c_void_p, C.c_int]
        L.orc_mode_lookup.restype = C.c_int
        L.orc_mode_lookup.argtypes = [C.c_int, C.POINTER(Mode)]
        L.orc_polar_sysenc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_polar_enc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_chan_awgn.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_uint64, C.c_uint64]
        L.orc_chan_cfo.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_int]
        L.orc_chan_sfo.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float]
        L.orc_chan_multipath.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_quantise.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.orc_hilbert_coeffs.argtypes = [C.c_void_p, C.c_void_p]
        for name in ("orc_psk8_hard", "orc_psk4_hard"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_float * 2]
        _lib = L
    return _lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def payload_for(seed, count=1):
    """deterministic 5380-byte payloads (counter PRNG, seed = frame index)"""
    rng = np.random.Generator(np.random.PCG64(1000 + int(seed)))
    return rng.integers(0, 256, size=count * DATA_BYTES, dtype=np.uint8)


def frame_len(count=1, rows=50):
    return 2 * 8000 + (2 + count * (3 + rows)) * 1440


def encode_pcm(payload, bits=16, channels=1, freq_off=2000, call_sign="ANONYMOUS", mode=6, rate=8000):
    """orc_encode_pcm_rate -> numpy int16/uint8 array of shape [samples, channels]"""
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    count = payload.size // DATA_BYTES
    n = int(lib().orc_frame_samples(rate, mode, count))
    assert n > 0
    out = np.zeros((n, channels), dtype=np.int16 if bits == 16 else np.uint8)
    got = lib().orc_encode_pcm_rate(rate, ptr(out), bits, channels, ptr(payload), count, freq_off,
                                    call_sign.encode(), mode)
    assert got == n, (got, n)
    return out


def pcm_to_cf(pcm):
    """what ReadWAV does to a 2-channel stream: scale, take (re, im)"""
    if pcm.dtype == np.int16:
        f = pcm.astype(np.float32) / np.float32(32767)
    elif pcm.dtype == np.uint8:
        f = (pcm.astype(np.float32) - np.float32(128)) / np.float32(127)
    else:
        f = pcm.astype(np.float32)
    return np.ascontiguousarray(f)


def quantise(z, bits=16, channels=2):
    z = np.ascontiguousarray(z, dtype=np.float32)
    n = z.shape[0]
    out = np.zeros((n, channels), dtype=np.int16 if bits == 16 else np.uint8)
    lib().orc_quantise(ptr(out), bits, channels, ptr(z), n)
    return out


class TapBuffers:
    def __init__(self, L=8):
        self.hdr_soft = np.zeros(255, np.int8)
        self.cons_raw = np.zeros((32400, 2), np.float32)
        self.cons_rot = np.zeros((32400, 2), np.float32)
        self.slope = np.zeros(126, np.float32)
        self.yint = np.zeros(126, np.float32)
        self.precision = np.zeros(126, np.float32)
        self.llr = np.zeros(CODE_LEN, np.float32)
        self.metric = np.zeros(8, np.float32)
        self.lane_mesg = np.zeros((8, 5512), np.uint8)
        self.c = Taps(*[ptr(getattr(self, n)) for n, _ in Taps._fields_])


def decode(pcm, skip=0, list_size=8, descramble=1, taps=False, rate=8000):
    """orc_decode_rate on a [samples, channels] int16/uint8/float32 array"""
    pcm = np.ascontiguousarray(pcm)
    fmt = {np.dtype(np.int16): FMT_S16, np.dtype(np.uint8): FMT_U8, np.dtype(np.float32): FMT_F32}[pcm.dtype]
    channels = pcm.shape[1] if pcm.ndim == 2 else 1
    out = np.zeros(DATA_BYTES, np.uint8)
    res = Result()
    tb = TapBuffers() if taps else None
    lib().orc_decode_rate(rate, ptr(pcm), fmt, channels, pcm.shape[0], skip, list_size, descramble, ptr(out),
                          C.byref(res), C.byref(tb.c) if tb else None)
    return (out, res, tb) if taps else (out, res)


def frozen(table=0):
    p = lib().orc_frozen_get(table)
    return np.ctypeslib.as_array(p, shape=(2048,)).copy()


def impair(pcm2, noise_db=None, cfo_hz=0.0, sfo_ppm=0.0, multipath=None, seed=1, frame=0, bits=16, rate=8000):
    """2-channel analytic stream -> impairment chain (build-owned models) -> re-quantised"""
    z = pcm_to_cf(pcm2)
    n = z.shape[0]
    if multipath is not None:
        delays = np.ascontiguousarray([d for d, _ in multipath], dtype=np.int32)
        gains = np.ascontiguousarray([[g.real, g.imag] for _, g in multipath], dtype=np.float32)
        o = np.zeros_like(z)
        lib().orc_chan_multipath(ptr(o), ptr(z), n, ptr(delays), ptr(gains), len(multipath))
        z = o
    if cfo_hz:
        lib().orc_chan_cfo(ptr(z), n, cfo_hz, rate)
    if sfo_ppm:
        o = np.zeros_like(z)
        lib().orc_chan_sfo(ptr(o), ptr(z), n, sfo_ppm)
        z = o
    if noise_db is not None:
        lib().orc_chan_awgn(ptr(z), n, noise_db, seed, frame)
    return quantise(z, bits, 2)


def polar_lane_mesg(llr, L=8, table=0):
    """oracle D9 + systematic(): per-lane systematic message bits [L, 5476] and metrics [L]"""
    llr = np.ascontiguousarray(llr, dtype=np.float32)
    fr = frozen(table)
    mesg = np.zeros((L, 5476), np.uint8)
    metric = np.zeros(L, np.float32)
    lib().orc_polar_lane_mesg(ptr(llr), ptr(fr), 16, L, ptr(mesg), 5476, ptr(metric))
    return mesg, metric


def polar_sc_path(llr, fr=None, level=16):
    """the sign-following path alone (oracle/polar.c: orc_polar_sc_path): codeword bits [N] (1 = -1), metric, min_fork"""
    llr = np.ascontiguousarray(llr, dtype=np.float32)
    fr = frozen(0) if fr is None else np.ascontiguousarray(fr, dtype=np.uint32)
    hard = np.zeros(1 << level, np.int8)
    metric, fork = C.c_float(), C.c_float()
    lib().orc_polar_sc_path(ptr(llr), ptr(fr), level, ptr(hard), C.byref(metric), C.byref(fork))
    return (hard < 0).astype(np.uint8), np.float32(metric.value), np.float32(fork.value)


def polar_list_decode(llr, fr, level, L):
    """orc_polar_list_decode on any code length: per-lane re-encoded codewords [L, N] (1 = -1) and metrics [L]"""
    llr = np.ascontiguousarray(llr, dtype=np.float32)
    fr = np.ascontiguousarray(fr, dtype=np.uint32)
    N = 1 << level
    mesg = np.zeros((N, L), np.int8)
    metric = np.zeros(L, np.float32)
    count = lib().orc_polar_list_decode(ptr(metric), ptr(mesg), ptr(llr), ptr(fr), level, L)
    code = np.zeros((L, N), np.int8)
    for k in range(L):
        u = np.ascontiguousarray(mesg[:count, k])
        lib().orc_polar_enc(ptr(code[k]), ptr(u), ptr(fr), level)
    return (code < 0).astype(np.uint8), metric


def theil_sen(y):
    y = np.ascontiguousarray(y, dtype=np.float32)
    x = (np.arange(y.size) - y.size // 2).astype(np.float32)
    s, i = C.c_float(), C.c_float()
    lib().orc_theil_sen(ptr(x), ptr(y), y.size, C.byref(s), C.byref(i))
    return s.value, i.value


def osd(soft):
    g = np.zeros((71, 255), np.int8)
    lib().orc_bch_genmat(ptr(g))
    soft = np.ascontiguousarray(soft, dtype=np.int8)
    hard = np.zeros(32, np.uint8)
    u = lib().orc_osd_decode(ptr(hard), ptr(soft), ptr(g))
    return hard, u
