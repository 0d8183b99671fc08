"""ctypes mirror of include/ofdmrx.h (the drop-in boundary for decode.cc's Decoder seam)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")

FMT_S16, FMT_U8, FMT_F32 = 0, 1, 2
PAYLOAD_BYTES = 5380
CODE_LEN = 65536
FRAME_SAMPLES = 95200
MESG_BYTES = 5476
STATUS_NAMES = ["OK", "NO_SYNC", "OSD_ERROR", "HEADER_CRC", "BAD_MODE", "BAD_CALLSIGN", "PAYLOAD_CRC"]
TAPS = dict(HDR_SOFT=1, CONS_RAW=2, CONS_ROT=3, SLOPE=4, YINT=5, PRECISION=6, LLR=7, METRIC=8, LANE_MESG=9, ANALYTIC=10)
STAGES = ["front", "sync", "header", "demod", "theilsen", "llr", "polar", "finish", "total"]

# every symbol include/ofdmrx.h declares
EXPORTS = [
    "ofdmrx_abi_version", "ofdmrx_abi_minor", "ofdmrx_strerror", "ofdmrx_create", "ofdmrx_destroy", "ofdmrx_decode_batch",
    "ofdmrx_decode_batch_device", "ofdmrx_synchronize", "ofdmrx_get_timing", "ofdmrx_chunk_frames", "ofdmrx_last_chunk_first_frame", "ofdmrx_list_decoded_frames", "ofdmrx_sc_decided_frames", "ofdmrx_get_sc_timing", "ofdmrx_set_esn0_rows", "ofdmrx_set_attempt_log",
    "ofdmrx_debug_dump", "ofdmrx_debug_polar", "ofdmrx_debug_sc_path", "ofdmrx_debug_decode_cons", "ofdmrx_debug_theil_sen", "ofdmrx_debug_osd", "ofdmrx_debug_fft",
    "ofdmrx_util_awgn_tile", "ofdmrx_util_channel", "ofdmrx_frame_samples", "ofdmrx_tx_frame_samples",
    "ofdmrx_tx_encode_device", "ofdmrx_stream_samples", "ofdmrx_tx_encode_stream_device", "ofdmrx_tx_encode_stream",
    "ofdmrx_callsign_value",
]


class OfdmRxError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("sample_rate", C.c_int32), ("list_size", C.c_int32),
                ("device", C.c_int32), ("chunk_frames", C.c_int32), ("max_samples", C.c_int32),
                ("descramble", C.c_int32), ("flags", C.c_int32), ("stream", C.c_void_p)]


class FrameResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("symbol_pos", C.c_int32), ("sc_start", C.c_int64),
                ("cfo_rad", C.c_float), ("cfo_fine", C.c_float), ("sfo_slope", C.c_float),
                ("oper_mode", C.c_int32), ("call_sign", C.c_uint64), ("best_lane", C.c_int32),
                ("bit_flips", C.c_int32), ("esn0_db_last", C.c_float), ("n_sync_rejects", C.c_int32)]


RESULT_DTYPE = np.dtype([("status", "<i4"), ("symbol_pos", "<i4"), ("sc_start", "<i8"), ("cfo_rad", "<f4"),
                         ("cfo_fine", "<f4"), ("sfo_slope", "<f4"), ("oper_mode", "<i4"), ("call_sign", "<u8"),
                         ("best_lane", "<i4"), ("bit_flips", "<i4"), ("esn0_db_last", "<f4"),
                         ("n_sync_rejects", "<i4")], align=True)
assert RESULT_DTYPE.itemsize == C.sizeof(FrameResult)
# ofdmrx_attempt: one preamble of a frame's SKIP loop (decode.cc:390-448)
ATTEMPT_DTYPE = np.dtype([("status", "<i4"), ("symbol_pos", "<i4"), ("cfo_rad", "<f4"), ("oper_mode", "<i4"), ("call_sign", "<u8")], align=True)
assert ATTEMPT_DTYPE.itemsize == 24
MAX_SKIP = 64


class Channel(C.Structure):
    _fields_ = [("cfo_hz", C.c_float), ("sfo_ppm", C.c_float), ("ntaps", C.c_int32), ("delays", C.c_int32 * 8),
                ("gains_re", C.c_float * 8), ("gains_im", C.c_float * 8)]


class Timing(C.Structure):
    _fields_ = [("ms", C.c_float * 9), ("launches", C.c_int32 * 9)]


def lib_path():
    """MODEM_AMD_LIB selects an alternative build of the same library (A/B runs of kernel variants)"""
    return os.environ.get("MODEM_AMD_LIB") or os.path.join(HERE, "lib", "libofdmrx.so")


def build(force=False):
    """compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)"""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CSRC, "-j8", "all"], stdout=subprocess.DEVNULL)
    return lib_path()


_LIB = None


def load_library():
    """dlopen libofdmrx.so; fails loudly when the HIP extension has not been built"""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch bundles its own HIP runtime (same SONAME libamdhip64.so.7).  Two HIP runtimes in one
    # process cannot both see the GPU, so when torch is importable it is loaded FIRST and libofdmrx
    # binds to the runtime already in the process.  MODEM_AMD_NO_TORCH=1 skips this (pure C ABI use).
    if not os.environ.get("MODEM_AMD_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    path = lib_path()
    if not os.path.exists(path):
        raise OfdmRxError("libofdmrx.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "-- the receive path has no CPU fallback" % path)
    L = C.CDLL(path)
    L.ofdmrx_abi_version.restype = C.c_int
    L.ofdmrx_strerror.restype = C.c_char_p
    L.ofdmrx_strerror.argtypes = [C.c_int]
    L.ofdmrx_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.ofdmrx_destroy.argtypes = [C.c_void_p]
    L.ofdmrx_destroy.restype = None
    batch = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ofdmrx_decode_batch.argtypes = batch
    L.ofdmrx_decode_batch_device.argtypes = batch
    L.ofdmrx_synchronize.argtypes = [C.c_void_p]
    L.ofdmrx_get_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
    L.ofdmrx_chunk_frames.argtypes = [C.c_void_p]
    L.ofdmrx_set_esn0_rows.argtypes = [C.c_void_p, C.c_void_p]
    L.ofdmrx_set_attempt_log.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ofdmrx_list_decoded_frames.argtypes = [C.c_void_p]
    L.ofdmrx_list_decoded_frames.restype = C.c_longlong
    L.ofdmrx_last_chunk_first_frame.argtypes = [C.c_void_p]
    L.ofdmrx_last_chunk_first_frame.restype = C.c_longlong
    L.ofdmrx_sc_decided_frames.argtypes = [C.c_void_p]
    L.ofdmrx_sc_decided_frames.restype = C.c_longlong
    L.ofdmrx_get_sc_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    L.ofdmrx_debug_sc_path.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ofdmrx_debug_dump.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
    L.ofdmrx_debug_polar.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.ofdmrx_debug_decode_cons.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ofdmrx_debug_theil_sen.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    L.ofdmrx_debug_osd.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.ofdmrx_debug_fft.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    L.ofdmrx_util_awgn_tile.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                                        C.c_float, C.c_uint64, C.c_uint64]
    L.ofdmrx_util_channel.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(Channel)]
    L.ofdmrx_tx_frame_samples.restype = C.c_long
    L.ofdmrx_tx_frame_samples.argtypes = [C.c_int]
    L.ofdmrx_frame_samples.restype = C.c_long
    L.ofdmrx_frame_samples.argtypes = [C.c_int, C.c_int]
    L.ofdmrx_stream_samples.restype = C.c_long
    L.ofdmrx_stream_samples.argtypes = [C.c_int, C.c_int, C.c_int]
    L.ofdmrx_tx_encode_stream_device.restype = C.c_int
    L.ofdmrx_tx_encode_stream_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_char_p,
                                                 C.c_int, C.c_int, C.c_void_p]
    L.ofdmrx_tx_encode_stream.restype = C.c_int
    L.ofdmrx_tx_encode_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int,
                                          C.c_void_p]
    L.ofdmrx_callsign_value.restype = C.c_longlong
    L.ofdmrx_callsign_value.argtypes = [C.c_char_p]
    L.ofdmrx_tx_encode_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_void_p]
    _LIB = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Receiver:
    """Batch counterpart of `new Decoder<float, Complex<float>, 8000>(out, pcm, skip)` (decode.cc:592).

    stream: a hipStream_t handle (int) to run on.  None / 0 (note: torch's DEFAULT stream has handle 0) makes the
    library create its own non-blocking stream - then synchronise explicitly before sharing device buffers with
    other libraries, or pass a non-default stream (torch.cuda.Stream().cuda_stream) and use it on both sides.

    decode(pcm) takes raw PCM frames [n_frames, samples, channels] (int16 / uint8 / float32, what
    DSP::ReadWAV would deliver) and returns (payload[n_frames, 5380] uint8, results structured array).
    """

    def __init__(self, device=0, chunk_frames=0, max_samples=0, descramble=True, keep_raw_cons=False, stream=None,
                 sample_rate=8000, list_size=8, scl_always=False, no_sc=False, two_lanes=False):
        self._lib = load_library()
        if self._lib.ofdmrx_abi_version() != 1:
            raise OfdmRxError("ABI mismatch")
        self.sample_rate = int(sample_rate)
        cfg = Config(1, self.sample_rate, int(list_size), device, chunk_frames, max_samples, 1 if descramble else 0,
                     (1 if keep_raw_cons else 0) | (2 if scl_always else 0) | (4 if no_sc else 0) | (8 if two_lanes else 0), stream)
        self._h = C.c_void_p()
        self._check(self._lib.ofdmrx_create(C.byref(cfg), C.byref(self._h)))

    def _check(self, r):
        if r != 0:
            raise OfdmRxError("ofdmrx: %s (%d)" % (self._lib.ofdmrx_strerror(r).decode(), r))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.ofdmrx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def chunk_frames(self):
        return self._lib.ofdmrx_chunk_frames(self._h)

    @staticmethod
    def _fmt(dtype):
        return {np.dtype(np.int16): FMT_S16, np.dtype(np.uint8): FMT_U8, np.dtype(np.float32): FMT_F32}[np.dtype(dtype)]

    def decode(self, pcm, skip=None, esn0_rows=False, attempts=False):
        """-> (payload, results[, esn0 rows][, (attempt log [n, 65], counts [n])])"""
        pcm = np.ascontiguousarray(pcm)
        if pcm.ndim == 2:
            pcm = pcm[None]
        n, spf, ch = pcm.shape
        stride = spf * ch * pcm.dtype.itemsize
        if stride % 4:
            pad = np.zeros((n, (stride + 3) // 4 * 4), np.uint8)
            pad[:, :stride] = pcm.reshape(n, -1).view(np.uint8)
            buf, stride = pad, pad.shape[1]
        else:
            buf = pcm
        out = np.zeros((n, PAYLOAD_BYTES), np.uint8)
        res = np.zeros(n, RESULT_DTYPE)
        sk = None if skip is None else np.ascontiguousarray(skip, dtype=np.int32)
        rows = np.zeros((n, 126), np.float32) if esn0_rows else None
        if esn0_rows:
            self._check(self._lib.ofdmrx_set_esn0_rows(self._h, _ptr(rows)))
        if attempts:
            alog, acnt = np.zeros((n, MAX_SKIP + 1), ATTEMPT_DTYPE), np.zeros(n, np.int32)
            self._check(self._lib.ofdmrx_set_attempt_log(self._h, _ptr(alog), _ptr(acnt)))
        try:
            self._check(self._lib.ofdmrx_decode_batch(self._h, _ptr(buf), self._fmt(pcm.dtype), ch, spf, stride, n,
                                                      _ptr(sk) if sk is not None else None, _ptr(out), _ptr(res)))
        finally:
            if esn0_rows:
                self._lib.ofdmrx_set_esn0_rows(self._h, None)
            if attempts:
                self._lib.ofdmrx_set_attempt_log(self._h, None, None)
        ret = (out, res) + ((rows,) if esn0_rows else ()) + (((alog, acnt),) if attempts else ())
        return ret

    def set_esn0_rows(self, d_rows):
        """device pointer (int) to n x 126 floats for the decode_device calls that follow, or None"""
        self._check(self._lib.ofdmrx_set_esn0_rows(self._h, d_rows))

    def decode_device(self, d_samples, fmt, channels, spf, stride, n, d_payload, d_results, d_skip=None):
        """device pointers (ints); asynchronous on the handle's stream"""
        self._check(self._lib.ofdmrx_decode_batch_device(self._h, d_samples, fmt, channels, spf, stride, n,
                                                         d_skip, d_payload, d_results))

    def list_decoded_frames(self):
        """frames of the last decode call the syndrome certificate left to the list decoder (-1: certificate off)"""
        return int(self._lib.ofdmrx_list_decoded_frames(self._h))

    def last_chunk_first_frame(self):
        """index, in the last decode call, of the first frame of the chunk the taps belong to"""
        return int(self._lib.ofdmrx_last_chunk_first_frame(self._h))

    def sc_decided_frames(self):
        """frames of the last decode call finished by the list-1 pass (-1: that pass is off)"""
        return int(self._lib.ofdmrx_sc_decided_frames(self._h))

    def sc_timing(self):
        ms, n = C.c_float(), C.c_int32()
        self._check(self._lib.ofdmrx_get_sc_timing(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def synchronize(self):
        self._check(self._lib.ofdmrx_synchronize(self._h))

    def timing(self):
        t = Timing()
        self._check(self._lib.ofdmrx_get_timing(self._h, C.byref(t)))
        return {s: (t.ms[i], t.launches[i]) for i, s in enumerate(STAGES)}

    def tap(self, name, frame, samples=None, cons_cnt=21600, rows=50):
        """stage tap of the last resident chunk; cons_cnt / rows default to mode 6 (other modes: up to 32400 / 126)"""
        shapes = dict(HDR_SOFT=((255,), np.int8), CONS_RAW=((cons_cnt, 2), np.float32), CONS_ROT=((cons_cnt, 2), np.float32),
                      SLOPE=((rows,), np.float32), YINT=((rows,), np.float32), PRECISION=((rows,), np.float32),
                      LLR=((CODE_LEN,), np.float32), METRIC=((8,), np.float32), LANE_MESG=((8, MESG_BYTES), np.uint8),
                      ANALYTIC=((samples or 0, 2), np.float32))
        shape, dt = shapes[name]
        a = np.zeros(shape, dt)
        self._check(self._lib.ofdmrx_debug_dump(self._h, TAPS[name], frame, _ptr(a), a.nbytes))
        return a

    # ---- single-stage entry points (parity tests)
    def polar(self, llr):
        llr = np.ascontiguousarray(llr, dtype=np.float32).reshape(-1, CODE_LEN)
        n = llr.shape[0]
        mesg = np.zeros((n, 8, MESG_BYTES), np.uint8)
        metric = np.zeros((n, 8), np.float32)
        self._check(self._lib.ofdmrx_debug_polar(self._h, _ptr(llr), n, _ptr(mesg), _ptr(metric)))
        return mesg, metric

    def sc_path(self, llr, modes=None):
        """k_sc alone: codeword bits [n, 65536], hard decisions of the LLRs [n, 65536], metric, min_fork, rule [n]; modes: the
        operation mode of every vector (its frozen table), None = all mode 6"""
        llr = np.ascontiguousarray(llr, dtype=np.float32).reshape(-1, CODE_LEN)
        n = llr.shape[0]
        mode = None if modes is None else _ptr(np.ascontiguousarray(np.broadcast_to(np.asarray(modes, np.int32), (n,))))
        cw = np.zeros((n, CODE_LEN // 8), np.uint8)
        hd = np.zeros((n, CODE_LEN // 8), np.uint8)
        metric, fork, ok = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
        self._check(self._lib.ofdmrx_debug_sc_path(self._h, _ptr(llr), n, mode, _ptr(cw), _ptr(hd), _ptr(metric), _ptr(fork), _ptr(ok)))
        return np.unpackbits(cw, axis=1, bitorder="little"), np.unpackbits(hd, axis=1, bitorder="little"), metric, fork, ok

    def decode_cons(self, cons, use_cert=True):
        """rotated constellation rows (n x 21600 complex64, mode 6) -> payloads, results, who finished each frame (1 the syndrome
        certificate, 2 the list-1 pass, 0 the list decoder).  use_cert: 0 / False list decoder only, 1 / True syndrome certificate
        first, 2 the default chain (certificate, list-1 pass, list decoder), 3 list-1 pass then list decoder"""
        cons = np.ascontiguousarray(cons, dtype=np.complex64).reshape(-1, 21600)
        n = cons.shape[0]
        out = np.zeros((n, 5380), np.uint8)
        res = np.zeros(n, RESULT_DTYPE)
        cert = np.zeros(n, np.int32)
        self._check(self._lib.ofdmrx_debug_decode_cons(self._h, _ptr(cons), n, int(use_cert), _ptr(out), _ptr(res), _ptr(cert)))
        return out, res, cert

    def theil_sen(self, y):
        y = np.ascontiguousarray(y, dtype=np.float32)
        rows, cols = y.shape
        s, i = np.zeros(rows, np.float32), np.zeros(rows, np.float32)
        self._check(self._lib.ofdmrx_debug_theil_sen(self._h, _ptr(y), rows, cols, _ptr(s), _ptr(i)))
        return s, i

    def osd(self, soft):
        soft = np.ascontiguousarray(soft, dtype=np.int8).reshape(-1, 255)
        n = soft.shape[0]
        hard, uniq = np.zeros((n, 32), np.uint8), np.zeros(n, np.int32)
        self._check(self._lib.ofdmrx_debug_osd(self._h, _ptr(soft), n, _ptr(hard), _ptr(uniq)))
        return hard, uniq

    def fft(self, x, sign=-1):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        if x.ndim == 1:
            x = x[None]
        out = np.zeros_like(x)
        self._check(self._lib.ofdmrx_debug_fft(self._h, _ptr(x), x.shape[0], x.shape[1], sign, _ptr(out)))
        return out

    def awgn_tile(self, d_base, n_base, d_out, n_out, spf, noise_db, seed, first_frame=0):
        self._check(self._lib.ofdmrx_util_awgn_tile(self._h, d_base, n_base, d_out, n_out, spf, noise_db, seed, first_frame))

    def channel(self, d_in, d_out, n, spf, cfo_hz=0.0, sfo_ppm=0.0, multipath=()):
        """multipath | cfo | sfo on n device-resident 2-channel int16 frames (multipath = [(delay, complex gain), ...])"""
        ch = Channel()
        ch.cfo_hz, ch.sfo_ppm, ch.ntaps = cfo_hz, sfo_ppm, len(multipath)
        for i, (d, g) in enumerate(multipath):
            ch.delays[i], ch.gains_re[i], ch.gains_im[i] = int(d), float(complex(g).real), float(complex(g).imag)
        self._check(self._lib.ofdmrx_util_channel(self._h, d_in, d_out, n, spf, C.byref(ch)))

    def tx_frame_samples(self, mode=6):
        return int(self._lib.ofdmrx_frame_samples(self.sample_rate, mode))

    def encode_stream(self, payloads, mode=6, freq_off=2000, call_sign="ANONYMOUS", channels=1, bits=16):
        """host convenience (the `encode` CLI's call): payloads [count, 5380] uint8 -> one PCM stream [samples, channels]"""
        payloads = np.ascontiguousarray(payloads, dtype=np.uint8).reshape(-1, PAYLOAD_BYTES)
        count = payloads.shape[0]
        n = int(self._lib.ofdmrx_stream_samples(self.sample_rate, mode, count))
        if n < 0:
            raise OfdmRxError("bad mode / count")
        pcm = np.zeros((n, channels), np.int16 if bits == 16 else np.uint8)
        self._check(self._lib.ofdmrx_tx_encode_stream(self._h, _ptr(payloads), count, mode, freq_off, call_sign.encode(),
                                                      channels, bits, _ptr(pcm)))
        return pcm

    def tx_encode(self, d_payload, n, d_pcm, mode=6, freq_off=2000, call_sign="ANONYMOUS", channels=2):
        """device transmitter: n x 5380 payload bytes -> n x tx_frame_samples(mode) x channels int16 (device pointers)"""
        self._check(self._lib.ofdmrx_tx_encode_device(self._h, d_payload, n, mode, freq_off, call_sign.encode(), channels, d_pcm))
