"""modem_amd -- MI355X-native OFDM receive path (aicodix/modem `decode`, mode 6 @ 8 kHz).

The product is libofdmrx.so (hand-written HIP kernels for gfx950 behind the C ABI of
include/ofdmrx.h); this package is the thin Python mirror of that ABI.  There is no CPU
fallback: importing works anywhere, creating a Receiver needs the built library and a GPU.
"""
from .ofdmrx import (  # noqa: F401
    FMT_F32, FMT_S16, FMT_U8, FrameResult, OfdmRxError, Receiver, STATUS_NAMES, build, lib_path, load_library,
)

__all__ = ["Receiver", "FrameResult", "OfdmRxError", "build", "load_library", "lib_path",
           "FMT_S16", "FMT_U8", "FMT_F32", "STATUS_NAMES"]
