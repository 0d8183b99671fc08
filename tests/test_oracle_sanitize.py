"""ASan + UBSan build of the CPU oracle (oracle/Makefile `san`): one encode -> decode round trip per input flavour,
plus the failure exits, must run clean.  CPU only (GPU sanitizers are not available on this pool)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "oracle", "_san")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="1")


@pytest.fixture(scope="module")
def san_bins():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "san"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-300:])
    return os.path.join(SAN, "encode"), os.path.join(SAN, "decode")


def _clean(r):
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-2000:]


@pytest.mark.parametrize("bits,channels", [(16, 1), (8, 2)])
def test_round_trip_under_sanitizers(san_bins, tmp_path, bits, channels):
    enc, dec = san_bins
    pay = np.random.default_rng(bits + channels).integers(0, 256, 5380, dtype=np.uint8)
    (tmp_path / "p.bin").write_bytes(pay.tobytes())
    r = subprocess.run([enc, str(tmp_path / "x.wav"), "8000", str(bits), str(channels), "2000", "6", "ANONYMOUS", str(tmp_path / "p.bin")],
                       capture_output=True, text=True, env=ENV, timeout=300)
    assert r.returncode == 0, r.stderr
    _clean(r)
    r = subprocess.run([dec, str(tmp_path / "o.bin"), str(tmp_path / "x.wav")], capture_output=True, text=True, env=ENV, timeout=300)
    assert r.returncode == 0, r.stderr
    _clean(r)
    assert (tmp_path / "o.bin").read_bytes() == pay.tobytes()


def test_failure_exits_under_sanitizers(san_bins, tmp_path):
    """a stream that ends before the preamble (decode.cc:393) and a truncated frame: no payload, no sanitizer report"""
    enc, dec = san_bins
    pay = np.zeros(5380, np.uint8)
    (tmp_path / "p.bin").write_bytes(pay.tobytes())
    subprocess.run([enc, str(tmp_path / "x.wav"), "8000", "16", "1", "2000", "6", "ANONYMOUS", str(tmp_path / "p.bin")],
                   check=True, capture_output=True, env=ENV, timeout=300)
    wav = (tmp_path / "x.wav").read_bytes()
    for keep in (44 + 2 * 4000, 44 + 2 * 60000):
        cut = bytearray(wav[:keep])
        cut[4:8] = (len(cut) - 8).to_bytes(4, "little")
        cut[40:44] = (len(cut) - 44).to_bytes(4, "little")
        (tmp_path / "c.wav").write_bytes(bytes(cut))
        r = subprocess.run([dec, str(tmp_path / "o.bin"), str(tmp_path / "c.wav")], capture_output=True, text=True, env=ENV, timeout=300)
        # (like the reference's main, decode.cc:606-619, the CLI still writes OUTPUT and returns 0)
        assert r.returncode == 0 and "bit flips" not in r.stderr
        _clean(r)
