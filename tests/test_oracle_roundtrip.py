"""Config 1 (README quick start): encode -> decode -> payload identity, CPU only.
This is the only results oracle the reference itself offers (README.md:4-40, decode.cc:533-541)."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O


@pytest.mark.parametrize("bits,channels,freq,mode", [
    (16, 1, 2000, 6), (16, 2, 2000, 6), (8, 1, 2000, 6),      # README / `make test` variants
    (16, 1, 1500, 7), (16, 2, 0, 8), (16, 1, 2000, 9), (16, 2, -500, 10), (16, 1, 1600, 12), (16, 2, 1000, 13),
])
def test_roundtrip(bits, channels, freq, mode):
    p = O.payload_for(mode * 10 + channels)
    pcm = O.encode_pcm(p, bits=bits, channels=channels, freq_off=freq, call_sign="AB1CD", mode=mode)
    out, res = O.decode(pcm)
    assert res.status == 0 and res.oper_mode == mode and res.bit_flips == 0
    assert (out == p).all()
    assert abs(res.cfo_rad * 8000 / (2 * np.pi) - freq) < 0.5
    if channels == 2 and mode == 6:
        assert res.sc_start == 8000 + 1440 + 160          # S&C body right after the leading pilot


def test_cli_roundtrip_and_skip(tmp_path):
    """argv contract of encode.cc:337-342 / decode.cc:559-563 incl. the SKIP argument"""
    O.build()
    a, b = tmp_path / "a.dat", tmp_path / "b.dat"
    a.write_bytes(bytes(O.payload_for(1))); b.write_bytes(bytes(O.payload_for(2)))
    wav, out = tmp_path / "e.wav", tmp_path / "d.dat"
    enc, dec = os.path.join(O.ORACLE_DIR, "encode"), os.path.join(O.ORACLE_DIR, "decode")
    subprocess.check_call([enc, str(wav), "8000", "16", "1", "2000", "6", "ANONYMOUS", str(a), str(b)])
    assert wav.stat().st_size == 44 + 2 * (16000 + (2 + 2 * 53) * 1440)
    for skip, want in ((0, a), (1, b)):
        r = subprocess.run([dec, str(out), str(wav), str(skip)], capture_output=True)
        assert r.returncode == 0 and out.read_bytes() == want.read_bytes()
    assert subprocess.run([dec], capture_output=True).returncode == 1


def test_noise_and_full_impairment_chain():
    """configs 3/4 operating points (README.md:49): AWGN -30 dB level; multipath+CFO+SFO+AWGN"""
    p = O.payload_for(77)
    pcm = O.encode_pcm(p, channels=2)
    x = O.impair(pcm, noise_db=-30, seed=3, frame=0)
    out, res = O.decode(x)
    assert res.status == 0 and (out == p).all() and 20 < res.esn0_db_last < 24
    x = O.impair(pcm, noise_db=-30, cfo_hz=234.567, sfo_ppm=147,
                 multipath=[(0, 1 + 0j), (7, 0.3 - 0.2j), (19, -0.1 + 0.15j)], seed=3, frame=1)
    out, res = O.decode(x)
    assert res.status == 0 and (out == p).all()
    assert abs(res.cfo_fine * 8000 / (2 * np.pi) - 2234.567) < 1.0


def test_failure_statuses():
    """every exit of Decoder::Decoder maps to a status (decode.cc:393,419,430,435,440,543)"""
    silence = np.zeros((30000, 1), np.int16)
    out, res = O.decode(silence)
    assert res.status == 1 and not out.any()                 # NO_SYNC; payload zeroed (documented F9 deviation)
    p = O.payload_for(5)
    pcm = O.encode_pcm(p, channels=2)
    x = O.impair(pcm, noise_db=-6, seed=1)                   # far too noisy for rate-2/3 8PSK
    out, res = O.decode(x)
    assert res.status != 0
    # destroy only the payload symbols: header survives, CRC-32 must fail
    y = pcm.copy()
    s = 8000 + 4 * 1440
    y[s + 10 * 1440: s + 40 * 1440] = 0
    out, res = O.decode(y)
    assert res.status == 6 and res.oper_mode == 6 and not out.any()


@pytest.mark.parametrize("rate,mode,channels,bits", [(16000, 6, 1, 16), (16000, 13, 2, 16), (44100, 6, 2, 16),
                                                      (44100, 9, 1, 8), (48000, 6, 1, 16), (48000, 10, 2, 16)])
def test_round_trip_other_sample_rates(rate, mode, channels, bits):
    """encode.cc:424-436 / decode.cc:590-602: the 16 / 44.1 / 48 kHz instantiations (symbol_len 2560 / 7056 / 7680,
    Hilbert 41 / 113 / 125 taps) round-trip like README.md:4-40"""
    p = O.payload_for(rate + mode)
    pcm = O.encode_pcm(p, bits=bits, channels=channels, freq_off=1500, mode=mode, rate=rate)
    sl = 1280 * rate // 8000
    assert (pcm.shape[0] - 2 * rate) % (sl + sl // 8) == 0
    out, res = O.decode(pcm, rate=rate)
    assert res.status == 0 and res.oper_mode == mode and (out == p).all() and res.bit_flips == 0
    # coarse CFO = the frequency offset in rad/sample at this rate (decode.cc:401)
    assert abs(res.cfo_rad * rate / (2 * np.pi) - 1500) < 1.0
