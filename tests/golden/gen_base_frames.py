#!/usr/bin/env python3
"""Generate tests/golden/base_frames_2ch.npz: a few clean mode-6 frames exactly as the oracle's
restatement of `encode OUT 8000 16 2 2000 6 ANONYMOUS payload` writes them (2-channel analytic
int16, 95200 samples) together with their payloads.  bench.py tiles them and adds per-frame AWGN on
the device (config 3 of BASELINE.json); tests use them as known-answer inputs.  Data only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

N = 4
pays = np.stack([O.payload_for(9000 + i) for i in range(N)])
pcm = np.stack([O.encode_pcm(pays[i], bits=16, channels=2, freq_off=2000, call_sign="ANONYMOUS", mode=6) for i in range(N)])
np.savez_compressed(os.path.join(HERE, "base_frames_2ch.npz"), pcm=pcm, payload=pays)
print(pcm.shape, pcm.dtype, os.path.getsize(os.path.join(HERE, "base_frames_2ch.npz")))
