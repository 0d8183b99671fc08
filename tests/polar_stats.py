#!/usr/bin/env python3
"""debug: how often the rate-1 sub-tree shortcut of k_polar applies (library built with -DPOLAR_STATS)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as O
import modem_amd
rx = modem_amd.Receiver(device=0, chunk_frames=8)
for db in (-30, -20, -16):
    p = O.payload_for(5)
    pcm = O.impair(O.encode_pcm(p, channels=2), noise_db=db, seed=3, frame=1)
    out, res = rx.decode(pcm[None])
    llr = rx.tap("LLR", 0)
    lanes, metric = rx.polar(llr)
    print(db, int(res[0]["status"]), "rate-1 blocks", metric[0][6], "shortcut ok", metric[0][7])
