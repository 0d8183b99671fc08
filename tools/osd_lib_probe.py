import numpy as np, modem_amd
rx = modem_amd.Receiver(chunk_frames=8192)
soft = np.random.default_rng(1).integers(-100, 100, (8192, 255)).astype(np.int8)
for i in range(2):
    rx.osd(soft)
