"""k_sc alone on n random LLR vectors (ofdmrx_debug_sc_path): its duration under rocprofv3 --kernel-trace --stats.  The decoder has no
data-dependent control flow, so the time of a launch does not depend on the values - which also makes a build whose arithmetic
is wrong on purpose (e.g. half the LDS array, to see what more resident decoders would buy) a valid TIMING probe.
  rocprofv3 --kernel-trace --stats -d gpurun_out/p -- python3 tools/experiments/sc_time_probe.py [n]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import modem_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rx = modem_amd.Receiver(device=0, chunk_frames=n)
rng = np.random.default_rng(1)
llr = rng.standard_normal((n, 65536), dtype=np.float32)
for rep in range(3):
    t0 = time.time()
    cw, hd, M, F, ok = rx.sc_path(llr)
    print("sc_path %d codewords: %.2f s wall (with the host copies), rule holds for %d" % (n, time.time() - t0, int(np.sum(ok))))
