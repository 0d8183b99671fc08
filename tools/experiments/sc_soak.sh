#!/bin/bash
# sc_soak.sh -- GPU == oracle on fresh frames at levels where the list-1 pass decides (clean-node skips, lower-bound reruns, half arrays): 8192 frames per level,
# everything decided identical, every flip-count difference explained (tests/test_gpu_sweeps.py: _sweep)
cd tests && PYTHONPATH=..:. python3 - <<'PY' 2>&1 | tee ../gpurun_out/sc_soak.txt
import test_gpu_sweeps as T
for db, seed in ((-19.2, 9101), (-21.0, 9102), (-23.0, 9103), (-25.0, 9104), (-18.6, 9105)):
    s = T._sweep(8192, db, seed, allow_row_ties=True)
    print(db, "routes (certified, list-1, list)", s["routes"], "decoded", s["ok"], "frames that differ", s["differ"], s["classes"], "flip counts that differ (all explained)", s["flips_differ"], flush=True)
    assert s["ok"] == 8192 and all(c for c in s["classes"]) and len(s["differ"]) <= 2
print("soak ok")
PY
