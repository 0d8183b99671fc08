#!/bin/bash
# sc_quick.sh -- k_sc against the oracle (path, metric, min_fork bit-exact), then the legs that lean on it: -20 dB and configs[3]
O=$PWD/gpurun_out/${OUT:-sc_quick.txt}; : > $O
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sc_path_kernel or sc_certificate_at_scale or list1_pass" 2>&1 | tail -5 >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline())
print("value", round(d["value"]), "kernel_only", round(d["value_kernel_only"]))
for k in ("value_config3", "value_noise_m20", "value_mono"):
    l = d.get(k)
    if isinstance(l, dict):
        t = l.get("value_two_lanes") or {}
        print(k, round(l["value"]), "two lanes", round(t.get("value", 0)), "routes", l.get("routes"), "stage ms", {a: round(b, 3) for a, b in l.get("stage_ms_per_step", {}).items()})'
timeout 900 python3 bench.py --steps 4 --warmup 1 --cpu-frames 0 --host-frames 0 --scl-steps 0 2>&1 | tail -1 > gpurun_out/sc_quick_bench.json
python3 -c "$pick" < gpurun_out/sc_quick_bench.json >> $O 2>&1
cat $O
