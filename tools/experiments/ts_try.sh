O=$PWD/gpurun_out/ts_try.txt; : > $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "theil or other_rates or all_modes" 2>&1 | tail -3 >> $O
pick='import json,sys
d=json.loads(sys.stdin.readline()); s=d["stage_ms_per_launch_alone"]; print("value", round(d["value"]), "kernel_only", round(d["value_kernel_only"]), "alone:", {k: round(v,3) for k,v in s.items()})'
for i in 1 2; do timeout 300 python3 bench.py --steps 6 --warmup 2 --cpu-frames 0 --host-frames 0 --scl-steps 0 --leg-steps 0 2>&1 | tail -1 | python3 -c "$pick" >> $O 2>&1; done
cat $O
