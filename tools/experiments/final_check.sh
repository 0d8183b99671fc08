#!/bin/bash
# final_check.sh -- what the driver runs at round end: GPU suite, smoke(), the default bench line
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -2
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python3 bench.py > gpurun_out/final_bench_n1.json 2>gpurun_out/final_bench_n1.err; cut -c1-300 gpurun_out/final_bench_n1.json
python3 -c "
import json; d=json.loads(open('gpurun_out/final_bench_n1.json').readline()); print(json.dumps(d['roofline'])); print(json.dumps(d['cpu_baseline']))"
