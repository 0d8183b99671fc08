#!/bin/bash
# kernel-level profile of the Monte-Carlo sweep (configs[4]): where the time of transmitter + channel + decode goes
O=$PWD/gpurun_out/ber_prof.txt; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $GRAFT_REPO_ROOT/tools/ber_sweep.py --frames 65536 --levels -30 -28 -26 2>&1 | tail -4 >> $O
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/berprof -o ber -- python3 $GRAFT_REPO_ROOT/tools/ber_sweep.py --frames 65536 --levels -30 -26 > /tmp/rp.log 2>&1; true
python3 $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/berprof -name "*.db" | head -1) >> $O 2>&1
cat $O
