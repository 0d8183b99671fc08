#!/bin/bash
# sc_time.sh "LIB:ENV[,ENV] ..." -- k_sc's launch duration on 8192 random codewords per library / environment (rocprofv3 kernel trace)
# The timing probe "sclds8" of profiles/r05_v7_sc_resident_decoders_timing_probe.txt (half the LDS array, wrong arithmetic on purpose) was:
#   sed -e 's/__shared__ float lds\[C \* 64 \* J\];/__shared__ float lds[C * 32 * J];/' -e 's/my\[(x + 32) \* J\]/my[((x + 1) \& 31) * J]/g' \
#       -e 's/\t\tlds\[lidx\] = t\[0\];/\t\tlds[lidx \& (32 * ScCfg<LB>::J - 1)] = t[0];/' modem_amd/csrc/k_sc.hip > /tmp/k_sc_lds8_probe.hip
#   SRC_k_sc=<that file, under the repo> tools/build_variant.sh sclds8 ""
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "$@"; do
	lib=${spec%%:*}; envs=${spec#*:}
	[ "$lib" != product ] && export MODEM_AMD_LIB=$R/modem_amd/lib/variants/libofdmrx_$lib.so || unset MODEM_AMD_LIB
	for e in ${envs//,/ }; do [ -n "$e" ] && export "$e"; done
	rm -rf /tmp/sct; timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/sct -o t -- python3 $R/tools/experiments/sc_time_probe.py 8192 > /tmp/sct.log 2>&1
	echo "== $spec: $(python3 -c "
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for name, calls, avg in c.execute('select name,total_calls,average from top_kernels'):
    if 'k_sc<' in name: print(name.split('(')[0], calls, 'launches, avg %.1f us' % (avg,))
" $(find /tmp/sct -name '*.db' | head -1) 2>&1 < /dev/null | head -2 | tr '\n' ' ')"
	for e in ${envs//,/ }; do [ -n "$e" ] && unset "${e%%=*}"; done
done
