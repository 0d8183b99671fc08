#!/bin/bash
# ts_insts.sh -- instruction counts of the Theil-Sen kernel per stage: probe builds (tools/ts_probe.cpp, 51 200 rows of 432 points) under rocprofv3 --pmc
R=$PWD; O=$R/gpurun_out/ts_insts.txt; mkdir -p $R/gpurun_out; : > $O
cd $R/tools
for v in "full:" "skip_yint:-DTS_PROBE_SKIP_YINT" "skip_list:-DTS_PROBE_SKIP_LIST" "skip_main:-DTS_PROBE_SKIP_MAIN -DTS_PROBE_NO_FALLBACK" "skip_all:-DTS_PROBE_SKIP_MAIN -DTS_PROBE_SKIP_LIST -DTS_PROBE_SKIP_YINT"; do
	name=${v%%:*}; flags=${v#*:}
	hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -DVARIANT="\"$name\"" $flags ts_probe.cpp -o /tmp/tsp_$name || continue
	for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_BRANCH SQ_INSTS_LDS"; do
		d=/tmp/tsi_${name}_$(echo $c | tr ' ' '_'); rm -rf $d
		( cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $c -d $d -o x -- /tmp/tsp_$name > /dev/null 2>&1 )
		db=$(find $d -name "*.db" | head -1)
		python3 $R/tools/pmc_kernel.py $db k_theil_sen_raw 2>&1 | sed "s/^/$name /" >> $O
	done
done
cat $O
