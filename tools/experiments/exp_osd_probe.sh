#!/bin/bash
# exp_osd_probe.sh -- stage split of the OSD (k_osd_only, tools/osd_probe.cpp): early exits after each phase, clean saturated input
O=$PWD/gpurun_out/osd_probe.txt; mkdir -p gpurun_out; : > $O
cd tools
for v in 0 1 2 3 4; do
	F=""; [ $v != 0 ] && F="-DOSD_PROBE_STOP=$v"
	hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -DVARIANT="\"stop$v\"" $F osd_probe.cpp ../modem_amd/csrc/tables.cpp -o /tmp/osdp_$v 2>>$O && PROBE_N=8192 PROBE_DATA=n /tmp/osdp_$v | tail -1 >> $O
done
cd ..
cat $O
