#!/bin/bash
O=$PWD/gpurun_out/misc.txt; mkdir -p gpurun_out; : > $O
make -C modem_amd/csrc -q all && echo "library up to date with sources" >> $O || echo "STALE LIBRARY" >> $O
timeout 1800 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6 >> $O
V=$PWD/modem_amd/lib/variants
for lib in default tx512 tx1024; do
	L=$V/libofdmrx_$lib.so; [ $lib = default ] && L=$PWD/modem_amd/lib/libofdmrx.so
	echo "== $lib: sweep driver 3 levels x 65536" >> $O
	MODEM_AMD_LIB=$L timeout 600 python3 tools/ber_sweep.py --frames 65536 --levels -40 -30 -20 2>&1 | tail -4 | cut -c1-200 >> $O
done
cat $O
