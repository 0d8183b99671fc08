#!/bin/bash
# exp_ts2.sh -- what bounds the Theil-Sen classification pass: hit handling, scalar work per pair, or something else
O=$PWD/gpurun_out/ts2.txt; mkdir -p gpurun_out; : > $O
cd tools
build() { hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -DVARIANT="\"$1\"" $2 ts_probe.cpp -o /tmp/tsp_$1 && /tmp/tsp_$1 | tail -1 >> $O; }
build new_u8 ''
build nohit '-DTS_VARIANT_NOHIT -DTS_PROBE_NO_FALLBACK'
build perlane '-DTS_VARIANT_PERLANE -DTS_PROBE_NO_FALLBACK'
build perlane_u16 '-DTS_VARIANT_PERLANE -DTS_PROBE_NO_FALLBACK -DTS_U=16 -DTS_WAVES=4'
build skip_main '-DTS_PROBE_SKIP_MAIN -DTS_PROBE_NO_FALLBACK'
cd ..
cat $O
