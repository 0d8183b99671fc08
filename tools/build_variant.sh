#!/bin/bash
# build_variant.sh NAME "EXTRA HIPCC FLAGS" -- an alternative build of libofdmrx.so for A/B runs; SRC_<stem>=path replaces one source file
# (e.g. SRC_k_polar=tools/experiments/variants/k_polar_level9_in_registers.hip)
# (MODEM_AMD_LIB=modem_amd/lib/variants/libofdmrx_NAME.so).  PERFILE_<stem>="flags" in the environment adds flags to one file.  Objects go to /tmp; only the .so lands in-tree
# (git-ignored, travels with gpurun).  Files default to every source of the library.
set -e
NAME=$1; FLAGS=$2; shift 2 || true
R=$(cd "$(dirname "$0")/.." && pwd); S=$R/modem_amd/csrc; O=/tmp/variant_$NAME; mkdir -p $O $R/modem_amd/lib/variants
SRC="k_sync.hip k_header.hip k_demod.hip k_theilsen.hip k_polar.hip k_sc.hip k_finish.hip k_channel.hip k_tx.hip api_create.cpp api_pipeline.cpp api_debug.cpp api_tx.cpp tables.cpp"
pids=()
for f in $SRC; do
	( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Wno-unused-result $FLAGS $( [ $f = k_demod.hip ] && echo -fno-slp-vectorize ) $( [ $f = k_theilsen.hip -o $f = k_sc.hip ] && echo "-mllvm -disable-machine-licm" ) $(eval echo \$PERFILE_${f%.*}) -I$S -c $( o=$(eval echo \$SRC_${f%.*}); [ -n "$o" ] && echo $R/$o || echo $S/$f ) -o $O/${f%.*}.o ) &
	pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/modem_amd/lib/variants/libofdmrx_$NAME.so $O/*.o
echo "built modem_amd/lib/variants/libofdmrx_$NAME.so"
