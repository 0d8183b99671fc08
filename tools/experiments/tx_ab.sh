#!/bin/bash
# tx_ab.sh -- the transmitter alone (tools/tx_probe.py, ms per 8192 mode-6 frames) for the current build and VARIANTS
O=$PWD/gpurun_out/${OUT:-tx_ab.txt}; mkdir -p gpurun_out; : > $O
for v in "" $VARIANTS; do
	echo "== ${v:-current}" >> $O
	( [ -n "$v" ] && export MODEM_AMD_LIB=$PWD/modem_amd/lib/variants/libofdmrx_$v.so; python3 tools/tx_probe.py 8192 2>&1 | grep tx_encode >> $O )
done
[ -n "$TESTS" ] && timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "transmitter or encode_cli" 2>&1 | tail -3 >> $O
cat $O
