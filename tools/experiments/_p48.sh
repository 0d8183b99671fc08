R=$PWD; cd /tmp; export TMPDIR=/tmp; export OFDMRX_NO_OVERLAP=1
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
 d=/tmp/p48_$(echo $c | tr ' ' '_')
 rocprofv3 --pmc $c -d $d -o x -- python3 $R/bench.py --rate 48000 --frames 4096 --steps 1 --warmup 0 --cpu-frames 0 > /dev/null 2>&1
 python3 $R/tools/pmc_kernel.py $(find $d -name "*.db" | head -1) k_demod
 python3 $R/tools/pmc_kernel.py $(find $d -name "*.db" | head -1) k_sync
done
