#!/bin/bash
# kres.sh file.hip kernel-substring [extra hipcc flags] -- registers / spills / LDS / instruction count of one kernel (cross-compiled, no GPU)
F=$(realpath $1); K=$2; shift 2
D=$(mktemp -d); cd $D
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I/root/repo/modem_amd/csrc "$@" -c $F -o x.o -save-temps=obj 2>&1 | grep -E "error" -A4
S=$(ls $D/*gfx950.s 2>/dev/null | head -1)
[ -z "$S" ] && { echo "no asm"; exit 1; }
python3 - "$S" "$K" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
    blk = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if pat not in name:
        continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    body = re.search(r"^%s:.*?s_endpgm" % re.escape(name), txt, re.S | re.M)
    n = len(re.findall(r"^\s+(?:v_|s_|ds_|buffer_|global_|scratch_)", body.group(0), re.M)) if body else -1
    print("%s: vgpr %s agpr %s sgpr %s sgpr_spill %s vgpr_spill %s scratch %s lds %s insts %d" % (
        name[:60], g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"),
        g("private_segment_fixed_size"), g("group_segment_fixed_size"), n))
PY
rm -rf $D
