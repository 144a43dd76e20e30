#!/bin/bash
# registers, scratch (spills) and LDS of the kernels of one translation unit:  tools/kregs.sh k_sgs [pattern] [extra hipcc flags]
# (compiles the unit to gfx950 assembly and reads the .amdhsa_* lines)
UNIT=${1:-k_sgs}; PAT=${2:-.}; shift 2 2>/dev/null
cd "$(dirname "$0")/../cales_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S "$@" $UNIT.hip -o /tmp/kregs_$UNIT.s 2>/dev/null || exit 1
python3 - "$PAT" /tmp/kregs_$UNIT.s <<'PY'
import re, subprocess, sys
pat, path = sys.argv[1], sys.argv[2]
name = None; d = {}
for line in open(path):
    t = line.split()
    if not t: continue
    if t[0] == ".amdhsa_kernel": name = t[1]; d = {}
    elif t[0].startswith(".amdhsa_") and len(t) > 1: d[t[0][8:]] = t[1]
    elif t[0] == ".end_amdhsa_kernel" and name:
        nm = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        nm = re.sub(r"\(.*", "", nm)
        if re.search(pat, nm):
            v = int(d.get("next_free_vgpr", 0)); a = int(d.get("accum_offset", v))
            print(f"{nm}: vgpr {a} agpr {v - a} lds {d.get('group_segment_fixed_size')} scratch {d.get('private_segment_fixed_size')}")
PY
