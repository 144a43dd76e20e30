#!/usr/bin/env python3
"""Per-kernel averages (per dispatch) of rocprofv3 --pmc counters and of the dispatch duration:  python tools/pmcagg.py DIR [DIR...] [--match SUBSTR]"""
import csv, glob, re, sys
from collections import defaultdict
args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
if match in args: args.remove(match)
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
meta = {}
for d in args:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]); k = re.sub(r"\(.*", "", k)
            a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
            a = acc[k]["_us_" + r["Counter_Name"]]; a[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; a[1] += 1
            meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"])
for k in sorted(acc, key=lambda k: -max(v[0] for c, v in acc[k].items() if c.startswith("_us_"))):
    if match and match not in k: continue
    us = [v[0] / v[1] for c, v in acc[k].items() if c.startswith("_us_")]
    print(f"{k}\n   vgpr/agpr/sgpr/lds/wg/grid {meta[k]}  us/dispatch {min(us):.1f}..{max(us):.1f}")
    c = {n: v[0] / max(v[1], 1) for n, v in acc[k].items() if not n.startswith("_us_")}
    print("   " + "  ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())))
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        d = {n: c[n] / wc for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC") if n in c}
        print("   of wave cycles: " + "  ".join(f"{n[3:]}={v:.3f}" for n, v in d.items()))
    if "SQ_BUSY_CYCLES" in c and "SQ_ACTIVE_INST_VALU" in c:
        print(f"   VALU busy (ACTIVE_INST_VALU*4/ (BUSY_CYCLES/ (#SE..))) raw ratio ACTIVE_INST_VALU/BUSY_CYCLES = {c['SQ_ACTIVE_INST_VALU'] / c['SQ_BUSY_CYCLES']:.3f}")
    if "SQ_WAVES" in c:
        print("   per wave: " + "  ".join(f"{n[8:]}={c[n] / c['SQ_WAVES']:.1f}" for n in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if n in c))
    if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
        print(f"   LDS bank-conflict share {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.3f}")
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        print(f"   FETCH_SIZE {c.get('FETCH_SIZE', float('nan')):.4g} KB x2 (gfx950, 8-16 B/lane)  WRITE_SIZE {c.get('WRITE_SIZE', float('nan')):.4g} KB")
