#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters: python tools/pmcagg.py DIR [DIR...]"""
import csv, glob, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"]))
            a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
names = sorted({c for k in acc for c in acc[k]})
print("kernel," + ",".join(names))
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", [0, 1])[0]):
    print(k[:40] + "," + ",".join(f"{acc[k][c][0] / max(acc[k][c][1], 1):.4g}" if c in acc[k] else "" for c in names))
