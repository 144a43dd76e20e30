#!/usr/bin/env python3
"""Per-kernel HBM traffic of the side configurations from the --pmc passes of tools/profile_configs.sh:  python tools/configs_md.py TAG > profiles/TAG_configs.md
(FETCH_SIZE x 2 + WRITE_SIZE, the factors of this round's calibration; counter traffic includes Infinity-Cache hits, MI355X_MICROARCH.md "HBM")"""
import csv, glob, json, re, sys
from collections import defaultdict
tag = sys.argv[1]
NC = {"c1": 64 ** 3, "c2": 256 * 128 * 128, "c4": 512 * 256 * 256, "c5": 1024 ** 3}
print(f"# Side configurations {tag}: kernel time (rocprofv3 --kernel-trace --stats) and HBM counters (separate --pmc FETCH_SIZE / WRITE_SIZE passes)\n")
print("Bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; read factor measured by this round's calibration on 8 B/lane streams). The counters sit on the L2 side of the")
print("Infinity Cache: traffic that the 256 MB cache serves is counted (the 256 x 128 x 128 case lives in it almost entirely).\n")
for cfg in ("c2", "c4", "c5"):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for kind in ("fetch", "write"):
        for f in glob.glob(f"gpurun_out/{tag}_{cfg}_{kind}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"]))
                a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    stats = {}
    for f in glob.glob(f"gpurun_out/{tag}_{cfg}_stats/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            stats[re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Name"]))] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
    try:
        b = json.load(open(f"gpurun_out/{tag}_{cfg}.json"))["configs"][cfg]; ms = b["ms_per_step"]
    except Exception:
        ms = float("nan")
    print(f"## {cfg}: {ms:.3f} ms/step under the profiler's kernel trace\n")
    print("| kernel | calls | avg us | % of kernel time | read MB | write MB | GB/s | words/cell |")
    print("|---|---|---|---|---|---|---|---|")
    for k, (calls, us, pct) in sorted(stats.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:14]:
        c = acc.get(k, {})
        rd = 2. * c["FETCH_SIZE"][0] / max(c["FETCH_SIZE"][1], 1) * 1024 if "FETCH_SIZE" in c else float("nan")
        wr = c["WRITE_SIZE"][0] / max(c["WRITE_SIZE"][1], 1) * 1024 if "WRITE_SIZE" in c else float("nan")
        print(f"| `{k}` | {calls} | {us:.1f} | {pct:.1f} | {rd / 1e6:.1f} | {wr / 1e6:.1f} | {(rd + wr) / us / 1e3:.0f} | {(rd + wr) / 8. / NC[cfg]:.2f} |")
    print()
