#!/usr/bin/env python3
"""Condenses rocprofv3 output (gpurun_out/...) into the per-round summaries kept under profiles/.

  python profiles/summarize.py ROUND STATS_DIR [FETCH_DIR WRITE_DIR] [NCELL]

Per kernel: calls, average duration, share; with the two --pmc passes also HBM traffic per launch
(FETCH_SIZE and WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE counts 64 B per
128-B request of a wide coalesced stream, so the read side is doubled as MI355X_MICROARCH.md
prescribes) and, with NCELL, bytes per cell.
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def main():
    rnd, stats_dir = sys.argv[1], sys.argv[2]
    fetch_dir = sys.argv[3] if len(sys.argv) > 4 else None
    write_dir = sys.argv[4] if len(sys.argv) > 4 else None
    ncell = float(sys.argv[5]) if len(sys.argv) > 5 else None
    rows = []
    for f in glob.glob(f"{stats_dir}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append(dict(kernel=short(r["Name"]), calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                             total_ms=float(r["TotalDurationNs"]) / 1e6, pct=float(r["Percentage"])))
    cnt = {}
    for key, d in (("fetch", fetch_dir), ("write", write_dir)):
        if not d:
            continue
        acc = defaultdict(lambda: [0.0, 0])
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                a = acc[short(r["Kernel_Name"])]
                a[0] += float(r["Counter_Value"]); a[1] += 1
        cnt[key] = {k: v[0] / v[1] * 1024.0 for k, v in acc.items()}      # KiB -> bytes per launch
    for r in rows:
        k = r["kernel"]
        if "fetch" in cnt and k in cnt["fetch"]:
            r["fetch_bytes_raw"] = cnt["fetch"][k]
            r["hbm_read_bytes"] = 2.0 * cnt["fetch"][k]                   # gfx950 correction
        if "write" in cnt and k in cnt["write"]:
            r["hbm_write_bytes"] = cnt["write"][k]
        if "hbm_read_bytes" in r and "hbm_write_bytes" in r:
            tot = r["hbm_read_bytes"] + r["hbm_write_bytes"]
            r["hbm_GBps"] = tot / (r["avg_us"] * 1e-6) / 1e9
            if ncell:
                r["words_per_cell"] = tot / 8.0 / ncell
    rows.sort(key=lambda r: -r["total_ms"])
    json.dump(rows, open(f"profiles/{rnd}_kernels.json", "w"), indent=1)
    # what bench.py reads for roofline.traffic: timer name -> kernel-name prefix, per-launch HBM bytes of each kernel
    b2k = {"mom_rk_fused": "k_momrk", "strain_filter_uvw": "k_strain_tile", "filter_s0sij": "k_filter6_tile",
           "lij_mij_contract": "k_lij_mij_tile", "correc": "k_correc", "fillps": "k_fillps", "updatep": "k_updatep",
           "gaussel_z": "k_gaussel", "fft_x_fwd": "k_fft_x8<0, 0, 0", "fillps_fft_x_fwd": "k_fft_x8<0, 0, 1", "correc_updatep": "k_correc_cell", "fft_x_bwd": "k_fft_x8<1", "fft_y_fwd": "k_fft_y8<0",
           "fft_y_bwd": "k_fft_y8<1"}
    if ncell:
        json.dump({"source": f"profiles/{rnd}_kernels.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 read x2)",
                   "ncell": ncell, "bench_to_kernel": b2k, "kernels": rows}, open("profiles/latest_kernels.json", "w"), indent=1)
    with open(f"profiles/{rnd}_kernels.md", "w") as fh:
        fh.write(f"# rocprofv3 summary {rnd} (see profiles/summarize.py)\n\n")
        fh.write("| kernel | calls | avg us | total ms | % | HBM read MB | HBM write MB | HBM GB/s | words/cell |\n|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            fh.write("| {kernel} | {calls} | {avg_us:.1f} | {total_ms:.2f} | {pct:.2f} | {rd} | {wr} | {bw} | {wc} |\n".format(
                **r, rd=f"{r['hbm_read_bytes'] / 1e6:.1f}" if "hbm_read_bytes" in r else "-",
                wr=f"{r['hbm_write_bytes'] / 1e6:.1f}" if "hbm_write_bytes" in r else "-",
                bw=f"{r['hbm_GBps']:.0f}" if "hbm_GBps" in r else "-",
                wc=f"{r['words_per_cell']:.2f}" if "words_per_cell" in r else "-"))
    print(open(f"profiles/{rnd}_kernels.md").read())


if __name__ == "__main__":
    main()
