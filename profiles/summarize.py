#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile_round.sh (gpurun_out/TAG_*) into the per-round summaries kept here.

  python profiles/summarize.py TAG [NCELL] [--label NAME]

Reads   gpurun_out/TAG_stats   (--kernel-trace --stats)            -> calls, average duration, share
        gpurun_out/TAG_fetch / TAG_write  (--pmc FETCH_SIZE / WRITE_SIZE, separate passes) -> HBM bytes per launch
        gpurun_out/TAG_sq1 / TAG_sq2      (two SQ passes)          -> issue / wait / LDS cycles and instruction counts
        gpurun_out/TAG_calib_fetch / _write + TAG_calib.jsonl      -> counter / known-bytes factors on 8 B and 16 B per lane streams
Writes  profiles/TAG_kernels.{md,json}, profiles/TAG_kernel_stats.csv, profiles/TAG_sq.md, profiles/TAG_calibration.json and
        profiles/latest_kernels.json (what bench.py reads for roofline.traffic).

FETCH_SIZE / WRITE_SIZE are reported in KiB. On gfx950 FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md, "HBM");
the factor applied to it is the one MEASURED by the calibration pass of the same round on an 8 B/lane FP64 stream (the access
width of this library's kernels), not the guide's figure for 16 B/lane.
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

GO = "gpurun_out"


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def counters(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: {c: v[0] / v[1] for c, v in d_.items()} for k, d_ in acc.items()}


def calibration(tag):
    """counter bytes / known bytes for the streams of tools/micro/calib.hip"""
    path = f"{GO}/{tag}_calib.jsonl"
    if not os.path.exists(path):
        return None
    known = {}
    for line in open(path):
        try:
            d = json.loads(line); known[d["kernel"]] = d
        except ValueError:
            pass
    fe, wr = counters(f"{GO}/{tag}_calib_fetch"), counters(f"{GO}/{tag}_calib_write")
    out = {}
    for k, d in known.items():
        kf = [x for x in fe if x.startswith(k)]; kw = [x for x in wr if x.startswith(k)]
        row = dict(read_bytes=d["read_bytes"], write_bytes=d["write_bytes"], ms=d["ms"], GBps=d["GBps"])
        if kf and d["read_bytes"]:
            row["FETCH_SIZE_over_bytes"] = fe[kf[0]]["FETCH_SIZE"] * 1024.0 / d["read_bytes"]
        if kw and d["write_bytes"]:
            row["WRITE_SIZE_over_bytes"] = wr[kw[0]]["WRITE_SIZE"] * 1024.0 / d["write_bytes"]
        out[k] = row
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tag = args[0]
    ncell = float(args[1]) if len(args) > 1 else None
    if ncell is None:      # the grid of the profiled bench run, from its own JSON line ("... channel 512x512x512, ...")
        try:
            line = open(f"{GO}/{tag}_bench_under_rocprof.json").read().strip().splitlines()[-1]
            m = re.search(r"(\d+)x(\d+)x(\d+)", json.loads(line)["config"]["workload"])
            ncell = float(int(m.group(1)) * int(m.group(2)) * int(m.group(3)))
        except Exception:
            ncell = None
    rows = []
    for f in glob.glob(f"{GO}/{tag}_stats/**/*kernel_stats.csv", recursive=True):
        shutil.copy(f, f"profiles/{tag}_kernel_stats.csv")
        for r in csv.DictReader(open(f)):
            rows.append(dict(kernel=short(r["Name"]), calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                             total_ms=float(r["TotalDurationNs"]) / 1e6, pct=float(r["Percentage"])))
    if os.path.exists(f"{GO}/{tag}_bench_under_rocprof.json"):
        shutil.copy(f"{GO}/{tag}_bench_under_rocprof.json", f"profiles/{tag}_bench_under_rocprof.json")
    cal = calibration(tag)
    rd_factor, wr_factor, cal_note = 2.0, 1.0, "guide figure (no calibration pass in this round)"
    if cal and "calib_copy8_rows" in cal and "FETCH_SIZE_over_bytes" in cal["calib_copy8_rows"]:
        rd_factor = 1.0 / cal["calib_copy8_rows"]["FETCH_SIZE_over_bytes"]
        wr_factor = 1.0 / cal["calib_copy8_rows"]["WRITE_SIZE_over_bytes"]
        cal_note = "measured in this round on an 8 B/lane FP64 copy of rows laid out like the library's fields (tools/micro/calib.hip)"
        json.dump({"note": "counter bytes / known bytes per stream; factors applied = 1 / (value of calib_copy8_rows)", "streams": cal,
                   "read_factor": rd_factor, "write_factor": wr_factor}, open(f"profiles/{tag}_calibration.json", "w"), indent=1)
    fe, wr = counters(f"{GO}/{tag}_fetch"), counters(f"{GO}/{tag}_write")
    sq = counters(f"{GO}/{tag}_sq1")
    for k, d in counters(f"{GO}/{tag}_sq2").items():
        sq.setdefault(k, {}).update({c: v for c, v in d.items() if c not in sq.get(k, {})})
    for r in rows:
        k = r["kernel"]
        if k in fe:
            r["fetch_bytes_raw"] = fe[k]["FETCH_SIZE"] * 1024.0
            r["hbm_read_bytes"] = rd_factor * r["fetch_bytes_raw"]
        if k in wr:
            r["hbm_write_bytes"] = wr_factor * wr[k]["WRITE_SIZE"] * 1024.0
        if "hbm_read_bytes" in r and "hbm_write_bytes" in r:
            tot = r["hbm_read_bytes"] + r["hbm_write_bytes"]
            r["hbm_GBps"] = tot / (r["avg_us"] * 1e-6) / 1e9
            if ncell:
                r["words_per_cell"] = tot / 8.0 / ncell
        if k in sq:
            r["sq"] = sq[k]
    rows.sort(key=lambda r: -r["total_ms"])
    json.dump(rows, open(f"profiles/{tag}_kernels.json", "w"), indent=1)
    b2k = {"mom_rk_fused": "k_momrk", "strain_filter_uvw": "k_strain_tile", "correc_strain_filter_uvw": "k_corr_strain_tile",
           "lij_mij_filter_contract": "k_lmf_tile", "correc": "k_correc", "fillps": "k_fillps", "updatep": "k_updatep",
           "gaussel_z": "k_gaussel", "fft_x_fwd": "k_fft_x8<0, 0, 0", "fillps_fft_x_fwd": "k_fft_x8<0, 0, 1", "correc_updatep": "k_correc_cell", "fft_x_bwd": "k_fft_x8<1", "fft_y_fwd": "k_fft_y8r<0",
           "fft_y_bwd": "k_fft_y8r<1"}
    if ncell:
        json.dump({"source": f"profiles/{tag}_kernels.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; read factor {rd_factor:.4f}, "
                             f"write factor {wr_factor:.4f}: {cal_note})",
                   "ncell": ncell, "bench_to_kernel": b2k, "kernels": rows}, open("profiles/latest_kernels.json", "w"), indent=1)
    with open(f"profiles/{tag}_kernels.md", "w") as fh:
        fh.write(f"# rocprofv3 summary {tag} (profiles/summarize.py; tools/profile_round.sh)\n\n")
        fh.write(f"HBM bytes = FETCH_SIZE x {rd_factor:.4f} + WRITE_SIZE x {wr_factor:.4f} ({cal_note}).\n\n")
        fh.write("| kernel | calls | avg us | total ms | % | HBM read MB | HBM write MB | HBM GB/s | words/cell |\n|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            fh.write("| {kernel} | {calls} | {avg_us:.1f} | {total_ms:.2f} | {pct:.2f} | {rd} | {wr} | {bw} | {wc} |\n".format(
                **r, rd=f"{r['hbm_read_bytes'] / 1e6:.1f}" if "hbm_read_bytes" in r else "-",
                wr=f"{r['hbm_write_bytes'] / 1e6:.1f}" if "hbm_write_bytes" in r else "-",
                bw=f"{r['hbm_GBps']:.0f}" if "hbm_GBps" in r else "-",
                wc=f"{r['words_per_cell']:.2f}" if "words_per_cell" in r else "-"))
    # SQ summary: fractions of the kernel's duration in which the four SIMDs of a CU issue VALU / the CU's LDS is busy
    with open(f"profiles/{tag}_sq.md", "w") as fh:
        fh.write(f"# SQ counters {tag} (two --pmc passes of 8 counters, averages per launch; tools/profile_round.sh)\n\n"
                 "SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md, cycle constants): "
                 "`valu/wave` = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES etc. are shares of a wave's life time. `VALU busy` = 4 x SQ_ACTIVE_INST_VALU / "
                 "(1024 SIMDs x duration x 2.4 GHz) and `LDS busy` = SQ_LDS_IDX_ACTIVE / (256 CUs x duration x 2.4 GHz) are upper-clock estimates of "
                 "pipe occupancy. `inst/wave` = instructions per wave (VALU, LDS, VMEM read+write, SALU).\n\n"
                 "| kernel | avg us | waves | wait_any/wave | wait_inst/wave | active/wave | valu/wave | lds_wait_inst/wave | VALU busy | LDS busy | bank conflict | VALU inst/wave | LDS inst/wave | VMEM inst/wave | SALU inst/wave |\n"
                 "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            s = r.get("sq")
            if not s or r["avg_us"] < 50:
                continue
            wc = max(s.get("SQ_WAVE_CYCLES", 0.), 1.); dur_cyc = r["avg_us"] * 1e-6 * 2.4e9; wv = max(s.get("SQ_WAVES", 0.), 1.)
            g = lambda c: s.get(c, float("nan"))
            fh.write(f"| {r['kernel']} | {r['avg_us']:.0f} | {g('SQ_WAVES'):.0f} | {g('SQ_WAIT_ANY') / wc:.2f} | {g('SQ_WAIT_INST_ANY') / wc:.2f} | {g('SQ_ACTIVE_INST_ANY') / wc:.2f} | "
                     f"{g('SQ_ACTIVE_INST_VALU') / wc:.2f} | {g('SQ_WAIT_INST_LDS') / wc:.2f} | {4 * g('SQ_ACTIVE_INST_VALU') / (1024 * dur_cyc):.2f} | {g('SQ_LDS_IDX_ACTIVE') / (256 * dur_cyc):.2f} | "
                     f"{g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):.3f} | {g('SQ_INSTS_VALU') / wv:.0f} | {g('SQ_INSTS_LDS') / wv:.0f} | {(g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR')) / wv:.0f} | {g('SQ_INSTS_SALU') / wv:.0f} |\n")
    print(open(f"profiles/{tag}_kernels.md").read())
    print(open(f"profiles/{tag}_sq.md").read())


if __name__ == "__main__":
    main()
