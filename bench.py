#!/usr/bin/env python3
"""bench.py -- time-steps/s of the CaLES hot path on MI355X, with roofline and CPU-baseline objects.

Workload (BASELINE.json, configs[2], the one the metric is quoted on): turbulent channel, 512^3,
dynamic Smagorinsky, from the reference's examples/les/_manuscript_turbulent_channel/input.nml
(ng -> 512^3, sgstype -> 'dsmag', visci = 10000), Poiseuille + vortex-pair initial field (inivel='poi',
is_wallturb=T; deterministic), bulk-velocity forcing in x. A step = 3 RK substeps (src/main.f90:417-508)
with the fields resident in HBM. dt is fixed to 0.5*dt_cfl(initial field).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--ng n1 n2 n3] [--sgs none|smag|dsmag] [--configs all|none|c1,c2,c4,c5] [--no-cpu]

After the headline measurement (N = 1) the other four BASELINE.json configurations are timed for a few steps each on the same GPU and reported
under "configs" (C1 64^3 Taylor-Green, C2 256x128x128 wall-modelled channel, C4 512x256x256 z-implicit duct, C5 1024^3 cavity); the headline
fields are the 512^3 channel's alone.

N > 1: one process per GPU (torch.distributed.run), y-slab decomposition of the SAME 512^3 problem
(strong scaling), see cales_amd/decomp.py.
"""
import os as _os
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC between the ranks' processes (RCCL); must be set before the HIP runtime starts
# The CPU baseline runs in a CHILD process (cpu_baseline below) that alone gets OMP_PROC_BIND / OMP_PLACES: set process-wide they would make the
# OpenMP run-time bind the launching thread of EVERY rank of a --gpus N run to the first core of the common affinity mask, and the HIP / RCCL helper
# threads created from it would inherit that one-core mask.
_HOST_THREADS = len(_os.sched_getaffinity(0)) if hasattr(_os, "sched_getaffinity") else (_os.cpu_count() or 1)
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
HBM_COPY = 6.29e12         # B/s, the copy rate the same guide measures (79 % of the peak); this repo's own copies of 2 GiB reach 5.4-5.5 TB/s,
                           # its read-only streams 6.3-6.4 (tools/micro/calib.hip, profiles/r03*_calibration.json)

# ALGORITHMIC FP64 words per cell per launch = the words of the reference loop nests a kernel replaces (SURVEY.md 8a and
# App. E). Fused kernels are credited with the sum of the loops they fuse, so `achieved` is an
# effective bandwidth; the hardware traffic of the same kernel (rocprofv3 PMC) is reported in `traffic`.
# The roofline object prices a kernel at its COMPULSORY traffic: every input field read once, every output field written once
# (stencil halos, re-reads and scratch are overhead and show up in `traffic`).
WORDS = {"mom_xyz_ad": 7, "rk_update": 13, "fillps": 4, "correc": 7, "updatep": 3, "bulk_forcing": 2,
         "fft_x_fwd": 2, "fft_y_fwd": 2, "gaussel_z": 2, "fft_y_bwd": 2, "fft_x_bwd": 2,
         "fillps_fft_x_fwd": 4,     # fillps folded into the forward x transform: u,v,w in, spectrum out
         "correc_updatep": 9,       # pp,u,v,w,p in; u,v,w,p out
         "mom_rk_fused": 12,        # u,v,w,visct,p + 3 old r.h.s. in; u,v,w + 3 r.h.s. out = 14, less the old r.h.s. of substep 1 (weight 0)
                                    # and the r.h.s. of substep 3 (never read): 11, 14, 11 -> 12 on average over a step
         "strain_filter_uvw": 13,   # u,v,w in; |S|, 6 |S|Sij, 3 test-filtered velocities out (+ 3 cell-centred ones when the last pass does not form them itself)
         "correc_strain_filter_uvw": 19,   # the same pass with the projection and the pressure update folded in (cales_step, one rank): u*,v*,w*,pp,p in; u,v,w,p + |S|, 6 |S|Sij, 3 filtered velocities out
         "lij_mij_filter_contract": 12,   # the same pass with the test filter of |S|Sij formed on the fly: 3 (u,v,w, or the stored cell-centred velocity) + 3 + 6 (raw |S|Sij) in; partial sums out
         "strain_rate": 10, "filter3d": 2}
# momentum pass without subgrid model (no eddy viscosity read): u,v,w,p + 3 old r.h.s. in, u,v,w + 3 r.h.s. out = 13; 10, 13, 10 over the substeps.
# With the projection of substeps 1 and 2 folded into the next momentum pass (the default, cales_step): substep 1 as before (10), substeps 2 and 3 also read
# pp and write p: 8 in + 7 out = 15 and 8 + 4 = 12 -- 37/3 on average, and two of the three correction passes (9 words each) are gone.
NOSGS_MOM_WORDS = 11 if "CALES_UNFOLDED_MOM" in os.environ else 37. / 3.
W_STEP = {"none": 44, "smag": 51, "dsmag": 178}   # words/cell/substep of the REFERENCE's loop nests; x3 per step


def channel_case(ng, sgs):
    from cales_amd.nml import parse_text
    text = open(os.path.join(ROOT, "cales_amd", "cases", "turbulent_channel.nml")).read()
    case = parse_text(text)
    case.ng[:] = ng
    case.sgstype = sgs
    return case


# The other BASELINE.json configurations, timed after the headline measurement on the same GPU. `words` = compulsory words per cell
# and launch of the kernels these cases run that differ from the table above; `w_ref` = words per cell and substep of the reference's loop nests.
CONFIGS = {
    "c1": {"file": "taylor_green_64.nml", "impdiff": 0, "warmup": 20, "steps": 200, "w_ref": 44, "baseline_config": 0,
           "what": "triply periodic Taylor-Green vortex 64^3, explicit diffusion, no subgrid model",
           "words": {"mom_rk_fused": NOSGS_MOM_WORDS}},
    "c2": {"file": "channel_wall_model_256.nml", "impdiff": 0, "warmup": 10, "steps": 60, "w_ref": 51, "baseline_config": 1,
           "what": "turbulent channel 256x128x128, static Smagorinsky + log-law wall model on both z walls, bulk forcing in x",
           "words": {"cmpt_sgs_smag": 4}},     # u,v,w in, visct out
    "c4": {"file": "duct_wall_model_512.nml", "impdiff": 2, "warmup": 3, "steps": 20, "w_ref": 72, "baseline_config": 3,
           "what": "square duct 512x256x256, static Smagorinsky + wall model on four walls, z-implicit (Crank-Nicolson) diffusion, NN pressure in y and z",
           "words": {"cmpt_sgs_smag": 4, "helmholtz_z": 3,      # per component: u, implicit r.h.s. in; u out
                     "mom_rk_fused": 15}},    # + 3 implicit r.h.s. out: 14, 17, 14 over the substeps
    "c5": {"file": "lid_driven_cavity_1024.nml", "impdiff": 0, "warmup": 1, "steps": 3, "w_ref": 44, "baseline_config": 4,
           "what": "lid-driven cavity 1024^3, no subgrid model, all-Neumann pressure (DCT-II/III in x and y), fields of 8.6 GB",
           "words": {"mom_rk_fused": NOSGS_MOM_WORDS}},
}
SOLVE = ["fft_x_fwd", "fft_y_fwd", "gaussel_z", "fft_y_bwd", "fft_x_bwd"]


def load_case(fname, impdiff=0):
    from cales_amd.nml import parse_text
    case = parse_text(open(os.path.join(ROOT, "cales_amd", "cases", fname)).read())
    case.impdiff = impdiff
    return case


def solve_figures(stats, nloc, RB):
    """fillps + Poisson solve from the per-kernel timers: time of one solve and its compulsory words (10: five passes x read + write; 12 when fillps
    is folded into the forward x pass: u,v,w in, spectrum out)."""
    ms = sum(stats[k][1] / stats[k][0] for k in SOLVE if k in stats and stats[k][0])
    words, note = 10, "x fwd, y fwd, z tridiagonal, y bwd, x bwd: 5 passes x (read + write)"
    if stats.get("fillps_fft_x_fwd", (0, 0))[0]:
        ms += stats["fillps_fft_x_fwd"][1] / stats["fillps_fft_x_fwd"][0]; words = 12
        note = "fillps + solve: fillps folded into the x-forward pass (u,v,w in, spectrum out) + y fwd, z tridiagonal, y bwd, x bwd"
    return ms, words, note


def run_config(key, HotPath, initflow, SMALL, RB):
    """A few steps of one of the other BASELINE.json configurations on the current device: ms/step without per-kernel events, then the same number
    of steps with events for the dominant kernel's roofline fraction and the solve's."""
    cfg = CONFIGS[key]
    case = load_case(cfg["file"], cfg["impdiff"])
    t_setup = time.perf_counter()
    h = HotPath(case)
    try:
        if case.inivel == "zer":      # one zero field uploaded four times (1024^3: 8.6 GB of host memory instead of 34)
            z = np.zeros(tuple(int(x) + 2 for x in case.ng), order="F", dtype=np.float32 if RB == 4.0 else np.float64)
            h.upload(z, z, z, z); del z
        else:
            u, v, w, p = initflow(case); h.upload(u, v, w, p); del u, v, w, p
        h.startup()
        dt = 0.5 * h.chkdt()
        icheck = int(case.icheck) if int(case.icheck) > 0 else 10
        nblocks = [0]

        dts = [dt]

        def run(first, count):
            for istep in range(first + 1, first + count + 1):
                h.step(dts[-1])
                if istep % icheck == 0:
                    dtmax = h.chkdt(); divtot, divmax = h.chkdiv(); nblocks[0] += 1
                    if not np.isfinite(divtot) or divmax > SMALL:       # main.f90:538
                        raise RuntimeError(f"{key} invalid at step {istep}: dtmax {dtmax}, divergence {divmax}")
                    # the flow evolves (the Taylor-Green vortex breaks down within the timed steps): where the fixed dt would trip the reference's
                    # abort rule dt > cfl dtmax (main.f90:530) the step is taken down to 0.5 dtmax again -- the reference itself sets dt = cfl dtmax
                    # at every icheck (main.f90:526-527). The work per step does not depend on dt.
                    if dts[-1] > dtmax * case.cfl:
                        dts.append(0.5 * dtmax)
        W, K = cfg["warmup"], cfg["steps"]
        run(0, W); h.sync()
        t_setup = time.perf_counter() - t_setup
        nblocks[0] = 0
        t0 = time.perf_counter(); run(W, K); h.sync(); t = time.perf_counter() - t0
        timed_blocks = nblocks[0]
        h.profile_reset(); h.profile(True)
        run(W + K, K); h.sync()
        h.profile(False)
        stats = h.profile_stats()
        divtot, divmax = h.chkdiv()
        if not np.isfinite(divtot) or divmax > SMALL:
            raise RuntimeError(f"{key} invalid: divergence {divmax}")
        plan = h.describe_plan()
    finally:
        h.close()
    ncell = float(np.prod(case.ng))
    words = dict(WORDS); words.update(cfg["words"])
    leaf = {k: v for k, v in stats.items() if k in words and v[0] > 0}
    dom = max(leaf, key=lambda k: leaf[k][1])
    calls, ms = leaf[dom]
    ach = words[dom] * RB * ncell / (ms / calls * 1e-3)
    sms, sw, snote = solve_figures(stats, ncell, RB)
    ms_step = 1e3 * t / K
    return {"baseline_config": cfg["baseline_config"], "workload": cfg["what"], "grid": "x".join(str(int(x)) for x in case.ng), "case_file": "cales_amd/cases/" + cfg["file"],
            "impdiff": cfg["impdiff"], "path": plan, "dt": dt, "dt_lowered_times": len(dts) - 1, "steps": K, "warmup": W, "ms_per_step": ms_step, "time_steps_per_s": 1e3 / ms_step, "setup_s": t_setup,
            "icheck_blocks_in_timed_region": timed_blocks,
            "dominant_kernel": dom,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
                         "words_per_cell": words[dom], "avg_launch_ms": ms / calls, "launches": calls},
            "poisson_solve": {"ms": sms, "words_per_cell": sw, "frac_of_hbm_peak": sw * RB * ncell / (sms * 1e-3) / HBM_PEAK if sms else None, "passes": snote},
            # the reference's loop nests would move 3 x 8 B x N x w_ref per step: the step's time at 100 % of the peak, and the measured step against it
            "ideal_ms_reference_traffic_at_peak": 1e3 * 3 * RB * ncell * cfg["w_ref"] / HBM_PEAK,
            "step_vs_reference_traffic_frac_of_peak": 3 * RB * ncell * cfg["w_ref"] / (t / K) / HBM_PEAK,
            "timer_scopes_per_step": sum(v[0] for v in stats.values()) / K,
            "kernels_ms_per_step": {k: round(v[1] / K, 4) for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1])[:12]},
            "divmax": divmax}


def _transpose_report(out, stats, case, world, a, solve, h):
    if world <= 1 or not stats.get("alltoall", (0, 0))[0]:
        return
    # the y <-> x-mode re-slab of the Poisson solve (replaces the reference's 2decomp/cuDecomp pencil transposes, solver.f90:50-66):
    # per solve and direction every rank sends (P-1) blocks of n3 * (n2/P) * ceil((n1/2+1)/P) complex modes, in one call or -- with the
    # second stream (cales_set_comm_overlap) -- in k-chunks that travel beside the x/y transforms. Durations from HIP events on the
    # stream the exchange is queued on, rank 0 (with the library's own RCCL calls these bracket the ncclSend/ncclRecv group itself).
    cw = -(-(int(case.ng[0]) // 2 + 1) // world)
    try:      # (the library pads the columns per rank to whole 128-B lines where that costs 6 % or less: its own figure)
        cw = int(h.describe_plan().get("mode_columns_per_rank", cw))
    except Exception:
        pass
    from cales_amd import capi
    blk = int(case.ng[2]) * (int(case.ng[1]) // world) * cw * (8.0 if capi.SINGLE else 16.0)      # complex modes
    calls, ms = stats["alltoall"]
    solves = 6.0 * a.steps                                  # 3 substeps x 2 directions
    chunks = calls / solves
    solve1 = {k: stats[k][1] / stats[k][0] * (stats[k][0] / (3.0 * a.steps)) for k in solve + ["fillps_fft_x_fwd"] if stats.get(k, (0, 0))[0]}
    halo = stats.get("halo_exchange", (0, 0.0))
    out["transpose"] = {"alltoall_calls_per_step": calls / a.steps, "chunks_per_exchange": chunks, "alltoall_ms_per_exchange": ms / solves,
                        "alltoall_ms_per_step": ms / a.steps, "bytes_out_per_rank_per_exchange": (world - 1) * blk,
                        "GBps_out_per_rank": (world - 1) * blk / (ms / solves * 1e-3) / 1e9,
                        "GBps_per_link": blk / (ms / solves * 1e-3) / 1e9,
                        "solve_kernels_ms_per_solve_rank0": sum(solve1.values()), "solve_alltoall_ms_per_solve": 2 * ms / solves,
                        "overlapped": bool(chunks > 1.5),
                        "halo_exchange_calls_per_step": halo[0] / a.steps, "halo_exchange_ms_per_step": halo[1] / a.steps,
                        "note": "pencil-transpose scaling efficiency = t_solve(1 GPU) / (P * t_solve(P)), t_solve = solve kernels + the part of the 2 exchanges "
                                "that is not hidden (with chunks: exchange time beyond the transforms it runs beside); the driver computes it from the per-N lines. "
                                "No N > 1 run on hardware exists yet (no multi-GPU box was available to the build)"}


def cpu_baseline(a):
    """The CPU baseline in a child process of its own: only that process sees OMP_PROC_BIND / OMP_PLACES (threads pinned to cores), it never touches
    the GPU, and the memory of the 512^3 oracle run (~35 GB) is returned to the host when it exits."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("OMP_PROC_BIND", "spread"); env.setdefault("OMP_PLACES", "cores")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--ng"] + [str(x) for x in a.ng] + ["--sgs", a.sgs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode or not lines:
        return {"error": f"cpu baseline child failed (exit code {r.returncode})", "value": None, "unit": "time-steps/s", "cores": 0, "kind": "port", "sample": "none"}
    return json.loads(lines[-1])


def cpu_quota_cores():
    """CPU time the container may use, in cores (cgroup v2 cpu.max, v1 cfs quota), or None when unlimited / unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def _cpu_baseline_impl(case_full, seconds_budget=40.0, full_budget=150.0, sample_dims=(256, 256, 128)):
    """Oracle (oracle/cales_oracle.c, OpenMP) timed on the host's cores, twice:
      1. a bounded sample, 256x256x128 (1/16 of the cells), on the team size that runs it fastest among {32, 64, 128, all host threads}
         (threads pinned: OMP_PROC_BIND=spread, OMP_PLACES=cores in this child process's environment; fields first-touched by the threads that use them);
      2. ONE real step of the full workload (512^3: ~35 GB of host memory, ~30 s) with that team, when the host has the memory and the first
         measurement says it fits `full_budget` seconds (a warm-up step comes first when there is time for two).
    `value` is the full-size measurement when it exists, otherwise the sample's rate scaled by cell count (labelled `scaled`)."""
    from oracle.oracle import Oracle
    avail = _HOST_THREADS

    def prepare(case, nthreads):
        o = Oracle(case, nthreads=nthreads, team_sums=True)
        u, v, w, p = o.initflow(case.inivel, case.is_wallturb)          # (every field first touched by the team: Oracle.zeros)
        visct, pp = o.zeros(), o.zeros()
        o.bounduvw(u, v, w, True, False); o.boundp(p, 0); o.cmpt_sgs(u, v, w, visct); o.boundp(visct, 1)
        dt = 0.5 * o.chkdt(visct, u, v, w)
        return o, dt, (u, v, w, p, pp, visct)

    case = case_full.copy()
    case.ng[:] = tuple(min(int(a), b) for a, b in zip(case_full.ng, sample_dims))
    tried = {}
    t_begin = time.perf_counter()
    best = None
    # what the container may use is not what the host shows: the GPU boxes of this pool run with a CPU quota (cgroup cpu.max = 1600000 100000, i.e.
    # 16 cores' worth of time, on a host that lists 256 hardware threads) -- a team larger than the quota is throttled and runs SLOWER (0.49 / 0.80 /
    # 1.15 / 2.50 s per 256x256x128 step with 16 / 32 / 64 / 128 threads, tools/oracle_prof.py), which is the whole of the "port does not scale
    # beyond 32 threads" of rounds 2 and 3. The teams tried are the quota and its neighbours; without a quota 32, 64, 128 and all hardware threads.
    quota = cpu_quota_cores()
    sizes = (32, 64, 128, avail) if quota is None else (max(1, int(quota) // 2), int(quota + 0.5), 2 * int(quota + 0.5))
    for cores in sorted({max(1, min(avail, c)) for c in sizes}):
        # a trial = one warm-up step (twiddles, scratch first touch) + one timed step; stop trying larger teams when the budget is half used
        if tried and time.perf_counter() - t_begin > 0.5 * seconds_budget:
            break
        o, dt, st = prepare(case, cores)
        o.step(dt, *st)
        t0 = time.perf_counter(); o.step(dt, *st); tried[cores] = time.perf_counter() - t0
        if best is None or tried[cores] < tried[best[0]]:
            best = (cores, o, dt, st)
        elif tried[cores] > 1.5 * tried[best[0]]:
            break                                         # larger teams only get slower (oversubscribed / bandwidth-bound host)
    cores, o, dt, st = best
    t0 = time.perf_counter(); k = 0
    while k < 1 or (time.perf_counter() - t_begin < seconds_budget and k < 10):
        o.step(dt, *st); k += 1
    t = (time.perf_counter() - t0) / k
    o.close(); del o, st, best
    ncell_s = float(np.prod(case.ng)); ncell_f = float(np.prod(case_full.ng))
    # ---- one real step at full size
    full = None
    est = t * ncell_f / ncell_s
    try:
        free_gb = int(open("/proc/meminfo").read().split("MemAvailable:")[1].split()[0]) / 2 ** 20
    except (OSError, IndexError, ValueError):
        free_gb = 0.
    need_gb = 42. * 8. * float(np.prod([int(x) + 2 for x in case_full.ng])) / 2 ** 30      # fields of the oracle's dynamic model + the caller's six
    if ncell_f > ncell_s and free_gb > 1.3 * need_gb and 1.6 * est < full_budget:
        tf0 = time.perf_counter()
        of, dtf, stf = prepare(case_full, cores)
        t_setup = time.perf_counter() - tf0
        warm = None
        if t_setup + 2.4 * est < full_budget:
            t0 = time.perf_counter(); of.step(dtf, *stf); warm = time.perf_counter() - t0
        t0 = time.perf_counter(); of.step(dtf, *stf); tfull = time.perf_counter() - t0
        full = {"grid": "x".join(str(int(x)) for x in case_full.ng), "s_per_step": tfull, "steps": 1, "warmup_step_s": warm, "setup_s": t_setup,
                "threads": cores, "time_steps_per_s": 1.0 / tfull}
        of.close(); del of, stf
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    dims = "x".join(str(int(x)) for x in case.ng)
    scaled_value = (1.0 / t) * ncell_s / ncell_f
    sample = (f"{k} steps of the same case at {dims} ({t:.3f} s/step, OpenMP over {cores} of {avail} host threads"
              + (f" -- the container's CPU quota is {quota:g} cores, larger teams are throttled" if quota else "") + f", fastest of {sorted(tried)}; "
              f"scaled by cell count that is {scaled_value:.4f} steps/s at full size)")
    if full:
        sample = (f"ONE real step of the workload itself ({full['grid']}: {full['s_per_step']:.1f} s on {cores} threads"
                  + (f", after a warm-up step of {full['warmup_step_s']:.1f} s" if full["warmup_step_s"] else ", no warm-up step") + "); besides it " + sample)
    else:
        sample += f"; no full-size step (host memory available {free_gb:.0f} GB, needed ~{1.3 * need_gb:.0f} GB; estimated {est:.0f} s per step)"
    return {"value": full["time_steps_per_s"] if full else scaled_value, "unit": "time-steps/s", "cores": cores, "kind": "port",
            "full_size": full,
            "measured": {"grid": dims, "s_per_step": t, "steps": k, "time_steps_per_s": 1.0 / t, "scaled_to_full_size": scaled_value,
                         "s_per_step_by_threads": {str(c): round(v, 4) for c, v in sorted(tried.items())}},
            "scaled": full is None and ncell_s != ncell_f, "cpu_model": model, "host_threads_available": avail, "cpu_quota_cores": quota,
            "omp": {"OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_PLACES": os.environ.get("OMP_PLACES")},
            "sample": sample + ". The oracle is a C port of the path (OpenMP over planes), not the reference's 2decomp/FFTW build (FFTW is not in the image); "
                               "the reference itself parallelises with one MPI rank per core"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ng", type=int, nargs=3, default=[512, 512, 512])
    ap.add_argument("--sgs", default="dsmag", choices=["none", "smag", "dsmag"])
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--configs", default="all", help="N = 1: the other BASELINE.json configurations timed after the headline measurement: all | none | c1,c2,c4,c5")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--skip-headline", action="store_true",
                    help="profiling aid (tools/profile_configs.sh): only the configurations named by --configs, printed as {\"configs\": ...}; NOT a bench line")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: several ranks on ONE GPU with host-staged messages (tests of the launch path on a one-GPU box; not a measurement)")
    ap.add_argument("--overlap", action="store_true",
                    help="N > 1: exchanges on the library's second stream beside kernels (CALES_OVERLAP=1; default: in order on one stream, the form "
                         "with the fewest assumptions, until a node with real peers has confirmed the overlapped one)")
    ap.add_argument("--one-order", action="store_true",
                    help="N > 1: time only one exchange order (in order, or with --overlap the second-stream one) instead of both in one invocation")
    ap.add_argument("--timeout", type=int, default=int(os.environ.get("CALES_BENCH_TIMEOUT_S", "1200")),
                    help="N > 1: seconds after which a rank that is still running (rendezvous that never completes, an exchange whose peer never "
                         "arrives) prints what it was doing and exits with status 124, which makes the launcher end the other ranks and exit non-zero")
    a = ap.parse_args()
    if a.cpu_baseline_child:      # (cpu_baseline above; nothing here touches the GPU)
        print(json.dumps(_cpu_baseline_impl(channel_case(a.ng, a.sgs))))
        return
    if a.overlap:
        os.environ["CALES_OVERLAP"] = "1"
    want = [] if a.configs == "none" else sorted(CONFIGS) if a.configs == "all" else [k.strip().lower() for k in a.configs.split(",") if k.strip()]
    for k in want:
        if k not in CONFIGS:
            raise SystemExit(f"--configs: unknown configuration '{k}' (c1, c2, c4, c5)")

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: start the N ranks ourselves, as CHILD processes of a parent that never touches the GPU (nothing has
        # imported torch or called HIP yet), the way the driver does (torch.distributed.run, one rank per GPU); rank 0's JSON line is relayed
        # and the children's exit code is ours
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        # hang guard: the ranks arm their own watchdog (rank_watchdog below: a rank that cannot rendezvous or sits in an exchange whose peer never
        # arrives exits non-zero after --timeout seconds, torch.distributed.run then ends the others); this parent is the second line -- its children
        # live in a process group of their own and are KILLED (never re-executed: they have touched the GPU) when they outlive the limit + 60 s
        import signal
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, start_new_session=True)
        try:
            stdout, _ = proc.communicate(timeout=a.timeout + 60)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            stdout, _ = proc.communicate()
            sys.stderr.write(f"bench.py --gpus {a.gpus}: the ranks did not finish within {a.timeout + 60} s and were killed\n")
            sys.stdout.write(stdout or ""); sys.stdout.flush()
            raise SystemExit(124)
        lines = [l for l in (stdout or "").splitlines() if l.startswith("{")]
        sys.stdout.write((stdout or "") if not lines else lines[-1] + "\n"); sys.stdout.flush()
        raise SystemExit(proc.returncode if proc.returncode or lines else 1)

    if a.skip_headline:
        from cales_amd import capi
        from cales_amd.hotpath import SMALL, HotPath, initflow
        print(json.dumps({"configs": {k: run_config(k, HotPath, initflow, SMALL, 4.0 if capi.SINGLE else 8.0) for k in want}}))
        return
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # ---- hang guard of a rank (N > 1): a daemon thread that ends THIS process with status 124 when the run has not finished after --timeout seconds --
    # os._exit works while the main thread sits in hipStreamSynchronize or in a collective (ctypes and torch release the GIL there); the launcher
    # (torch.distributed.run, or the parent above) then ends the other ranks and exits non-zero. `stage` says where the rank was.
    stage = ["start"]
    if world > 1:
        import threading

        # `soft`: a second deadline around the SECOND exchange order of one invocation -- when it passes, rank 0 prints the line it already holds (the in-order
        # measurement, with the reason under "overlap") and every rank leaves with status 0: a second-stream path that hangs on real peers must not cost the
        # first measurement of this path on hardware. All ranks set it behind one barrier, so they act within a tick of each other.
        wd = {"hard": time.time() + a.timeout, "soft": None, "line": None}

        def rank_watchdog():
            while True:
                time.sleep(0.5)
                now = time.time()
                if wd["soft"] is not None and now > wd["soft"]:
                    sys.stderr.write(f"bench.py rank {rank}/{world}: the second-stream order is still in stage '{stage[0]}' past its deadline -- leaving with the in-order result\n"); sys.stderr.flush()
                    if rank == 0 and wd["line"]:
                        sys.stdout.write(wd["line"] + "\n"); sys.stdout.flush()
                    os._exit(0)
                if now > wd["hard"]:
                    sys.stderr.write(f"bench.py rank {rank}/{world}: still in stage '{stage[0]}' after {a.timeout} s -- giving up (exit 124)\n"); sys.stderr.flush()
                    os._exit(124)
        threading.Thread(target=rank_watchdog, daemon=True).start()
        if os.environ.get("CALES_BENCH_TEST_HANG_RANK") == str(rank):      # test hook of the hang guard (tests/test_gpu_decomp.py): this rank never reaches the rendezvous
            stage[0] = "test hook: hanging before the rendezvous"
            time.sleep(10 ** 6)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if a.backend == "gloo":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        import datetime
        stage[0] = "rendezvous (init_process_group)"
        pg_timeout = datetime.timedelta(seconds=max(60, a.timeout // 2))
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=pg_timeout)
        else:
            dist.init_process_group("gloo", timeout=pg_timeout)
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")

    from cales_amd import capi
    from cales_amd.hotpath import SMALL, initflow      # SMALL = epsilon(1._rp)*10**(precision(1._rp)/2) of the working precision (param.f90:24)
    capi_single = capi.SINGLE                      # CALES_PRECISION=single: the -D_SINGLE_PRECISION build (not the headline number: BASELINE.json quotes FP64)
    RB = 4.0 if capi_single else 8.0               # bytes per real
    case = channel_case(a.ng, a.sgs)
    icheck = int(case.icheck) if int(case.icheck) > 0 else 10

    def measure(label):
        """One context, W warm-up steps, K timed steps, K more with per-kernel events; the exchange order is whatever CALES_OVERLAP says when the context
        is created (Flags::read_env). Returns the open context (rank 0 reads its plan and exchange kind) and the measurements."""
        stage[0] = f"{label}: set-up"
        if world == 1:
            from cales_amd.hotpath import HotPath
            h = HotPath(case)
            u, v, w, p = initflow(case)
            h.upload(u, v, w, p)
            del u, v, w, p
        else:
            from cales_amd.decomp import SlabHotPath
            h = SlabHotPath(case, dist, torch)
            h.upload_initial()
        h.startup()
        dt = 0.5 * h.chkdt()
        # same-box calibration (read / write / copy streams over this context's own fields, cales_calibrate) BEFORE the warm-up steps, so that the
        # timed region starts from the steady state of consecutive steps
        calib = h.calibrate(3)

        def barrier():
            h.sync()                      # the context's own HIP stream
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        checks = []

        def run_steps(first, count):
            """`count` time steps of the reference's loop (main.f90:405-544): every `icheck` steps chkdt and chkdiv with their abort
            rules -- two reductions that wait for the device, as the reference's do (dt itself stays fixed)."""
            for istep in range(first + 1, first + count + 1):
                h.step(dt)
                if istep % icheck == 0:
                    dtmax = h.chkdt(); divtot, divmax = h.chkdiv()
                    checks.append((istep, dtmax, divtot, divmax))
                    if dt > dtmax * case.cfl or not np.isfinite(divtot) or divmax > SMALL:       # main.f90:530,538 (fixed dt = 0.5 dt_cfl); small of param.f90:24
                        raise SystemExit(f"bench invalid at step {istep}: dtmax {dtmax}, divergence {divmax}")

        stage[0] = f"{label}: warm-up steps"
        run_steps(0, a.warmup)
        barrier()
        # timed region: exactly K steps (icheck blocks included), no per-kernel events (two hipEventRecords around each of the ~200
        # launches of a step cost 1-3 % of the step at 512^3)
        stage[0] = f"{label}: timed steps"
        t0 = time.perf_counter()
        run_steps(a.warmup, a.steps)
        barrier()
        t = time.perf_counter() - t0
        # the same K steps again with HIP events on the context's stream around every kernel: durations for the roofline object
        stage[0] = f"{label}: steps with kernel events"
        h.profile_reset(); h.profile(True)
        t0 = time.perf_counter()
        run_steps(a.warmup + a.steps, a.steps)
        barrier()
        t_prof = time.perf_counter() - t0
        h.profile(False)
        if world > 1:
            tt = torch.tensor([t, t_prof], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t, t_prof = float(tt[0].item()), float(tt[1].item())
        stats = h.profile_stats()
        divtot, divmax = h.chkdiv()
        if not np.isfinite(divtot) or divmax > SMALL:       # the reference's abort rule, main.f90:538
            raise SystemExit(f"bench invalid: divergence {divmax}")
        stage[0] = f"{label}: done"
        return h, {"t": t, "t_prof": t_prof, "stats": stats, "checks": checks, "divmax": divmax, "dt": dt, "calibration": calib}

    overlap_env = os.environ.get("CALES_OVERLAP", "0") not in ("", "0") and "CALES_NO_OVERLAP" not in os.environ
    t_begin = time.perf_counter()
    h, m = measure("second-stream exchanges" if overlap_env and world > 1 else "in-order exchanges" if world > 1 else "single GPU")
    t, t_prof, stats, checks, divmax, dt = m["t"], m["t_prof"], m["stats"], m["checks"], m["divmax"], m["dt"]
    # N > 1: BOTH exchange orders in one invocation -- `value` is the in-order run above, the second-stream order (k-chunked
    # transposition beside the x / y transforms, scratch-field halos beside the interior tiles of the dynamic model's last pass) is timed on a fresh
    # context and reported under "overlap"
    both = world > 1 and not a.one_order and not overlap_env and (a.backend == "nccl" or "CALES_BENCH_TEST_BOTH" in os.environ)      # (test hook: the flow of the second measurement with gloo ranks on one GPU, whose staged exchanges have no second stream)
    m2 = None
    plan1 = h.describe_plan()
    native1 = getattr(h, "native", False)
    t_first = time.perf_counter() - t_begin
    out = None
    if rank == 0:
        ncell = float(np.prod(case.ng)); nloc = ncell / world
        ms_step = 1e3 * t / a.steps
        # dominant kernel (by measured time) among the per-kernel timers
        leaf = {k: v for k, v in stats.items() if k in WORDS and v[0] > 0}
        dom = max(leaf, key=lambda k: leaf[k][1])
        calls, ms = leaf[dom]
        ach = WORDS[dom] * RB * nloc / (ms / calls * 1e-3)
        # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc passes (profiles/summarize.py), if present
        traffic, traffic_src = None, None
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "latest_kernels.json")))
            kmap = prof.get("bench_to_kernel", {})
            for row in prof.get("kernels", []):
                if row["kernel"].startswith(kmap.get(dom, "\0")) and "hbm_read_bytes" in row and prof.get("ncell") == ncell and world == 1:
                    traffic = row["hbm_read_bytes"] + row["hbm_write_bytes"]; traffic_src = prof.get("source")
        except (OSError, ValueError, KeyError):
            pass
        # ... and the kernel FURTHEST below the HBM roof among those that weigh at least 3 % of the step (the dominant kernel by time is also the best
        # streaming pass of the step: its fraction alone hides where the headroom is)
        tot_ms = sum(v[1] for v in leaf.values())
        heavy = {k: v for k, v in leaf.items() if v[1] >= 0.03 * tot_ms}
        frac_of = lambda k: WORDS[k] * RB * nloc / (leaf[k][1] / leaf[k][0] * 1e-3) / HBM_PEAK
        worst = min(heavy, key=frac_of)
        # every heavy kernel within 0.02 of that minimum is listed with it (the z sweep at 0.477 hid the dynamic model's last pass at
        # 0.494 and 26 % of the step)
        near = sorted((k for k in heavy if k != worst and frac_of(k) <= frac_of(worst) + 0.02), key=frac_of)
        calib = m["calibration"]
        copy_rate = calib["copy_GBps"] * 1e9      # the read + write copy of field-shaped rows THIS box sustains (cales_calibrate, before the warm-up steps)
        wtraffic = None
        try:
            for row in prof.get("kernels", []):
                if row["kernel"].startswith(kmap.get(worst, "\0")) and "hbm_read_bytes" in row and prof.get("ncell") == ncell and world == 1:
                    wtraffic = row["hbm_read_bytes"] + row["hbm_write_bytes"]
        except (NameError, KeyError):
            pass
        solve = SOLVE
        # cales_step folds fillps into the forward x pass (u,v,w in instead of pp): the pair fillps + solve is priced at its
        # compulsory 4 + 4 x 2 words (the separate passes: 4 + 10)
        solve_ms, solve_words, solve_note = solve_figures(stats, nloc, RB)
        out = {
            "metric": "time-steps/sec", "value": a.steps / t, "unit": "time-steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_step, "ms_per_step_with_kernel_events": 1e3 * t_prof / a.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if capi_single else "f64", "data": "synthetic",
            "config": {"workload": f"turbulent channel {case.ng[0]}x{case.ng[1]}x{case.ng[2]}, sgstype={case.sgstype}, "
                                   "PP/PP/NN pressure BCs, bulk forcing in x (BASELINE.json configs[2]); 3 RK substeps/step",
                       "decomposition": f"y-slabs x{world}" if world > 1 else "single GPU", "dt": dt,
                       # the path cales_step took (struct StepPlan, cales_describe_plan): WHICH fused / folded form of every operator was timed
                       "path": plan1,
                       "exchange_order": (("second stream beside kernels (CALES_OVERLAP=1)" if overlap_env
                                           else "in order on the context's stream (default)") if world > 1 else None),
                       "exchanges": ("RCCL from the library" if native1 else
                                     "gloo with host staging, ranks sharing one GPU: a test of the launch path, NOT a measurement" if a.backend == "gloo" else
                                     "torch.distributed callbacks") if world > 1 else None},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": ach / HBM_PEAK, "frac_of_guide_copy_rate": ach / HBM_COPY, "frac_of_measured_copy_rate": ach / copy_rate,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": WORDS[dom] * RB * nloc, "avg_launch_ms": ms / calls, "launches": calls},
            "roofline_worst": {"bound": "hbm", "kernel": worst, "achieved": frac_of(worst) * HBM_PEAK / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                               "frac": frac_of(worst), "frac_of_measured_copy_rate": frac_of(worst) * HBM_PEAK / copy_rate,
                               "traffic": wtraffic, "algorithmic_bytes_per_launch": WORDS[worst] * RB * nloc,
                               "avg_launch_ms": leaf[worst][1] / leaf[worst][0], "launches": leaf[worst][0], "share_of_kernel_time": leaf[worst][1] / tot_ms,
                               "also_within_0.02": [{"kernel": k, "frac": frac_of(k), "frac_of_measured_copy_rate": frac_of(k) * HBM_PEAK / copy_rate,
                                                     "avg_launch_ms": leaf[k][1] / leaf[k][0], "launches": leaf[k][0], "share_of_kernel_time": leaf[k][1] / tot_ms,
                                                     "algorithmic_bytes_per_launch": WORDS[k] * RB * nloc} for k in near],
                               "what": "the kernel furthest below the HBM roof among those with >= 3 % of the step's kernel time, and every other such kernel within 0.02 of it"},
            # same-box calibration (cales_calibrate): streams over the rows of this context's own fields in the library's layout, 8 B per lane, before the warm-up steps
            "calibration": dict(calib, what="read-only / write-only / copy (bytes read + written) over field-shaped rows of the timed context, GB/s on THIS box; "
                                            "frac_of_measured_copy_rate = algorithmic bytes per second / copy_GBps"),
            "poisson_solve": {"ms": solve_ms, "words_per_cell": solve_words, "passes": solve_note,
                              "algorithmic_GBps": solve_words * RB * nloc / (solve_ms * 1e-3) / 1e9 if solve_ms else None,
                              "frac_of_hbm_peak": solve_words * RB * nloc / (solve_ms * 1e-3) / HBM_PEAK if solve_ms else None,
                              "frac_of_guide_copy_rate": solve_words * RB * nloc / (solve_ms * 1e-3) / HBM_COPY if solve_ms else None,
                              "frac_of_measured_copy_rate": solve_words * RB * nloc / (solve_ms * 1e-3) / copy_rate if solve_ms else None},
            # north_star: ">= 50 % of the HBM roofline on the Poisson + RK sweep": one (fillps +) solve + one fused
            # momentum/RK pass (its compulsory words, averaged over the three substeps) over the time of exactly those kernels
            "poisson_plus_rk": (lambda ms, w: {"ms": ms, "words_per_cell": w, "algorithmic_GBps": w * RB * nloc / (ms * 1e-3) / 1e9,
                                               "frac_of_hbm_peak": w * RB * nloc / (ms * 1e-3) / HBM_PEAK})(
                solve_ms + stats["mom_rk_fused"][1] / stats["mom_rk_fused"][0], solve_words + WORDS["mom_rk_fused"]) if solve_ms and stats.get("mom_rk_fused", (0, 0))[0] else None,
            # traffic the reference's kernel-per-loop sequence would move for the same step, over the measured time: >1 is
            # possible and only says that fusion removed traffic; it is NOT a roofline fraction
            "step_vs_reference_traffic": {"reference_GB_per_step": 3 * RB * ncell * W_STEP[case.sgstype] / 1e9,
                                          "equivalent_frac_of_hbm_peak": 3 * RB * ncell * W_STEP[case.sgstype] / (t / a.steps) / (world * HBM_PEAK)},
            "kernels_ms_per_step": {k: round(v[1] / a.steps, 4) for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1])},
            "divmax": divmax,
            "icheck": {"every": icheck, "blocks_in_timed_region": sum(1 for c in checks if a.warmup < c[0] <= a.warmup + a.steps),
                       "what": "chkdt + chkdiv (+ eddy viscosity materialised for chkdt) with the abort rules of main.f90:523-544"},
        }
        try:
            _transpose_report(out, stats, case, world, a, solve, h)
        except Exception as e:      # never lose the bench line over the extra report
            out["transpose"] = {"error": repr(e)}
    if both:
        # (all ranks) the second order on a fresh context; the in-order line is already complete on rank 0 and is what the run returns if this part fails
        h.close(); h = None
        if rank == 0:
            wd["line"] = json.dumps(dict(out, overlap={"error": "the second-stream order did not finish before its deadline: the in-order measurement is what this line holds"}))
        stage[0] = "barrier before the second-stream order"
        dist.barrier()
        wd["soft"] = time.time() + float(os.environ.get("CALES_BENCH_SOFT_S", max(180., 4. * t_first)))
        os.environ["CALES_OVERLAP"] = "1"
        try:
            if os.environ.get("CALES_BENCH_TEST_HANG_RANK2") == str(rank):      # test hook: this rank never joins the second measurement
                stage[0] = "test hook: hanging before the second-stream order"
                time.sleep(10 ** 6)
            h, m2 = measure("second-stream exchanges")
            m2["plan"] = h.describe_plan()
        except Exception as e:      # (a rank whose peers are still inside an exchange leaves through the soft deadline above)
            m2 = None
            if rank == 0:
                out["overlap"] = {"error": repr(e)}
        finally:
            os.environ["CALES_OVERLAP"] = "0"
            wd["soft"] = None
    if rank == 0:
        if m2 is not None:      # the second exchange order of the same invocation
            o2 = {"ms_per_step": 1e3 * m2["t"] / a.steps, "value": a.steps / m2["t"], "ms_per_step_with_kernel_events": 1e3 * m2["t_prof"] / a.steps,
                  "exchange_order": "second stream beside kernels (CALES_OVERLAP=1)", "path": m2["plan"], "divmax": m2["divmax"],
                  "speedup_over_in_order": m["t"] / m2["t"],
                  "kernels_ms_per_step": {k: round(v[1] / a.steps, 4) for k, v in sorted(m2["stats"].items(), key=lambda kv: -kv[1][1])}}
            try:
                _transpose_report(o2, m2["stats"], case, world, a, solve, h)
            except Exception as e:
                o2["transpose"] = {"error": repr(e)}
            out["overlap"] = o2
        elif world > 1 and "overlap" not in out:
            out["overlap"] = {"skipped": "--one-order" if a.one_order else "the timed run IS the second-stream order (--overlap / CALES_OVERLAP=1)" if overlap_env
                              else "gloo with host staging has no second-stream exchanges"}
        if h is not None:
            h.close(); h = None      # (frees the 45 GB of the 512^3 context before the 1024^3 case)
        if world == 1 and want:
            from cales_amd.hotpath import HotPath as _HP
            out["configs"] = {}
            for k in want:
                try:
                    out["configs"][k] = run_config(k, _HP, initflow, SMALL, RB)
                except Exception as e:      # never lose the headline line over a side measurement
                    out["configs"][k] = {"error": repr(e)}
        if world == 1 and not a.no_cpu:
            out["cpu_baseline"] = cpu_baseline(a)
        print(json.dumps(out))
    if h is not None:
        h.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
