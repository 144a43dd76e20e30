"""Multi-rank (y-slab) path on ONE GPU: P ranks are emulated by P threads whose exchange callbacks copy
device-to-device (cales_amd.decomp.LoopbackComm). Exercises the pack/unpack kernels, the blocked
mode layout of the distributed Poisson solve, the slab-aware initial field and the all-reduces, and
compares with the single-rank run of the same case (reduction order differs -> 1e-10)."""
import numpy as np
import pytest

from tests.util import F, load_golden, relerr

pytestmark = pytest.mark.gpu


def _case(name, ng):
    if name == "cavity_smag":      # static Smagorinsky with walls in x, y and z: the kernel-per-loop sequence (k_smag) with the y walls' shear planes
        g, case = load_golden("cavity_dsmag")
        case.ng[:] = ng; case.sgstype = "smag"
        return case
    if name == "openy_imp3d":      # inflow / outflow along y (the decomposed direction), side walls in x, 3-D implicit diffusion
        from tests.test_gpu_vs_oracle import _open_case
        return _open_case(("DD", "DD"), ("DN", "NN"), ng)
    imp3d = name in ("cavity_imp3d", "devchan_imp3d")
    g, case = load_golden({"cavity_imp3d": "cavity_nnn", "devchan_imp3d": "devchan_nd"}.get(name, name))
    case.ng[:] = ng
    if imp3d:
        case.impdiff = 1
    if case.sgstype == "none" and case.cbcvel[0, 0, 0] != "P":      # see tests/test_gpu_golden.py
        case.cbcsgs[:, 0] = "D"
    return case


def _single(case, nsteps):
    from cales_amd.hotpath import HotPath, initflow
    h = HotPath(case)
    u, v, w, p = initflow(case)
    h.upload(u, v, w, p); h.startup()
    dt = 0.5 * h.chkdt()
    for _ in range(nsteps):
        h.step(dt)
    out = h.download() + [dt, h.chkdiv(), h.dpdl()]
    h.close()
    return out


SLAB_CASES = [("chan_smag_wm", (32, 24, 16), 2), ("chan_smag_wm", (32, 24, 16), 4),
                                       ("chan_dsmag", (24, 32, 16), 4), ("tgv_ppp", (16, 24, 12), 3),
                                       ("chan_dsmag_wm", (72, 32, 40), 2), ("tgv_dsmag_ppp", (32, 24, 16), 3),
                                       ("halfchan_imp1d", (16, 16, 12), 2), ("chan_smag", (64, 16, 8), 8),
                                       ("duct_smag_wm", (16, 24, 24), 2), ("duct_smag_wm_imp1d", (16, 24, 24), 2), ("cavity_nnn", (16, 24, 12), 4),
                                       # static Smagorinsky between y walls on more than two slabs (BASELINE configs[3] on 4 and 8 GPUs): van Driest with global rows and
                                       # the shear planes of both y walls summed over the slabs; row-marching kernel, k_smag
                                       ("duct_smag_wm", (16, 32, 24), 4), ("duct_smag_wm", (16, 64, 24), 8), ("duct_smag_wm_imp1d", (16, 32, 24), 4),
                                       ("duct_smag_wm_imp1d", (32, 64, 16), 8), ("duct_smag_wm", (16, 24, 24), 3), ("cavity_smag", (16, 24, 12), 4), ("cavity_smag", (16, 24, 12), 3),
                                       ("devchan_nd", (32, 24, 12), 3),
                                       # static Smagorinsky duct without wall model, power-of-two rows: x ghost columns left alone inside the step, the shear planes wrap around
                                       ("duct_smag", (64, 32, 16), 4), ("duct_smag", (64, 16, 16), 2),
                                       # dynamic model with walls in y/z and in x/y/z (kernel-per-loop sequence, slab halos of its scratch fields)
                                       ("duct_dsmag_wm", (16, 24, 20), 2), ("duct_dsmag", (16, 24, 12), 3), ("cavity_dsmag", (16, 24, 12), 2),
                                       ("couette_imp3d_ops", (32, 24, 16), 2), ("chan_dsmag", (128, 32, 136), 2),
                                       # power-of-two lines in x and y: the radix-8 transforms, hence the k-chunked exchange beside them (NCH = 4 / 2), with the
                                       # bulk means fused into the forward pass or not (wall model), Neumann-Neumann kinds, periodic z, a duct with the dynamic model
                                       ("chan_dsmag_wm", (64, 32, 32), 4), ("cavity_nnn", (64, 32, 32), 2), ("tgv_ppp", (32, 32, 16), 2), ("duct_dsmag_wm", (32, 32, 32), 2),
                                       ("chan_smag", (128, 64, 64), 8),
                                       # dynamic model with whole 64-cell tiles in x: the projection folded into the strain-rate pass on every slab (ghost rows from the
                                       # neighbours, pp of the upper neighbour's row 2 through the companion field); slabs shorter and longer than a tile, z periodic
                                       ("chan_dsmag", (64, 32, 24), 4), ("chan_dsmag", (64, 64, 16), 8), ("tgv_dsmag_ppp", (64, 24, 16), 3), ("chan_dsmag", (128, 44, 20), 2),
                                       # 3-D implicit diffusion with no-slip walls in x and y (wall-normal DST-I in the slab and in the mode-block layout)
                                       ("cavity_imp3d", (32, 24, 12), 2), ("cavity_imp3d", (20, 36, 10), 4),
                                       # ... and with open boundaries: inflow / outflow along x (RODFT01/10 in the slab) and along y (in the mode-block layout)
                                       ("devchan_imp3d", (32, 24, 12), 3), ("openy_imp3d", (16, 24, 12), 2), ("openy_imp3d", (20, 36, 10), 4),
                                       # 256-, 512- and 1024-point y lines in the mode-block layout: eight columns per block, sixteen elements per thread (k_fft_y16:
                                       # Neumann 256 / 512 / 1024, periodic 1024), ranks whose mode columns are not a multiple of eight; 1024 planes of real x modes on
                                       # slabs: the persistent z tile (k_gaussel_tile_p) on the blocks of a peer segment
                                       ("cavity_nnn", (32, 1024, 12), 4), ("cavity_nnn", (48, 512, 10), 2), ("cavity_nnn", (64, 256, 12), 8), ("chan_smag", (32, 1024, 8), 2),
                                       ("cavity_nnn", (32, 16, 1024), 2)]


@pytest.mark.parametrize("name,ng,P", SLAB_CASES)
def test_slab_ranks_match_single_rank(name, ng, P):
    case = _case(name, ng)
    if P == 8 and name.startswith("duct_smag"):
        case.hwm = 0.2      # the sampling height must lie inside the slab that owns the wall (sanity.f90:224-231): l(2)/8 = 0.25 is its upper bound here
    _slabs_against_single_rank(case, P, 2)


@pytest.mark.parametrize("key,P", [("c3", 8), ("c3", 2), ("c2", 4), ("c4", 8)])
def test_slab_ranks_match_single_rank_at_baseline_sizes(key, P):
    """The decomposition the north star asks for, at BASELINE.json's own sizes and by value: the 512^3 channel with the dynamic model on 8 and 2 slabs, the
    256 x 128 x 128 wall-modelled channel on 4, the 512 x 256 x 256 wall-modelled duct (z-implicit) on 8 -- emulated ranks on one GPU (every rank a context of
    its own, exchanges through device copies) against the one-rank run of the same case file bench.py times; two steps, every field of every slab."""
    import bench
    case = bench.channel_case((512, 512, 512), "dsmag") if key == "c3" else bench.load_case(bench.CONFIGS[key]["file"], bench.CONFIGS[key]["impdiff"])
    _slabs_against_single_rank(case, P, 2)


def _slabs_against_single_rank(case, P, nsteps):
    from cales_amd.decomp import run_loopback
    u, v, w, p, visct, dt, div, dpdl = _single(case, nsteps)

    def body(h, r):
        h.upload_initial(); h.startup()
        dtr = 0.5 * h.chkdt()
        assert abs(dtr / dt - 1) < 1e-12
        for _ in range(nsteps):
            h.step(dtr)
        return h.download() + [h.chkdiv(), h.dpdl(), h.lo, h.n]

    res = run_loopback(case, P, body)
    for r, (ur, vr, wr, pr, visr, divr, dpdlr, lo, n) in enumerate(res):
        j0 = lo[1] - 1
        sl = slice(j0 + 1, j0 + n[1] + 1)
        for a, b, nm in ((ur, u, "u"), (vr, v, "v"), (wr, w, "w"), (visr, visct, "visct")):
            assert relerr(a[:, 1:-1, :], b[:, sl, :]) < 1e-10, (r, nm)
        assert divr[1] < max(1e-11, 10. * div[1])      # (open boundaries: the level of the one-rank run)
        assert np.abs(dpdlr - dpdl).max() < 1e-9 * max(1., np.abs(dpdl).max())
    # pressure: compare after removing the global mean (singular mode)
    pg = np.concatenate([res[r][3][:, 1:-1, :] for r in range(P)], axis=1)
    pref = p[:, 1:-1, :]
    assert relerr(pg[1:-1, :, 1:-1] - pg[1:-1, :, 1:-1].mean(), pref[1:-1, :, 1:-1] - pref[1:-1, :, 1:-1].mean()) < 1e-9


@pytest.mark.parametrize("lazy", [False, True], ids=["eager", "lazy-ignored"])
@pytest.mark.parametrize("name,ng,P", [("chan_nosgs", (64, 24, 16), 2), ("chan_nosgs", (32, 24, 12), 3), ("cavity_nnn", (32, 16, 12), 2), ("cavity_nnn", (16, 24, 12), 3)])
def test_slab_ranks_projection_folded_into_the_momentum_pass(name, ng, P, lazy, monkeypatch):
    """No subgrid model on several slabs: the projection of substeps 1 and 2 is applied by the next momentum pass on every slab (k_momrk<.., CORR = 1>:
    the p + pp store, the P / scr1 swap, the pressure's ghost rows riding with the prediction's bounduvw). Compared with the ONE-rank run with the separate
    correction pass: velocity, p (mean removed) AND pp, x and z ghost cells included. The third substep's projection is never left pending on several
    slabs -- completing it would make every later entry a collective --, so CALES_LAZY_PROJECTION changes nothing there: one correction pass per step and
    rank either way, and a download that only rank 0 makes returns."""
    from cales_amd.decomp import run_loopback
    from cales_amd.hotpath import HotPath, initflow
    from tests.test_gpu_golden import _nosgs_case
    case = _nosgs_case(name, ng)
    nsteps = 2

    def perturbed():
        u, v, w, p = initflow(case); r2 = np.random.RandomState(1)
        for a in (u, v, w): a[1:-1, 1:-1, 1:-1] += 0.02 * (r2.rand(*ng) - 0.5)
        return u, v, w, p
    monkeypatch.setenv("CALES_UNFOLDED_MOM", "1")
    h = HotPath(case); h.upload(*perturbed()); h.startup(); dt = 0.5 * h.chkdt()
    for _ in range(nsteps): h.step(dt)
    ref = h.download() + [h.get("pp")]; h.close()
    monkeypatch.delenv("CALES_UNFOLDED_MOM")
    if lazy:
        monkeypatch.setenv("CALES_LAZY_PROJECTION", "1")

    def body(h, r):
        h.upload_global(*perturbed()); h.startup()
        h.profile(True)
        for _ in range(nsteps): h.step(dt)
        h.profile(False); st = h.profile_stats()
        ncorr = st.get("correc_updatep", (0, 0.))[0] + st.get("correc", (0, 0.))[0]
        local = h.get("u") if r == 0 else None      # rank-local: the other ranks are not inside the library while this runs
        return h.download() + [h.get("pp"), h.lo, h.n, ncorr, local]
    res = run_loopback(case, P, body)
    for r, R in enumerate(res):
        sl = slice(R[6][1], R[6][1] + R[7][1])
        assert R[8] == nsteps, (r, R[8])
        for q, nm in ((0, "u"), (1, "v"), (2, "w")):
            assert relerr(R[q][:, 1:-1, :], ref[q][:, sl, :]) < 1e-10, (r, nm)
    assert relerr(res[0][9][:, 1:-1, :], ref[0][:, 1:res[0][7][1] + 1, :]) < 1e-10
    for q, a, nm in ((3, 3, "p"), (5, 5, "pp")):
        d = np.concatenate([R[q][:, 1:-1, :] - ref[a][:, R[6][1]:R[6][1] + R[7][1], :] for R in res], axis=1)
        d = d - d[1:-1, :, 1:-1].mean()
        assert np.abs(d).max() < 1e-9 * np.abs(ref[a] - ref[a][1:-1, 1:-1, 1:-1].mean()).max(), nm


@pytest.mark.parametrize("name,ng,P", SLAB_CASES)
def test_slab_ranks_overlapped_event_ordered(name, ng, P, monkeypatch):
    """CALES_OVERLAP=1: the exchanges of the Poisson solve (k-chunks) and the y halos of the dynamic model's scratch fields run on the library's
    second stream beside kernels. The emulated ranks order their copies by recorded HIP events ONLY and delay every copy (decomp.LoopbackWorld,
    events=True): a consumer kernel that did not wait for the event of the data it reads would see the previous contents. (The default of the
    library is in order on one stream -- test_slab_ranks_match_single_rank above -- until a node with real peers has confirmed this path.)"""
    monkeypatch.setenv("CALES_OVERLAP", "1")
    monkeypatch.setenv("CALES_LOOPBACK_EVENTS", "1")
    test_slab_ranks_match_single_rank(name, ng, P)


@pytest.mark.parametrize("key,P", [("c3", 8), ("c4", 4)])
def test_slab_ranks_overlapped_event_ordered_at_baseline_sizes(key, P, monkeypatch):
    """The second-stream path at production sizes (several k-chunks per exchange, whole tiles beside the halo rows in flight), event-ordered emulation."""
    monkeypatch.setenv("CALES_OVERLAP", "1")
    monkeypatch.setenv("CALES_LOOPBACK_EVENTS", "1")
    test_slab_ranks_match_single_rank_at_baseline_sizes(key, P)


@pytest.mark.parametrize("name,ng,P", [("chan_dsmag", (128, 32, 136), 2), ("chan_dsmag_wm", (72, 32, 40), 4), ("cavity_nnn", (64, 32, 32), 2), ("duct_dsmag_wm", (32, 32, 32), 2)])
def test_slab_ranks_overlapped_host_synchronised(name, ng, P, monkeypatch):
    """The same second-stream path with the plain loopback exchanges (host waits around every copy)."""
    monkeypatch.setenv("CALES_OVERLAP", "1")
    test_slab_ranks_match_single_rank(name, ng, P)


@pytest.mark.parametrize("name,ng,P", [("chan_dsmag", (128, 32, 136), 2), ("halfchan_imp1d", (16, 16, 12), 2)])
def test_slab_ranks_in_order_exchanges(name, ng, P, monkeypatch):
    """CALES_NO_OVERLAP wins over CALES_OVERLAP; in-order exchanges also with the event-ordered emulation."""
    monkeypatch.setenv("CALES_OVERLAP", "1")
    monkeypatch.setenv("CALES_NO_OVERLAP", "1")
    monkeypatch.setenv("CALES_LOOPBACK_EVENTS", "1")
    test_slab_ranks_match_single_rank(name, ng, P)


@pytest.mark.parametrize("events", [0, 1])
@pytest.mark.parametrize("name,ng,P", [("chan_dsmag", (64, 32, 24), 4), ("duct_dsmag_wm", (32, 32, 32), 2), ("tgv_dsmag_ppp", (32, 24, 16), 3)])
def test_slab_ranks_batched_scratch_field_exchange(name, ng, P, events, monkeypatch):
    """The dynamic model's ghost-cell calls queue their rows and one exchange follows (in order): eleven fields when the last pass forms the cell-centred
    velocity itself, thirteen = two exchanges (twelve + one) when K_AC stores it (here through CALES_DSMAG_XGHOSTS, which also turns the pair fields off); plain and
    event-ordered emulation."""
    if events:
        monkeypatch.setenv("CALES_LOOPBACK_EVENTS", "1")
    test_slab_ranks_match_single_rank(name, ng, P)
    monkeypatch.setenv("CALES_DSMAG_XGHOSTS", "1")
    test_slab_ranks_match_single_rank(name, ng, P)


@pytest.mark.parametrize("name,ng,P", [("duct_smag_wm", (16, 32, 24), 4), ("duct_smag_wm_imp1d", (64, 32, 16), 8)])
def test_slab_ranks_smag_reference_sequence(name, ng, P, monkeypatch):
    """The other form of the static-Smagorinsky pass on more than two slabs between y walls: the kernel-per-loop sequence."""
    monkeypatch.setenv("CALES_SMAG_REFERENCE_SEQUENCE", "1")
    test_slab_ranks_match_single_rank(name, ng, P)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("name,ng,P", [("chan_dsmag", (32, 32, 16), 4), ("duct_smag_wm_imp1d", (16, 24, 12), 3), ("tgv_dsmag_ppp", (32, 24, 16), 2)])
def test_slab_ranks_with_switch_combinations(name, ng, P, seed, monkeypatch):
    """Several slabs with three to five run-time switches at once (fixed seeds; overlap on for every other one): same bar as the plain slab test."""
    rng = np.random.RandomState(2000 + seed)
    pool = ["CALES_UNFUSED_RK", "CALES_UNFUSED_CORREC", "CALES_UNFOLDED_CORREC", "CALES_UNFOLDED_MOM", "CALES_LAZY_PROJECTION", "CALES_UNFUSED_FORCING", "CALES_UNFUSED_FILLPS", "CALES_UNFUSED_MEAN", "CALES_GAUSSEL_MARCH", "CALES_NO_NYQUIST_PACKING",
            "CALES_DSMAG_XGHOSTS", "CALES_WIDE_OFFSETS", "CALES_UNMERGED_BC",
            "CALES_XGHOSTS_IN_STEP", "CALES_FFT_GENERIC", "CALES_HELMHOLTZ_Z_PER_COLUMN", "CALES_UNFUSED_IMP_RHS"]
    for k in rng.choice(pool, size=rng.randint(3, 6), replace=False):
        monkeypatch.setenv(str(k), "1")
    if seed % 2:
        monkeypatch.setenv("CALES_OVERLAP", "1")
    test_slab_ranks_match_single_rank(name, ng, P)


def test_slab_initflow_equals_global():
    """cales_initflow_slab (host only) == rows of the global initial field, bit for bit."""
    import ctypes as C
    from cales_amd import capi
    from cales_amd.hotpath import _p, initflow
    for name in ("chan_smag_wm", "duct_smag_wm", "tgv_ppp"):
        g, case = load_golden(name)
        u, v, w, p = initflow(case)
        P = 2
        n2l = int(case.ng[1]) // P
        for r in range(P):
            cs = capi.make_case(case, P, r)
            shape = (int(case.ng[0]) + 2, n2l + 2, int(case.ng[2]) + 2)
            loc = [np.zeros(shape, order="F") for _ in range(4)]
            assert capi.lib().cales_initflow_slab(C.byref(cs), case.inivel.encode(), int(case.is_wallturb), *[_p(a) for a in loc]) == 0
            for a, b in zip(loc, (u, v, w, p)):
                assert np.array_equal(a[1:-1, 1:-1, 1:-1], b[1:-1, r * n2l + 1:(r + 1) * n2l + 1, 1:-1])


def test_native_rccl_selftest():
    """comm_rccl.cpp on a one-rank communicator: send/recv order of the halo rows, all-to-all, all-reduce."""
    from cales_amd import capi
    assert capi.lib().cales_comm_selftest() == 0


@pytest.mark.parametrize("layer", ["rccl", "torch"])
def test_one_rank_nccl_process_group(layer):
    """The launch path of `bench.py --gpus N` with N = 1 forced through the slab layer (tests/_nccl1_worker.py), with the
    library's own RCCL exchanges and with the torch.distributed callbacks."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, CALES_COMM=layer, MASTER_PORT=str(29571 + (layer == "torch")))
    r = subprocess.run([sys.executable, os.path.join(here, "_nccl1_worker.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "NCCL1 OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("overlap", [0, 1])
@pytest.mark.parametrize("layer", ["rccl", "torch"])
def test_n_rank_nccl_process_group(layer, overlap):
    """Real peers: one process per GPU through torch.distributed.run, the exchanges of comm_rccl.cpp (grouped send/recv pairing incl.
    P = 2 periodic with both neighbours the same peer, ncclAllToAll block layout, all-reduces) against the single-rank run. RCCL
    refuses two ranks on one device, so this needs at least two GPUs and is skipped on the one-GPU test box."""
    import os, subprocess, sys
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs")
    P = 2 if n < 4 else 4
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, CALES_COMM=layer, CALES_OVERLAP=str(overlap))      # overlap = 1: k-chunked transposition and deferred halos on the second stream, with real peers
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={P}", "--master-addr", "127.0.0.1",
                        "--master-port", str(29581 + (layer == "torch") + 2 * overlap), os.path.join(here, "_ncclN_worker.py")],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "NCCLN OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("P", [2, 3, 8])      # 8: the rank count of the node the north star names (rank order, 32 packed mode columns in blocks of 4 -- 3: blocks of 11, one padding column on the last rank --, 8-way reductions)
def test_n_processes_one_gpu_staged_gloo(P):
    """Real processes (torch.distributed.run, one per rank) sharing the ONE GPU of the test box: RCCL refuses that, so the exchanges go
    through gloo with host staging (decomp.StagedGlooComm). Everything else is the production path of `bench.py --gpus N`: rendezvous, rank
    order, slab-wise initial fields, packing kernels, mode-block layout, reductions -- against the single-rank run, as the nccl test does."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, CALES_TEST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={P}", "--master-addr", "127.0.0.1",
                        "--master-port", str(29591 + P), os.path.join(here, "_ncclN_worker.py")],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "NCCLN OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_bench_two_processes_one_gpu():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), with the two ranks on the one GPU of the
    test box and gloo instead of RCCL (which refuses two ranks per device): the launch contract, the slab layer, the icheck blocks with their
    reductions over ranks, the max-over-ranks timing and the JSON line with its transposition report are executed end to end."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29611", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
                        "--ng", "64", "64", "32", "--backend", "gloo", "--no-cpu"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["warmup"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.) < 1e-9
    assert d["config"]["decomposition"] == "y-slabs x2" and "transpose" in d and d["transpose"]["alltoall_calls_per_step"] == 6.0
    assert d["icheck"]["blocks_in_timed_region"] >= 1


def test_bench_eight_processes_one_gpu_bare_form():
    """`python bench.py --gpus 8` in the bare form at the REAL rank count of the target node, the eight ranks sharing the one GPU of the test box through
    gloo: the 32 mode columns of 64-point rows (modes 0 and 32 share column 0) over 8 ranks in blocks of 4, 8-row slabs, max-over-ranks timing over eight
    processes, the 8-way rendezvous -- everything but RCCL itself."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1", "--ng", "64", "64", "32",
                        "--backend", "gloo", "--no-cpu"], capture_output=True, text=True, timeout=1500, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 4 and d["value"] > 0 and d["config"]["decomposition"] == "y-slabs x8"
    assert d["transpose"]["alltoall_calls_per_step"] == 6.0 and d["divmax"] < 1e-11
    assert "configs" not in d and "cpu_baseline" not in d      # side measurements belong to the N = 1 line only


def test_bench_bare_form_spawns_its_ranks():
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run (the form the driver records for N = 1): the parent, which never touches the GPU,
    starts the two ranks as child processes, relays rank 0's one JSON line and returns their exit code."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--ng", "64", "64", "32",
                        "--backend", "gloo", "--no-cpu"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and d["config"]["decomposition"] == "y-slabs x2"
    # a failing rank must fail the parent: an unknown size is refused by every rank
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--ng", "64", "63", "32",
                        "--backend", "gloo", "--no-cpu"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode != 0


def test_bench_hung_rank_fails_the_launch_within_the_timeout():
    """a rank that cannot rendezvous (here: rank 1 never calls init_process_group, bench.py's test hook) must make the launcher exit
    NON-ZERO within a stated time instead of waiting forever -- every rank arms a watchdog (--timeout seconds, exit status 124), torch.distributed.run
    ends the others, and the bare form's parent kills the whole process group if even that fails. Both launch forms."""
    import os, subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CALES_BENCH_TEST_HANG_RANK"] = "1"
    args = ["--gpus", "2", "--steps", "2", "--warmup", "0", "--ng", "64", "64", "32", "--backend", "gloo", "--no-cpu", "--timeout", "45"]
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode != 0 and time.time() - t0 < 240, (r.returncode, time.time() - t0, r.stderr[-2000:])
    assert "giving up (exit 124)" in r.stderr or "were killed" in r.stderr, r.stderr[-3000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]      # no bench line from a run that did not finish
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29621",
                        os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode != 0 and time.time() - t0 < 240, (r.returncode, time.time() - t0, r.stderr[-2000:])


def test_bench_reports_both_exchange_orders():
    """`bench.py --gpus N` times BOTH exchange orders in one invocation (in order = `value`, second stream under "overlap"), with the plan string of each.
    Real peers (RCCL) only: skipped on the one-GPU box, where the gloo form reports why the second order was not timed."""
    import json, os, subprocess, sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CALES_OVERLAP")}
    n = torch.cuda.device_count()
    backend = ["--backend", "gloo"] if n < 2 else []
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--ng", "64", "64", "32", "--no-cpu"] + backend,
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["exchange_order"].startswith("in order") and d["config"]["path"]["ranks"] == "2" and d["config"]["path"]["exchanges"] == "in_order"
    assert d["calibration"]["copy_GBps"] > 0 and d["roofline"]["frac_of_measured_copy_rate"] > 0
    if n < 2:
        assert "skipped" in d["overlap"]
    else:
        o = d["overlap"]
        assert o["ms_per_step"] > 0 and o["path"]["exchanges"] == "second_stream" and o["transpose"]["chunks_per_exchange"] > 1.5 and o["divmax"] < 1e-11


def test_bench_second_order_cannot_cost_the_first():
    """`bench.py --gpus N` measures the in-order exchanges first and the second-stream order after them on a fresh context. If that second part hangs on real
    peers (it has never met any), the first measurement must survive: behind a deadline rank 0 prints the line it already holds -- the in-order value, the
    reason under "overlap" -- and every rank leaves with status 0. Exercised with gloo ranks on the one GPU (test hooks of bench.py: the second measurement
    enabled for gloo, rank 1 never joining it), and once without the hang for the flow of two measurements in one invocation."""
    import json, os, subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CALES_OVERLAP")}
    env["CALES_BENCH_TEST_BOTH"] = "1"
    args = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--ng", "64", "64", "32", "--backend", "gloo", "--no-cpu"]
    r = subprocess.run(args, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["value"] > 0 and d["overlap"]["ms_per_step"] > 0 and d["overlap"]["path"]["ranks"] == "2" and d["overlap"]["divmax"] < 1e-11, d.get("overlap")
    env["CALES_BENCH_TEST_HANG_RANK2"] = "1"; env["CALES_BENCH_SOFT_S"] = "30"
    t0 = time.time()
    r = subprocess.run(args, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0 and time.time() - t0 < 300, (r.returncode, r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["n_gpus"] == 2 and "error" in d["overlap"], d.get("overlap")


@pytest.mark.parametrize("name,ng,P,rows2", [("chan_dsmag", (64, 32, 24), 4, True), ("chan_dsmag", (64, 64, 16), 8, True), ("tgv_dsmag_ppp", (64, 24, 16), 3, True),
                                             ("chan_dsmag", (64, 24, 16), 8, False)])      # (three rows per slab: the second ghost row would be the slab's own ghost row -- the one-row form)
def test_slab_halo_volume_of_the_dynamic_model(name, ng, P, rows2):
    """Several slabs, dynamic model with the projection folded into the strain-rate pass (reference halos src/bound.f90:619-723, the dynamic
    model's ghost-cell calls src/sgs.f90:191-197,256,280-282): with two ghost rows of the prediction and three of pp the pass forms the ghost rows of all its
    outputs, so a substep exchanges 6 + 3 = 9 field planes in 2 messages (round 5: 20 in 3) -- counted here from the callbacks of the emulated ranks, with the plan
    string saying which form ran, and the fields against the one-rank run as everywhere."""
    from cales_amd.decomp import run_loopback
    case = _case(name, ng)
    u, v, w, p, visct, dt, div, dpdl = _single(case, 2)
    calls = {}

    def body(h, r):
        h.upload_initial(); h.startup()
        pl = h.describe_plan()
        halo = h.comm.halo
        cnt = [0, 0]

        def counted(a, b, c_, d, n_):
            cnt[0] += 1; cnt[1] += n_
            return halo(a, b, c_, d, n_)
        h.comm.halo = counted
        h.step(dt); h.step(dt)
        h.comm.halo = halo
        calls[r] = (cnt[0], cnt[1], pl)
        return h.download() + [h.lo, h.n]

    res = run_loopback(case, P, body)
    n1, n3 = ng[0], ng[2]
    s1 = (n1 + 3 + 15) // 16 * 16
    plane = s1 * (n3 + 2)
    for r in range(P):
        ncall, nreal, pl = calls[r]
        assert pl["projection"] == ("in_strain_rate_pass(ghost_rows_local)" if rows2 else "in_strain_rate_pass"), pl
        per_substep = nreal / plane / 6.      # two steps of three substeps; the count of one direction's staging block
        assert ncall == (12 if rows2 else 18) and abs(per_substep - (9 if rows2 else 16)) < 1e-9, (ncall, per_substep)
        ur, vr, wr, pr, visr, lo, n = res[r]
        sl = slice(lo[1], lo[1] + n[1])
        for a, b, nm in ((ur, u, "u"), (vr, v, "v"), (wr, w, "w"), (visr, visct, "visct")):
            assert relerr(a[:, 1:-1, :], b[:, sl, :]) < 1e-10, (r, nm)
