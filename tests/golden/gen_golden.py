#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- golden vectors from the REFERENCE ITSELF.

Runs only in the build container (needs /root/reference and oracle/_ref built by
`make -C oracle/ref`). Each case = one of the reference's own examples/*/input.nml with the grid
shrunk (and a few switches changed, listed in CASES), run through the reference's compiled
operators in the order of its driver (src/main.f90:361-375 for the start-up, 417-507 for one
time step of three RK substeps). A seeded perturbation is added to the initial fields so that
no velocity component is identically zero.

ONE operator of that sequence cannot come from the reference: `solver` / `solver_gaussel_z`
(src/solver.f90) need FFTW + 2decomp-fft, which are not vendored and not in this image. Those
two calls are made by the CPU restatement (oracle/cales_oracle.c); their outputs are stored under
keys marked `_orc` and every *reference* operator downstream consumes them, so the stored
divergence after `correc` (reference `chkdiv`) certifies them independently.

Usage:  python tests/golden/gen_golden.py            # all cases (one subprocess per case)
        python tests/golden/gen_golden.py --case NAME
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
EX = "/root/reference/examples"
sys.path.insert(0, ROOT)

RK = [(32. / 60., 0.), (25. / 60., -17. / 60.), (45. / 60., -25. / 60.)]   # src/param.f90:27-29

# name: (example file, {regex: replacement}, impdiff)
CASES = {
    "tgv_ppp": ("dns/triperiodic/input.nml",
                {r"ng\(1:3\) = .*": "ng(1:3) = 12, 10, 8", r"l\(1:3\) = .*": "l(1:3) = 6.283185307179586, 6.283185307179586, 6.283185307179586",
                 r"visci = .*": "visci = 1600.", r"inivel = .*": "inivel = 'tgv'"}, 0),
    "tgv_dsmag_ppp": ("dns/triperiodic/input.nml",
                      {r"ng\(1:3\) = .*": "ng(1:3) = 12, 10, 8", r"l\(1:3\) = .*": "l(1:3) = 6.283185307179586, 6.283185307179586, 6.283185307179586",
                       r"visci = .*": "visci = 1600.", r"inivel = .*": "inivel = 'tgv'", r"sgstype = 'none'": "sgstype = 'dsmag'"}, 0),
    "chan_smag_wm": ("les/_manuscript_turbulent_channel_wall_model/input.nml",
                     {r"ng\(1:3\) = .*": "ng(1:3) = 12, 8, 10", r"visci = .*": "visci = 5640."}, 0),
    "chan_smag": ("les/_manuscript_turbulent_channel/input.nml",
                  {r"ng\(1:3\) = .*": "ng(1:3) = 10, 12, 8", r"gr = 5\.": "gr = 2."}, 0),
    "chan_dsmag": ("les/_manuscript_turbulent_channel/input.nml",
                   {r"ng\(1:3\) = .*": "ng(1:3) = 12, 8, 10", r"gr = 5\.": "gr = 2.", r"sgstype = 'smag'": "sgstype = 'dsmag'"}, 0),
    "chan_dsmag_wm": ("les/_manuscript_turbulent_channel_wall_model/input.nml",
                      {r"ng\(1:3\) = .*": "ng(1:3) = 12, 8, 10", r"visci = .*": "visci = 5640.", r"sgstype = 'smag'": "sgstype = 'dsmag'"}, 0),
    # hwm: the example's 0.1 equals zc(1) of the shrunk grid, which sanity.f90:224-231 refuses (zc(1) < hwm)
    "duct_smag_wm": ("les/_manuscript_turbulent_duct_wall_model/input.nml",
                     {r"ng\(1:3\) = .*": "ng(1:3) = 8, 10, 10", r"hwm = 0\.1": "hwm = 0.25"}, 0),
    "duct_smag_wm_imp1d": ("les/_manuscript_turbulent_duct_wall_model/input.nml",
                           {r"ng\(1:3\) = .*": "ng(1:3) = 8, 10, 10", r"hwm = 0\.1": "hwm = 0.25"}, 2),
    # dynamic Smagorinsky with walls in y/z or x/y/z: the kernel-per-loop sequence of sgs.f90:153-380 and
    # the x/y branches of `extrapolate` (sgs.f90:719-766)
    "duct_dsmag_wm": ("les/_manuscript_turbulent_duct_wall_model/input.nml",
                      {r"ng\(1:3\) = .*": "ng(1:3) = 8, 10, 10", r"sgstype = 'smag'": "sgstype = 'dsmag'", r"hwm = 0\.1": "hwm = 0.25"}, 0),
    "duct_dsmag": ("les/_manuscript_turbulent_duct_wall_model/input.nml",
                   {r"ng\(1:3\) = .*": "ng(1:3) = 8, 10, 12", r"sgstype = 'smag'": "sgstype = 'dsmag'",
                    r"lwm\(0:1,1:3\) = .*": "lwm(0:1,1:3) = 0,0, 0,0, 0,0", r"gr = 0\.": "gr = 1.5"}, 0),
    "cavity_dsmag": ("dns/lid_driven_cavity/input.nml",
                     {r"ng\(1:3\) = .*": "ng(1:3) = 10, 8, 12", r"gr = 0\.": "gr = 1.5", r"sgstype = 'none'": "sgstype = 'dsmag'"}, 0),
    "cavity_nnn": ("dns/lid_driven_cavity/input.nml",
                   {r"ng\(1:3\) = .*": "ng(1:3) = 10, 8, 12", r"gr = 0\.": "gr = 1.5"}, 0),
    "devchan_nd": ("dns/developing_channel/input.nml",
                   {r"ng\(1:3\) = .*": "ng(1:3) = 10, 8, 8", r"gr = 0\.": "gr = 1."}, 0),
    "halfchan_imp1d": ("dns/half_channel/input.nml",
                       {r"ng\(1:3\) = .*": "ng(1:3) = 8, 10, 12", r"gr = 0\.": "gr = 2.", r"inivel = .*": "inivel = 'hcp'"}, 2),
    "couette_imp3d_ops": ("dns/couette/input.nml",
                          {r"ng\(1:3\) = .*": "ng(1:3) = 8, 8, 10", r"gr = 0\.": "gr = 1."}, 1),
}
# End-of-step states at POWER-OF-TWO row lengths (only the raw initial fields and the state after one step are kept: the stage-by-stage vectors above
# would be ~3 MB per case at these sizes): there cales_step takes the radix-8 transforms, forms fillps inside the forward x pass and leaves the x ghost
# columns alone until it returns -- paths the 8..12-point rows of the cases above never reach.
END_ONLY = {
    "chan_dsmag_p2": ("les/_manuscript_turbulent_channel/input.nml",
                      {r"ng\(1:3\) = .*": "ng(1:3) = 32, 16, 12", r"gr = 5\.": "gr = 2.", r"sgstype = 'smag'": "sgstype = 'dsmag'"}, 0),
    "chan_smag_p2": ("les/_manuscript_turbulent_channel/input.nml",
                     {r"ng\(1:3\) = .*": "ng(1:3) = 64, 12, 10", r"gr = 5\.": "gr = 2."}, 0),
    "duct_dsmag_p2": ("les/_manuscript_turbulent_duct_wall_model/input.nml",
                      {r"ng\(1:3\) = .*": "ng(1:3) = 16, 12, 12", r"sgstype = 'smag'": "sgstype = 'dsmag'",
                       r"lwm\(0:1,1:3\) = .*": "lwm(0:1,1:3) = 0,0, 0,0, 0,0", r"gr = 0\.": "gr = 1.5"}, 0),
    "tgv_ppp_p2": ("dns/triperiodic/input.nml",
                   {r"ng\(1:3\) = .*": "ng(1:3) = 32, 16, 16", r"l\(1:3\) = .*": "l(1:3) = 6.283185307179586, 6.283185307179586, 6.283185307179586",
                    r"visci = .*": "visci = 1600.", r"inivel = .*": "inivel = 'tgv'"}, 0),
    # rows of 64 cells = one whole tile in x: the only row lengths at which the dynamic model's strain-rate pass takes the projection on load
    # (k_corr_strain_tile, api.hip `fold_correc`: n1 % 64 == 0) -- the dominant kernel of the 512^3 bench
    "chan_dsmag_x64": ("les/_manuscript_turbulent_channel/input.nml",
                       {r"ng\(1:3\) = .*": "ng(1:3) = 64, 16, 12", r"gr = 5\.": "gr = 2.", r"sgstype = 'smag'": "sgstype = 'dsmag'"}, 0),
    # ... and a row of TWO 64-cell tiles (and two 10-row tiles in y): the x seam of k_corr_strain_tile -- lane 63's pp(i+1) from the halo column, the halo
    # waves' side job -- held to the reference itself and not only to the oracle
    "chan_dsmag_x128": ("les/_manuscript_turbulent_channel/input.nml",
                        {r"ng\(1:3\) = .*": "ng(1:3) = 128, 12, 10", r"gr = 5\.": "gr = 2.", r"sgstype = 'smag'": "sgstype = 'dsmag'"}, 0),
    "tgv_dsmag_ppp_x64": ("dns/triperiodic/input.nml",
                          {r"ng\(1:3\) = .*": "ng(1:3) = 64, 16, 16", r"l\(1:3\) = .*": "l(1:3) = 6.283185307179586, 6.283185307179586, 6.283185307179586",
                           r"visci = .*": "visci = 1600.", r"inivel = .*": "inivel = 'tgv'", r"sgstype = 'none'": "sgstype = 'dsmag'"}, 0),
    # static Smagorinsky at rows of 64 cells: the projection folded into the Smagorinsky pass (k_corr_smag_tile, k_smagfold.hip: whole 64-cell tiles in x)
    # with a wall model on the z walls, on the four walls of a duct, and with z-implicit diffusion (the pressure update with the z Laplacian of pp)
    "chan_smag_wm_x64": ("les/_manuscript_turbulent_channel_wall_model/input.nml",
                         {r"ng\(1:3\) = .*": "ng(1:3) = 64, 16, 10", r"visci = .*": "visci = 5640."}, 0),
    "duct_smag_wm_x64": ("les/_manuscript_turbulent_duct_wall_model/input.nml",
                         {r"ng\(1:3\) = .*": "ng(1:3) = 64, 10, 10", r"hwm = 0\.1": "hwm = 0.25"}, 0),
    "duct_smag_wm_imp1d_x64": ("les/_manuscript_turbulent_duct_wall_model/input.nml",
                               {r"ng\(1:3\) = .*": "ng(1:3) = 64, 10, 12", r"hwm = 0\.1": "hwm = 0.25"}, 2),
    # more fast paths at 64-cell rows, each reached only there: the projection folded into the NEXT momentum pass (k_momrk<.., CORR>) for a DNS channel with bulk
    # forcing (x ghost columns left alone), a cavity (Neumann transforms in the radix-8 kernels, every ghost cell through k_bc_all, the velocity's corrected
    # view) and a half channel with z-implicit diffusion (velocity only folded, CORR = 2); the dynamic model's tile passes with walls in y (k_strain_tile<YW>,
    # k_lmf_tile<YW>); the triply periodic box with the periodic-z tridiagonal tile
    "chan_nosgs_x64": ("les/_manuscript_turbulent_channel/input.nml",
                       {r"ng\(1:3\) = .*": "ng(1:3) = 64, 16, 12", r"gr = 5\.": "gr = 2.", r"sgstype = 'smag'": "sgstype = 'none'"}, 0),
    "cavity_nnn_x64": ("dns/lid_driven_cavity/input.nml",
                       {r"ng\(1:3\) = .*": "ng(1:3) = 64, 16, 12", r"gr = 0\.": "gr = 1.5"}, 0),
    "halfchan_imp1d_x64": ("dns/half_channel/input.nml",
                           {r"ng\(1:3\) = .*": "ng(1:3) = 64, 16, 12", r"gr = 0\.": "gr = 2.", r"inivel = .*": "inivel = 'hcp'"}, 2),
    "duct_dsmag_x64": ("les/_manuscript_turbulent_duct_wall_model/input.nml",
                       {r"ng\(1:3\) = .*": "ng(1:3) = 64, 12, 12", r"sgstype = 'smag'": "sgstype = 'dsmag'",
                        r"lwm\(0:1,1:3\) = .*": "lwm(0:1,1:3) = 0,0, 0,0, 0,0", r"gr = 0\.": "gr = 1.5"}, 0),
    "tgv_ppp_x64": ("dns/triperiodic/input.nml",
                    {r"ng\(1:3\) = .*": "ng(1:3) = 64, 16, 16", r"l\(1:3\) = .*": "l(1:3) = 6.283185307179586, 6.283185307179586, 6.283185307179586",
                     r"visci = .*": "visci = 1600.", r"inivel = .*": "inivel = 'tgv'"}, 0),
}
END_KEYS = ("input_nml", "impdiff", "dt", "dt_cfl", "dpdl", "r3_div", "s0raw_u", "s0raw_v", "s0raw_w", "s0raw_p", "r3_s7_u", "r3_s7_v", "r3_s7_w", "r3_s8_p", "r3_s9_visct")
CASES.update(END_ONLY)
GRIDS = [(1, 0., 12), (1, 3.2, 12), (2, 1.7, 10), (3, 2.1, 9), (4, 1.3, 14), (5, 0., 24), (6, 0., 20), (6, 0., 128)]


def make_nml(name):
    path, subs, _ = CASES[name]
    text = open(os.path.join(EX, path)).read()
    for pat, rep in subs.items():
        text, k = re.subn(pat, rep, text, count=1)
        assert k == 1, (name, pat)
    # examples/dns/* close &les with a backslash, which the amdflang run-time refuses (gfortran,
    # the reference's CI compiler, accepts it): normalise for the reference build only
    text = re.sub(r"(?m)^\\\s*$", "/", text)
    return text


def run_case(name, out):
    from cales_amd.nml import parse_text
    from oracle.oracle import Oracle
    from oracle.ref.refpy import Ref
    _, _, imp = CASES[name]
    text = make_nml(name)
    tmp = tempfile.mkdtemp(prefix="gold_")
    open(os.path.join(tmp, "input.nml"), "w").write(text)
    os.chdir(tmp)
    ref = Ref(imp)
    case = parse_text(text); case.impdiff = imp
    G = {"input_nml": np.array(text), "impdiff": np.array(imp)}
    par = ref.params()
    for k, v in par.items():
        G["par_" + k] = np.array(v)
    for k, v in ref.grid().items():
        G["g_" + k] = v
    rx, ry, rz = ref.rhsbp(); G["rhsbp_x"], G["rhsbp_y"], G["rhsbp_z"] = rx, ry, rz

    if imp == 1:
        # 3-D implicit build: only the operators whose arithmetic changes (mom, rk, updatep) -- its
        # Helmholtz solves need FFTW-r2r kinds the image cannot provide
        u, v, w, p = ref.initflow()
        rng = np.random.RandomState(7)
        for a in (u, v, w, p):
            a[1:-1, 1:-1, 1:-1] += 0.05 * (rng.rand(*ref.n) - 0.5)
        ref.bounduvw(u, v, w); ref.boundp(p)
        visct = ref.zeros(); visct[...] = 1e-3 * rng.rand(*ref.shape)
        G.update(s0_u=u.copy(), s0_v=v.copy(), s0_w=w.copy(), s0_p=p.copy(), s0_visct=visct.copy())
        m = ref.mom(u, v, w, visct)
        for k, a in zip(("dudt", "dvdt", "dwdt", "dudtd", "dvdtd", "dwdtd"), m):
            G["m_" + k] = a
        dt = 0.01; G["dt"] = np.array(dt)
        for irk in (1, 2):
            f = ref.rk(irk, dt, p, visct, u, v, w)
            G.update({f"r{irk}_s1_u": u.copy(), f"r{irk}_s1_v": v.copy(), f"r{irk}_s1_w": w.copy(), f"r{irk}_s1_f": f})
        pp = ref.zeros(); pp[...] = rng.rand(*ref.shape) - 0.5
        G["upd_pp"] = pp.copy(); alpha = -.5 * par["visc"] * dt
        ref.updatep(alpha, pp, p); G["upd_alpha"] = np.array(alpha); G["upd_p"] = p.copy()
        G["dt_cfl"] = np.array(ref.chkdt(visct, u, v, w))
        np.savez_compressed(out, **G)
        ref.finalize()
        return

    orc = Oracle(case)
    # operands and z sweep of the Poisson solve by the reference's own eigenvalues / tridmatrix / gaussel (initsolver.f90:66-169,
    # solver.f90:82-179; compiled from their lines, oracle/ref/Makefile)
    gr_ = ref.grid(); dzci_ref, dzfi_ref = 1. / gr_["dzc"], 1. / gr_["dzf"]
    cbp = par["cbcpre"]
    lamx = ref.eigenvalues(ref.n[0], cbp[:, 0], "c"); lamy = ref.eigenvalues(ref.n[1], cbp[:, 1], "c")
    sa, sb, sc = ref.tridmatrix(cbp[:, 2], ref.n[2], par["dli"][2], dzci_ref, dzfi_ref, "c")
    G.update(sol_lamx=lamx, sol_lamy=lamy, sol_a=sa, sol_b=sb, sol_c=sc)
    rngz = np.random.RandomState(2024)
    pz = np.asfortranarray(rngz.rand(*ref.n) - 0.5); G["sol_gz_in"] = pz.copy()
    lamxy = np.asfortranarray(lamx[:, None] * par["dli"][0] ** 2 + lamy[None, :] * par["dli"][1] ** 2)
    ref.gaussel(pz, sa, sb, sc, ref.n[2], 0, lamxy, periodic=(cbp[0, 2] == "P" and cbp[1, 2] == "P"))
    G["sol_gz_out"] = pz
    u, v, w, p = ref.initflow()
    G.update(if_u=u.copy(), if_v=v.copy(), if_w=w.copy(), if_p=p.copy())
    rng = np.random.RandomState(12345)
    for a in (u, v, w, p):
        a[1:-1, 1:-1, 1:-1] += 0.05 * (rng.rand(*ref.n) - 0.5)
    G.update(s0raw_u=u.copy(), s0raw_v=v.copy(), s0raw_w=w.copy(), s0raw_p=p.copy())
    # start-up, src/main.f90:370-375
    ref.bounduvw(u, v, w, True, False)
    ref.boundp(p, 0)
    visct = ref.zeros()
    ref.cmpt_sgs(u, v, w, visct)
    G["s0_visct_nobc"] = visct.copy()
    ref.boundp(visct, 1)
    G.update(s0_u=u.copy(), s0_v=v.copy(), s0_w=w.copy(), s0_p=p.copy(), s0_visct=visct.copy())
    for iv, nm in ((1, "bcu"), (2, "bcv"), (3, "bcw")):
        x, y, z = ref.bcvel_planes(iv); G.update({f"s0_{nm}_x": x, f"s0_{nm}_y": y, f"s0_{nm}_z": z})
    dt_cfl = ref.chkdt(visct, u, v, w); G["dt_cfl"] = np.array(dt_cfl)
    dt = 0.5 * min(par["cfl"] * dt_cfl, par["dtmax"]); G["dt"] = np.array(dt)
    m = ref.mom(u, v, w, visct)
    for k, a in zip(("dudt", "dvdt", "dwdt", "dudtd", "dvdtd", "dwdtd"), m):
        G["m_" + k] = a
    G["mean_u_f"] = np.array(ref.bulk_mean(u, "f")); G["mean_w_c"] = np.array(ref.bulk_mean(w, "c"))
    d0 = ref.chkdiv(u, v, w); G["div0"] = np.array(d0)
    pp = ref.zeros()
    dpdl = np.zeros(3)
    for irk in (1, 2, 3):
        K = f"r{irk}_"
        dtrk = (RK[irk - 1][0] + RK[irk - 1][1]) * dt; dtrki = dtrk ** (-1)
        f = ref.rk(irk, dt, p, visct, u, v, w)
        ref.bulk_forcing(f, u, v, w)
        G.update({K + "s1_u": u.copy(), K + "s1_v": v.copy(), K + "s1_w": w.copy(), K + "s1_f": f.copy()})
        alpha = 0.
        if imp == 2:
            alpha = -.5 * par["visc"] * dtrk
            for iv, q in ((1, u), (2, v), (3, w)):
                ref.updt_rhs_b_velz(iv, alpha, q)
            G.update({K + "s1a_u": u.copy(), K + "s1a_v": v.copy(), K + "s1a_w": w.copy()})
            # the same sweeps by the reference's own gaussel / tridmatrix (solver.f90:82-151,182-233 without the transposes; initsolver.f90:127-169,
            # main.f90:435-445): solver_gaussel_z is a transposition around exactly this call
            uvw_ref = []
            for iv, q in ((1, u), (2, v), (3, w)):
                cbz = par["cbcvel_after_initbc"][:, 2, iv - 1]; cf = "f" if iv == 3 else "c"
                a_, b_, c_ = ref.tridmatrix(cbz, ref.n[2], par["dli"][2], dzci_ref, dzfi_ref, cf)
                qq = 1 if (cf == "f" and cbz[1] == "D") else 0
                pz = np.asfortranarray(q[1:-1, 1:-1, 1:-1].copy())
                ref.gaussel(pz, a_ * alpha, b_ * alpha + 1., c_ * alpha, ref.n[2] - qq, 0, None, periodic=(cbz[0] == "P" and cbz[1] == "P"))
                r_ = q.copy(); r_[1:-1, 1:-1, 1:-1] = pz; uvw_ref.append(r_)
            G.update({K + "s1b_u": uvw_ref[0], K + "s1b_v": uvw_ref[1], K + "s1b_w": uvw_ref[2]})
            for iv, q in ((1, u), (2, v), (3, w)):
                orc.solver_gaussel_z(iv, alpha, q)
            G.update({K + "s1b_u_orc": u.copy(), K + "s1b_v_orc": v.copy(), K + "s1b_w_orc": w.copy()})
        dpdl += f
        ref.bounduvw(u, v, w, True, False)
        G.update({K + "s2_u": u.copy(), K + "s2_v": v.copy(), K + "s2_w": w.copy()})
        if np.any(par["lwm"] != 0):
            for iv, nm in ((1, "bcu"), (2, "bcv"), (3, "bcw")):
                x, y, z = ref.bcvel_planes(iv); G.update({f"{K}s2_{nm}_x": x, f"{K}s2_{nm}_y": y, f"{K}s2_{nm}_z": z})
        ref.fillps(dtrki, u, v, w, pp)
        ref.updt_rhs_b_p(pp)
        G[K + "s3_pp"] = pp.copy()
        orc.solver(pp)
        # (interior of s5_pp below == the oracle solver's output)
        ref.boundp(pp, 0)
        G[K + "s5_pp"] = pp.copy()
        ref.correc(dtrk, pp, u, v, w)
        if irk == 1:
            G.update({K + "s6_u": u.copy(), K + "s6_v": v.copy(), K + "s6_w": w.copy()})
        ref.bounduvw(u, v, w, True, True)
        G.update({K + "s7_u": u.copy(), K + "s7_v": v.copy(), K + "s7_w": w.copy()})
        ref.updatep(alpha, pp, p)
        ref.boundp(p, 0)
        G[K + "s8_p"] = p.copy()
        ref.cmpt_sgs(u, v, w, visct)
        ref.boundp(visct, 1)
        G[K + "s9_visct"] = visct.copy()
        G[K + "div"] = np.array(ref.chkdiv(u, v, w))
    G["dpdl"] = -dpdl / dt
    G["dt_cfl_end"] = np.array(ref.chkdt(visct, u, v, w))
    # plane statistics of the end-of-step state by the reference's own out1d_single_point_chan (src/output.f90:509-1061; built from
    # the routine's lines by oracle/ref/Makefile): 27 single-point sums, 38 budget sums, 6 divergence measures per plane
    st, bud, leak = ref.out1d_single_point_chan(u, v, w, p, visct)
    G["st_chan"], G["st_budget"], G["st_leak"] = st, bud, leak
    if name in END_ONLY:
        G = {k: G[k] for k in END_KEYS}
    np.savez_compressed(out, **G)
    print(name, "divmax after step:", G["r3_div"][1], "file KB:", os.path.getsize(out) // 1024)
    ref.finalize()


def run_grids(out):
    """initgrid for all stretching functions (src/initgrid.f90); through ref_init on a dummy case."""
    base = open(os.path.join(EX, "dns/triperiodic/input.nml")).read()
    G = {}
    for q, (gtype, gr, n3) in enumerate(GRIDS):
        code = f"""
import os, sys, numpy as np
sys.path.insert(0, {ROOT!r})
from oracle.ref.refpy import Ref
os.chdir({{tmp!r}})
r = Ref(0); g = r.grid()
np.savez({{tmp!r}} + '/g.npz', **g); r.finalize()
"""
        tmp = tempfile.mkdtemp(prefix="gridg_")
        text = re.sub(r"ng\(1:3\) = .*", f"ng(1:3) = 4, 4, {n3}", base)
        text = re.sub(r"gtype = 1, gr = 0\.", f"gtype = {gtype}, gr = {gr!r}", text)
        text = re.sub(r"l\(1:3\) = .*", "l(1:3) = 1., 1., 2.", text)
        text = re.sub(r"(?m)^\\\s*$", "/", text)
        open(os.path.join(tmp, "input.nml"), "w").write(text)
        subprocess.check_call([sys.executable, "-c", code.format(tmp=tmp)], stdout=subprocess.DEVNULL)
        g = np.load(os.path.join(tmp, "g.npz"))
        for k in g.files:
            G[f"g{q}_{k}"] = g[k]
        G[f"g{q}_spec"] = np.array([gtype, gr, n3, 2.0])
    np.savez_compressed(out, **G)
    print("grids: ok")


def run_outstats(out):
    """out1d, out1d_chan and out2d_duct of the reference (src/output.f90:50-163, 317-507; compiled from their lines, oracle/ref/Makefile) on the
    end-of-step state of every full golden case (taken from the committed vectors: no case is re-run). The routines write text files of 8
    significant digits; what they print is stored as it is read back."""
    from oracle.ref.refpy import Ref
    G = {}
    for name in CASES:
        _, _, imp = CASES[name]
        if imp == 1:
            continue                    # (operator-level vectors only: no end-of-step state)
        g = np.load(os.path.join(HERE, name + ".npz"))
        code = f"""
import os, sys, tempfile, numpy as np
sys.path.insert(0, {ROOT!r})
from oracle.ref.refpy import Ref
g = np.load({os.path.join(HERE, name + '.npz')!r})
tmp = tempfile.mkdtemp(prefix="gold_"); open(os.path.join(tmp, "input.nml"), "w").write(str(g["input_nml"])); os.chdir(tmp)
ref = Ref({imp})
F = lambda a: np.asfortranarray(a)
u, v, w = F(g["r3_s7_u"]), F(g["r3_s7_v"]), F(g["r3_s7_w"])
o = {{}}
for key, (idir, fld, dzc) in dict(u_z=(3, u, 0), v_y=(2, v, 0), w_x=(1, w, 1), w_z=(3, w, 1), u_y=(2, u, 0)).items():
    x, y = ref.out1d(idir, fld, bool(dzc)); o["out1d_" + key] = np.stack([x, y])
o["out1d_chan"] = ref.out1d_chan(u, v, w)
o["out2d_duct"] = ref.out2d_duct(u, v, w)
np.savez(os.path.join(tmp, "o.npz"), **o); print(tmp)
ref.finalize()
"""
        tmp = subprocess.check_output([sys.executable, "-c", code], text=True).strip().splitlines()[-1]
        o = np.load(os.path.join(tmp, "o.npz"))
        for k in o.files:
            G[name + "__" + k] = o[k]
        print("outstats:", name, "ok")
    np.savez_compressed(out, **G)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default=None)
    a = ap.parse_args()
    if a.case == "grids":
        run_grids(os.path.join(HERE, "grids.npz"))
    elif a.case == "outstats":
        run_outstats(os.path.join(HERE, "outstats.npz"))
    elif a.case:
        run_case(a.case, os.path.join(HERE, a.case + ".npz"))
    else:
        for name in CASES:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--case", name])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--case", "grids"])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--case", "outstats"])
    # the manifest follows whatever was (re)generated: case list, grid list and a digest of every vector file
    import hashlib
    json.dump({"cases": list(CASES), "grids": GRIDS,
               "files": {f: hashlib.sha256(open(os.path.join(HERE, f), "rb").read()).hexdigest()[:16] for f in sorted(os.listdir(HERE)) if f.endswith(".npz")}},
              open(os.path.join(HERE, "manifest.json"), "w"), indent=1)
