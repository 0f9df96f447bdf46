#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ by importing the reference.

Run in the BUILD CONTAINER ONLY (the reference lives at /root/reference and never travels to
the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Nothing from the reference is copied: this script imports `golemflavor` from /root/reference,
calls its functions on seeded inputs and stores inputs + outputs as plain arrays
(`golden.npz`, loadable with allow_pickle=False) plus `golden_meta.json`.

Harness-side shims -- none of them edits a reference file:
  * `fractions.gcd` / `collections.Sequence` aliases so the py2-era modules import on py3.10;
  * texture branch: `fr.params_to_BSMu` is wrapped so texture T + (logLam,) is re-expressed as
    Texture.NONE + T's fixed angles (fr.py:370-376) + logLam -- algebraically the same call;
    the unwrapped branch builds a ragged array that numpy >= 1.24 rejects;
  * `gf.get_llh` (GolemFit is proprietary and absent) is replaced by the Gaussian substitute
    the README sanctions: multi_gaussian(angles_to_fr(astroFlavorAngle1,2), injected, smearing).
"""
import argparse
import collections
import collections.abc
import fractions
import json
import math
import os
import sys
import time

fractions.gcd = math.gcd
collections.Sequence = collections.abc.Sequence
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

from golemflavor import fr, gf, llh  # noqa: E402
from golemflavor.enums import Likelihood, ParamTag, PriorsCateg, Texture  # noqa: E402
from golemflavor.param import Param, ParamSet  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = {}
META = {"generator": "tests/golden/make_golden.py", "reference": "/root/reference (ShiveshM/GolemFlavor)",
        "numpy": np.__version__, "timings": {}}

Z = 0. + 1e-9
TEX = {Texture.OEU: (0.5, 1.0, Z, Z), Texture.OET: (Z, 0.25, Z, Z), Texture.OUT: (Z, 1.0, 0.5, Z)}
_orig_bsmu = fr.params_to_BSMu


def _bsmu_shim(bsm_angles, dim, energy, mass_eigenvalues=fr.MASS_EIGENVALUES, sm_u=fr.NUFIT_U,
               no_bsm=False, texture=Texture.NONE, check_uni=True, epsilon=1e-7):
    if texture in TEX:
        sc = bsm_angles[0] if isinstance(bsm_angles, (list, tuple)) else bsm_angles
        bsm_angles = tuple(TEX[texture]) + (sc,)
        texture = Texture.NONE
    return _orig_bsmu(bsm_angles, dim, energy, mass_eigenvalues=mass_eigenvalues, sm_u=sm_u,
                      no_bsm=no_bsm, texture=texture, check_uni=check_uni, epsilon=epsilon)


fr.params_to_BSMu = _bsmu_shim


def f64(x):
    return np.asarray(x, dtype=np.float64)


# ---------------------------------------------------------------- paramsets (as the reference builds them)
def notebook_paramsets():
    """examples/inference.ipynb cells 5-17."""
    src = fr.normalize_fr((1, 0, 0))
    meas = fr.u_to_fr(src, fr.NUFIT_U)
    ang = fr.fr_to_angles(meas)
    tag = ParamTag.BESTFIT
    asimov = ParamSet([
        Param(name='measured_angle1', value=ang[0], ranges=[0., 1.], std=0.02, tag=tag),
        Param(name='measured_angle2', value=ang[1], ranges=[-1., 1.], std=0.02, tag=tag)])
    tag = ParamTag.SM_ANGLES
    lg = PriorsCateg.LIMITEDGAUSS
    nuis = [
        Param(name='s_12_2', value=0.307, seed=[0.26, 0.35], ranges=[0., 1.], std=0.013, prior=lg, tag=tag),
        Param(name='c_13_4', value=(1 - (0.02206)) ** 2, seed=[0.950, 0.961], ranges=[0., 1.], std=0.00147, prior=lg, tag=tag),
        Param(name='s_23_2', value=0.538, seed=[0.31, 0.75], ranges=[0., 1.], std=0.069, prior=lg, tag=tag),
        Param(name='dcp', value=4.08404, seed=[0, 2 * np.pi], ranges=[0., 2 * np.pi], std=2.0, tag=tag)]
    tag = ParamTag.SRCANGLES
    srcp = [Param(name='source_angle1', value=0, ranges=[0., 1.], tag=tag),
            Param(name='source_angle2', value=0, ranges=[-1., 1.], tag=tag)]
    return asimov, ParamSet(nuis + srcp)


def notebook_triangle_llh(theta, asimov_paramset, llh_paramset):
    """examples/inference.ipynb cell 21, restated call for call (it is notebook-local)."""
    for idx, param in enumerate(llh_paramset):
        param.value = theta[idx]
    sm_u = fr.angles_to_u(llh_paramset.from_tag(ParamTag.SM_ANGLES, values=True))
    source_composition = fr.angles_to_fr(llh_paramset.from_tag(ParamTag.SRCANGLES, values=True))
    measured = fr.u_to_fr(source_composition, sm_u)
    bestfit = fr.angles_to_fr(asimov_paramset.from_tag(ParamTag.BESTFIT, values=True))
    smearing = asimov_paramset['measured_angle1'].std
    return llh.multi_gaussian(measured, bestfit, smearing), measured


def notebook_ln_prob(theta, asimov_paramset, llh_paramset):
    """examples/inference.ipynb cell 23."""
    lp = llh.lnprior(theta, paramset=llh_paramset)
    if not np.isfinite(lp):
        return -np.inf, np.full(3, np.nan)
    l, meas = notebook_triangle_llh(theta, asimov_paramset, llh_paramset)
    return lp + l, f64(meas)


def sm6_nuisance(with_priors=True):
    """scripts/mc_texture.py:28-49 / scripts/fr.py:30-50 (priors) or mc_unitary.py:28-41 (flat)."""
    tag = ParamTag.SM_ANGLES
    g, lg = PriorsCateg.GAUSSIAN, PriorsCateg.LIMITEDGAUSS
    e = 1e-9
    kw = (lambda p: dict(prior=p)) if with_priors else (lambda p: {})
    return [
        Param(name='s_12_2', value=0.307, seed=[0.26, 0.35], ranges=[0., 1.], std=0.013, tag=tag, **kw(lg)),
        Param(name='c_13_4', value=(1 - (0.02206)) ** 2, seed=[0.950, 0.961], ranges=[0., 1.], std=0.00147, tag=tag, **kw(lg)),
        Param(name='s_23_2', value=0.538, seed=[0.31, 0.75], ranges=[0., 1.], std=0.069, tag=tag, **kw(lg)),
        Param(name='dcp', value=4.08404, seed=[0 + e, 2 * np.pi - e], ranges=[0., 2 * np.pi], std=2.0, tag=tag),
        Param(name='m21_2', value=7.40E-23, seed=[7.2E-23, 7.6E-23], ranges=[6.80E-23, 8.02E-23], std=2.1E-24, prior=g, tag=tag),
        Param(name='m3x_2', value=2.494E-21, seed=[2.46E-21, 2.53E-21], ranges=[2.399E-21, 2.593E-21], std=3.3E-23, prior=g, tag=tag)]


def gf_nuisance():
    """scripts/fr.py:51-58."""
    tag = ParamTag.NUISANCE
    lg = PriorsCateg.LIMITEDGAUSS
    return [
        Param(name='convNorm', value=1., seed=[0.5, 2.], ranges=[0.1, 10.], std=0.4, prior=lg, tag=tag),
        Param(name='promptNorm', value=0., seed=[0., 6.], ranges=[0., 20.], std=2.4, prior=lg, tag=tag),
        Param(name='muonNorm', value=1., seed=[0.1, 2.], ranges=[0., 10.], std=0.1, tag=tag),
        Param(name='astroNorm', value=6.9, seed=[0., 5.], ranges=[0., 20.], std=1.5, tag=tag),
        Param(name='astroDeltaGamma', value=2.5, seed=[2.4, 3.], ranges=[-5., 5.], std=0.1, tag=tag)]


def scale_param(dimension):
    b = fr.SCALE_BOUNDARIES[dimension]
    return Param(name='logLam', value=np.mean(b), ranges=b, std=3, tag=ParamTag.SCALE)


def texture_paramset(dimension):          # scripts/mc_texture.py:52-76, 7-dim
    return ParamSet(sm6_nuisance() + [scale_param(dimension)])


def fr12_paramsets(dimension, injected):  # scripts/fr.py:62-104, 12-dim
    nuis = gf_nuisance()
    llh_ps = ParamSet(sm6_nuisance() + nuis + [scale_param(dimension)])
    ang = fr.fr_to_angles(injected)
    tag = ParamTag.BESTFIT
    asimov = ParamSet(nuis + [
        Param(name='astroFlavorAngle1', value=ang[0], ranges=[0., 1.], std=0.2, tag=tag),
        Param(name='astroFlavorAngle2', value=ang[1], ranges=[-1., 1.], std=0.2, tag=tag)])
    return asimov, llh_ps


def make_args(dimension, texture, source_ratio):
    return argparse.Namespace(
        binning=np.logspace(np.log10(6e4), np.log10(1e7), 21),   # scripts/fr.py:122-124
        source_ratio=fr.normalize_fr(source_ratio), dimension=dimension, no_bsm=False,
        texture=texture, likelihood=Likelihood.GOLEMFIT)


def uniform_in(ps, rng, n, seeds):
    box = np.array(ps.seeds if seeds else ps.ranges, dtype=float)
    return rng.uniform(box[:, 0], box[:, 1], size=(n, len(ps)))



# ---------------------------------------------------------------- exact evaluation (mpmath, 60 digits)
# lives in tests/exact_mp.py (also used by the GPU fuzz test)
sys.path.insert(0, os.path.dirname(HERE))
from exact_mp import (mp, mp_abs2, mp_angles_roundtrip_fr, mp_angles_to_u, mp_bsmu, mp_cardano,  # noqa: E402,F401
                      mp_flux_avg, mp_u_to_fr)


# ---------------------------------------------------------------- G1..G9
def g1_angles_to_u(rng):
    ang = np.column_stack([rng.uniform(0, 1, 256), rng.uniform(0, 1, 256), rng.uniform(0, 1, 256),
                           rng.uniform(0, 2 * np.pi, 256)])
    ang = np.vstack([ang, [0.307, (1 - 0.02195) ** 2, 0.565, 3.97935], [0.2, 0.3, 0.5, 1.5]])
    us = np.array([np.asarray(fr.angles_to_u(a)) for a in ang])     # complex256
    OUT["g1_angles"] = ang
    OUT["g1_u_re"] = f64(us.real)
    OUT["g1_u_im"] = f64(us.imag)
    OUT["g1_u_abs2"] = f64(np.abs(us) ** 2)


def g2_flavor_angles(rng):
    a = np.column_stack([rng.uniform(0, 1, 200), rng.uniform(-1, 1, 200)])
    a = np.vstack([a, [[0, 0], [1, 1], [1, -1], [0, 1], [0.3, 0.4], [1, 0], [0.25, 0.999999], [1e-12, -0.5]]])
    frs = np.array([fr.angles_to_fr(x) for x in a])
    OUT["g2_src_angles"] = a
    OUT["g2_fr"] = frs
    f = np.vstack([rng.dirichlet([1, 1, 1], 200), [[0, 0, 1], [1, 0, 0], [0, 1, 0], [1, 2, 0], [1, 1, 1], [2, 4, 6]]])
    back = np.array([[float(v) for v in fr.fr_to_angles(list(x))] for x in f])
    OUT["g2_fr_in"] = f
    OUT["g2_angles_back"] = back


def g3_u_to_fr(rng):
    srcs = [(1, 0, 0), (0, 1, 0), (1, 2, 0), (0, 0, 1), (1, 1, 1)] + [(x, 1 - x, 0) for x in np.linspace(0, 1, 8)]
    srcs = np.array([fr.normalize_fr(s) for s in srcs])
    OUT["g3_src"] = srcs
    OUT["g3_fr_nufit"] = np.array([f64(fr.u_to_fr(s, fr.NUFIT_U)) for s in srcs])
    # un-normalised source (u_to_fr divides by sum(src), fr.py:535)
    OUT["g3_src_raw"] = np.array([[1., 2., 0.], [3., 1., 2.]])
    OUT["g3_fr_raw"] = np.array([f64(fr.u_to_fr(s, fr.NUFIT_U)) for s in OUT["g3_src_raw"]])
    ang = OUT["g1_angles"][:64]
    OUT["g3_fr_120_rand"] = np.array([f64(fr.u_to_fr(fr.normalize_fr((1, 2, 0)), fr.angles_to_u(a))) for a in ang])


def g4_lnprior(rng):
    _, nb = notebook_paramsets()
    sets = {"c1": nb, "c3": ParamSet(sm6_nuisance(with_priors=False)[:4]),
            "c4": texture_paramset(6), "c5": fr12_paramsets(6, (1, 1, 1))[1]}
    for key, ps in sets.items():
        rows = [uniform_in(ps, rng, 48, seeds=True), uniform_in(ps, rng, 48, seeds=False)]
        lo = np.array(ps.ranges, dtype=float)[:, 0]
        hi = np.array(ps.ranges, dtype=float)[:, 1]
        mid = uniform_in(ps, rng, 1, seeds=True)[0]
        edge = []
        for i in range(len(ps)):
            for v in (lo[i], hi[i], np.nextafter(lo[i], -np.inf), np.nextafter(hi[i], np.inf)):
                t = mid.copy(); t[i] = v; edge.append(t)
        t = mid.copy(); t[0] = np.nan; edge.append(t)
        th = np.vstack(rows + [np.array(edge)])
        t0 = time.perf_counter()
        lp = np.array([llh.lnprior(list(x), paramset=ps) for x in th])
        META["timings"]["lnprior_%s_us" % key] = 1e6 * (time.perf_counter() - t0) / len(th)
        OUT["g4_%s_theta" % key] = th
        OUT["g4_%s_lnprior" % key] = f64(lp)


def g5_multi_gaussian(rng):
    bf = np.array([0.55003613487264495, 0.18301212999260349, 0.26695173513475156])
    pts = [rng.dirichlet([1, 1, 1], 64)]
    # walk outwards from the mode so that logpdf sweeps 9 ... -1100 and crosses the subnormal band densely
    d = np.array([-0.55, -0.18, 0.73]); d /= np.linalg.norm(d)
    r = np.sqrt(np.linspace(0, (0.95) ** 2, 600))
    pts.append(bf + r[:, None] * d)
    rb = 0.02 * np.sqrt(2 * (np.linspace(700, 750, 400) + 8.98))
    pts.append(bf + rb[:, None] * d)
    pts = np.vstack(pts)
    OUT["g5_fr"] = pts
    OUT["g5_bf"] = bf
    OUT["g5_llh"] = f64([llh.multi_gaussian(p, bf, 0.02) for p in pts])
    OUT["g5_llh_s01_off0"] = f64([llh.multi_gaussian(p, bf, 0.1, offset=0) for p in pts[:64]])


def g6_notebook(rng):
    asimov, ps = notebook_paramsets()
    OUT["g6_asimov_angles"] = f64([float(v) for v in asimov.from_tag(ParamTag.BESTFIT, values=True)])
    OUT["g6_bestfit_fr"] = f64(fr.angles_to_fr(asimov.from_tag(ParamTag.BESTFIT, values=True)))
    th = np.vstack([uniform_in(ps, rng, 4096, seeds=True), uniform_in(ps, rng, 1024, seeds=False),
                    [[0.307, 0.9564, 0.538, 4.08404, 0.9, 0.1], [1.2, 0.9564, 0.538, 4.08404, 0.9, 0.1]]])
    t0 = time.perf_counter()
    res = [notebook_ln_prob(list(x), asimov, ps) for x in th]
    META["timings"]["notebook_ln_prob_us"] = 1e6 * (time.perf_counter() - t0) / len(th)
    OUT["g6_theta"] = th
    OUT["g6_lnprob"] = f64([r[0] for r in res])
    OUT["g6_fr"] = np.array([r[1] for r in res])


def g7_bsmu(rng):
    centres = np.sqrt(make_args(3, Texture.OET, (1, 2, 0)).binning[:-1] * make_args(3, Texture.OET, (1, 2, 0)).binning[1:])
    rows, ures, uims, oks, exact = [], [], [], [], []
    for dim in (3, 6):
        lo, hi = fr.SCALE_BOUNDARIES[dim]
        for tex in (Texture.OEU, Texture.OET, Texture.OUT):
            for ll in np.linspace(lo, hi, 9):
                for e in centres[[0, 7, 13, 19]]:
                    try:
                        u = fr.params_to_BSMu((ll,), dim, e, texture=tex)
                        ok = 1
                    except AssertionError:
                        u = fr.params_to_BSMu((ll,), dim, e, texture=tex, check_uni=False)
                        ok = 0
                    rows.append([dim, tex.value, ll, e]); oks.append(ok)
                    exact.append(mp_abs2(mp_bsmu(TEX[tex], ll, dim, e, fr.MASS_EIGENVALUES,
                                                 mp_angles_to_u((0.307, (1 - 0.02195) ** 2, 0.565, 3.97935)))))
                    ures.append(f64(np.asarray(u).real)); uims.append(f64(np.asarray(u).imag))
    # docstring case fr.py:354-358 (texture NONE, angles from the tuple)
    u = fr.params_to_BSMu((0.2, 0.3, 0.5, 1.5, -20), dim=3, energy=1000)
    OUT["g7_doc_u_re"], OUT["g7_doc_u_im"] = f64(np.asarray(u).real), f64(np.asarray(u).imag)
    OUT["g7_doc_fr_120"] = f64(fr.u_to_fr((1, 2, 0), u))
    OUT["g7_rows"] = np.array(rows)
    OUT["g7_abs2_exact"] = np.array(exact)
    OUT["g7_u_re"], OUT["g7_u_im"], OUT["g7_ok"] = np.array(ures), np.array(uims), np.array(oks, dtype=np.int32)
    # cardano on a random Hermitian set
    hs, ms = [], []
    for _ in range(64):
        a = rng.normal(size=(3, 3)) + 1j * rng.normal(size=(3, 3))
        h = (a + a.conj().T) / 2
        hs.append(h); ms.append(np.asarray(fr.cardano_eqn(np.array(h, dtype=np.complex256))))
    OUT["g7_card_h_re"], OUT["g7_card_h_im"] = f64(np.real(hs)), f64(np.imag(hs))
    OUT["g7_card_m_re"], OUT["g7_card_m_im"] = f64(np.real(ms)), f64(np.imag(ms))


def g8_flux_avg(rng):
    rows, frs, sts, exact = [], [], [], []
    srcs = [(1, 2, 0), (0, 1, 0), (1, 0, 0), (0.3, 0.7, 0)]
    t_acc, n_acc = 0.0, 0
    for dim in (3, 6):
        ps = texture_paramset(dim)
        lo, hi = fr.SCALE_BOUNDARIES[dim]
        for tex in (Texture.OEU, Texture.OET, Texture.OUT):
            for si, src in enumerate(srcs):
                args = make_args(dim, tex, src)
                base = uniform_in(ps, rng, 6, seeds=True)
                base[:, 6] = np.linspace(lo, hi, 6)
                base[0, :6] = [0.307, 0.9564, 0.538, 4.08404, 7.4e-23, 2.494e-21]
                for th in base:
                    t0 = time.perf_counter()
                    try:
                        r = f64(fr.flux_averaged_BSMu(list(th), args, -2.0, ps)); st = 0
                    except AssertionError:
                        r = np.full(3, np.nan); st = 2
                    t_acc += time.perf_counter() - t0; n_acc += 1
                    rows.append(np.concatenate([[dim, tex.value, si], th])); frs.append(r); sts.append(st)
                    exact.append([float(v) for v in mp_flux_avg(th, TEX[tex], dim, args.source_ratio, args.binning)])
    META["timings"]["flux_averaged_BSMu_us"] = 1e6 * t_acc / n_acc
    OUT["g8_sources"] = np.array([fr.normalize_fr(s) for s in srcs])
    OUT["g8_rows"] = np.array(rows)
    OUT["g8_fr"] = np.array(frs)
    OUT["g8_fr_exact"] = np.array(exact)
    OUT["g8_status"] = np.array(sts, dtype=np.int32)


def g9_lnprob12(rng):
    inj = fr.normalize_fr((1, 1, 1))
    smear = 0.02
    gf.get_llh = lambda ps: llh.multi_gaussian(
        fr.angles_to_fr((ps['astroFlavorAngle1'].value, ps['astroFlavorAngle2'].value)), inj, smear)
    rows, vals, sts, exact = [], [], [], []
    t_acc, n_acc = 0.0, 0
    for dim, tex, src in ((6, Texture.OET, (0, 1, 0)), (3, Texture.OUT, (1, 0, 0)), (6, Texture.OEU, (1, 2, 0)), (3, Texture.OET, (1, 2, 0))):
        asimov, ps = fr12_paramsets(dim, inj)
        args = make_args(dim, tex, src)
        lo, hi = fr.SCALE_BOUNDARIES[dim]
        th = uniform_in(ps, rng, 128, seeds=True)
        th[:, 11] = rng.uniform(lo, hi - 0.25 * (hi - lo), 128)
        th[0] = [0.307, 0.9564, 0.538, 4.08404, 7.4e-23, 2.494e-21, 1.0, 0.5, 1.0, 6.9, 2.5, lo + 6]
        th[1, 2] = 1.5                                    # outside the box -> -inf
        for x in th:
            t0 = time.perf_counter()
            try:
                v = llh.ln_prob(list(x), args, asimov, ps); st = 0
            except AssertionError:
                v = np.nan; st = 2
            t_acc += time.perf_counter() - t0; n_acc += 1
            rows.append(np.concatenate([[dim, tex.value], fr.normalize_fr(src), x])); vals.append(v); sts.append(st)
            if np.all((x >= np.array(ps.ranges)[:, 0]) & (x <= np.array(ps.ranges)[:, 1])):
                fx = mp_angles_roundtrip_fr(mp_flux_avg(np.concatenate([x[:6], x[11:]]), TEX[tex], dim, args.source_ratio, args.binning))
                exact.append([float(t) for t in fx])
            else:
                exact.append([np.nan] * 3)
    META["timings"]["llh_ln_prob_12dim_us"] = 1e6 * t_acc / n_acc
    OUT["g9_rows"] = np.array(rows)
    OUT["g9_lnprob"] = f64(vals)
    OUT["g9_fr_exact"] = np.array(exact)
    OUT["g9_status"] = np.array(sts, dtype=np.int32)
    OUT["g9_injected"] = f64(inj)


def known_answers():
    """In-tree known answers, re-evaluated through the reference."""
    OUT["ka_nufit_re"] = f64(np.asarray(fr.NUFIT_U).real)
    OUT["ka_nufit_im"] = f64(np.asarray(fr.NUFIT_U).imag)
    OUT["ka_angles_to_fr_03_04"] = f64(fr.angles_to_fr((0.3, 0.4)))
    _, ps = notebook_paramsets()
    OUT["ka_c1_lnprior"] = f64(llh.lnprior([0.307, 0.9564, 0.538, 4.08404, 0.9, 0.1], ps))
    OUT["ka_c4_lnprior"] = f64(llh.lnprior([0.307, 0.9564, 0.538, 4.08404, 7.4e-23, 2.494e-21, -40], texture_paramset(6)))


def main():
    rng = np.random.default_rng(0)
    for fn in (g1_angles_to_u, g2_flavor_angles, g3_u_to_fr, g4_lnprior, g5_multi_gaussian, g6_notebook,
               g7_bsmu, g8_flux_avg, g9_lnprob12):
        t0 = time.perf_counter()
        fn(rng)
        print("%-20s %.1fs" % (fn.__name__, time.perf_counter() - t0), flush=True)
    known_answers()
    META["host"] = {"cpus": os.cpu_count(), "note": "single process, 1 core (build container)"}
    np.savez_compressed(os.path.join(HERE, "golden.npz"), **OUT)
    with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
        json.dump(META, f, indent=1, sort_keys=True)
    print("wrote", len(OUT), "arrays")


if __name__ == "__main__":
    main()
