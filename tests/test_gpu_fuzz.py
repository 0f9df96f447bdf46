"""GPU: seeded differential fuzzing of the SM / PRIOR_ONLY kernels against the oracle.

Random posteriors -- which of the six physics parameters are sampled and in which column, extra nuisance columns
with random prior kinds, row lengths 1..16, AoS / SoA, host and device entry points, ragged sizes either side of
the fast-kernel / tail-kernel / zero-copy thresholds -- evaluated on random walkers (a share of them outside the
prior box).  Every configuration is a different kernel instantiation or dispatch path; all must agree with the
long-double oracle to the parity bar."""
import numpy as np
import pytest

from common import rel_err
from golemflavor_amd import _lib
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import ParamTag, PriorsCateg
from golemflavor_amd.model import GF_LAYOUT_AOS, GF_LAYOUT_SOA, Model
from golemflavor_amd.param import Param, ParamSet

pytestmark = pytest.mark.gpu

REL = 1e-10
ABS_FR = 1e-10


_X87 = {}


def _emulated_host_verdict(oracle, desc, row, edges, dim, tex):
    """Non-unitary by the host build of gf_x87.hpp (tests/x87/x87_host.cpp, g++): the reference's chain, bin by bin, with the
    SM matrix from the emulated functions and the texture's matrix in long double, as the device has them."""
    import ctypes as C
    import math
    import os
    import subprocess
    import tempfile
    if "lib" not in _X87:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        out = os.path.join(tempfile.mkdtemp(prefix="x87host"), "libx87host.so")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-fPIC", "-shared", "-ffp-contract=off", "-o", out,
                               os.path.join(root, "tests", "x87", "x87_host.cpp")])
        L = C.CDLL(out)
        L.x87t_bin_residual.restype = C.c_double
        L.x87t_bin_residual.argtypes = [C.POINTER(C.c_double)] * 2 + [C.c_double] * 4 + [C.c_int, C.c_void_p, C.c_void_p]
        _X87["lib"] = L
    L = _X87["lib"]
    z = 1e-9
    tex_ang = {1: (0.5, 1.0, z, z), 2: (z, 0.25, z, z), 3: (z, 1.0, 0.5, z)}[int(tex.value)]
    arr = lambda x: (C.c_double * len(x))(*[float(v) for v in x])
    npu = np.zeros(18, dtype=np.longdouble)
    oracle.lib().orc_angles_to_u_ldout(arr(tex_ang), npu.ctypes.data_as(C.c_void_p))
    sm = [row[desc.idx_sm[q]] if desc.idx_sm[q] >= 0 else desc.sm_fixed[q] for q in range(4)]
    mass = [row[desc.idx_mass[q]] if desc.idx_mass[q] >= 0 else desc.mass_fixed[q] for q in range(2)]
    sc2 = math.pow(10., row[desc.idx_scale] if desc.idx_scale >= 0 else desc.scale_fixed)
    centres = np.sqrt(edges[:-1] * edges[1:])
    worst = max(L.x87t_bin_residual(arr(sm), arr(tex_ang), mass[0], mass[1], sc2, float(e), dim, None, npu.ctypes.data_as(C.c_void_p)) for e in centres)
    return not (worst < 1e-7)


def _random_paramset(rng):
    """A paramset with a random subset of {4 mixing params} x {2 source angles} sampled, in random column order,
    plus random nuisance columns."""
    sm = [("s_12_2", 0.307, [0., 1.], 0.013), ("c_13_4", 0.9565, [0., 1.], 0.00147), ("s_23_2", 0.538, [0., 1.], 0.069),
          ("dcp", 4.08, [0., 2 * np.pi], 2.0)]
    params = []
    mode = "SM_GAUSS" if rng.random() < 0.75 else "PRIOR_ONLY"
    if rng.random() < 0.7:                                            # all four mixing params sampled (else NuFIT values)
        for name, val, rg, std in sm:
            kind = rng.choice([None, PriorsCateg.LIMITEDGAUSS, PriorsCateg.GAUSSIAN], p=[0.3, 0.5, 0.2])
            kw = {} if kind is None else {"prior": kind}
            params.append(Param(name=name, value=val, ranges=rg, std=std * (10 if kind is PriorsCateg.GAUSSIAN else 1),
                                tag=ParamTag.SM_ANGLES, **kw))
    if rng.random() < 0.7:
        params.append(Param(name="source_angle1", value=0.5, ranges=[0., 1.], tag=ParamTag.SRCANGLES))
        params.append(Param(name="source_angle2", value=0.0, ranges=[-1., 1.], tag=ParamTag.SRCANGLES))
    for i in range(int(rng.integers(0, 17 - len(params) - 1 + 1))):
        if len(params) >= 16 or rng.random() < 0.35:
            break
        kind = rng.choice([None, PriorsCateg.LIMITEDGAUSS, PriorsCateg.GAUSSIAN])
        kw = {} if kind is None else {"prior": kind}
        params.append(Param(name="nuis%d" % i, value=float(rng.uniform(-1, 1)), ranges=[-2., 2.], std=float(rng.uniform(0.2, 1.5)),
                            tag=ParamTag.NUISANCE, **kw))
    if not params:
        params.append(Param(name="nuis0", value=0.1, ranges=[-2., 2.], std=0.7, tag=ParamTag.NUISANCE, prior=PriorsCateg.GAUSSIAN))
    order = rng.permutation(len(params))
    # the four SM_ANGLES keep their relative order (the notebook reads them by tag in declaration order)
    sm_pos = sorted(i for i in range(len(params)) if params[order[i]].tag is ParamTag.SM_ANGLES)
    sm_items = [p for p in params if p.tag is ParamTag.SM_ANGLES]
    src_pos = sorted(i for i in range(len(params)) if params[order[i]].tag is ParamTag.SRCANGLES)
    src_items = [p for p in params if p.tag is ParamTag.SRCANGLES]
    out = [params[j] for j in order]
    for pos, item in zip(sm_pos, sm_items):
        out[pos] = item
    for pos, item in zip(src_pos, src_items):
        out[pos] = item
    return ParamSet(out), mode


def _theta(ps, n, rng):
    box = np.array(ps.ranges, dtype=float)
    th = rng.uniform(box[:, 0], box[:, 1], size=(n, len(ps)))
    out = rng.random(n) < 0.1                                        # a tenth of the walkers leave the box in one column
    cols = rng.integers(0, len(ps), n)
    th[out, cols[out]] = box[cols[out], 1] + 0.3
    edge = rng.random(n) < 0.02                                      # and some sit exactly on a boundary (closed box)
    th[edge, cols[edge]] = box[cols[edge], rng.integers(0, 2, edge.sum())]
    wild = rng.random(n) < 0.01                                      # NaN / +-inf coordinates: -inf like the reference
    th[wild, cols[wild]] = rng.choice([np.nan, np.inf, -np.inf], wild.sum())
    return th


import os  # noqa: E402


@pytest.mark.parametrize("seed", range(int(os.environ.get("GF_FUZZ_SEEDS", "24"))))      # more seeds: a longer hunt
def test_random_posterior_structures(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    ps, mode = _random_paramset(rng)
    nd = len(ps)
    kw = dict(source_ratio=tuple(rng.dirichlet((1, 1, 1))))
    if mode == "SM_GAUSS":
        kw.update(bestfit_fr=tuple(rng.dirichlet((4, 3, 3))), smearing=float(rng.choice([0.02, 0.1, 0.5])))
    om = oracle.make_model(ps, mode, **kw)
    n = int(rng.choice([1, 37, 64, 100, 1000, 2048, 2049, 64 * 41 + 17, 64 * 64]))
    th = _theta(ps, n, rng)
    ref, rfr, rst = oracle.lnprob_batch(om, th, want_fr=True, want_status=True)
    with Model(compile_model(ps, mode, **kw)) as m:
        lp, fr, st = m.lnprob(th, want_fr=True)
        lp2 = m.lnprob(th, want_status=False)                        # the no-fr / no-status instantiation
        # device entry point, both layouts
        d_out = m.alloc(8 * n)
        res = {}
        for layout, arr in ((GF_LAYOUT_AOS, th), (GF_LAYOUT_SOA, np.ascontiguousarray(th.T))):
            d_th = m.alloc(arr.nbytes).upload(arr)
            m.lnprob_device(d_th.ptr, n, d_out.ptr, None, None, layout=layout)
            m.sync()
            res[layout] = d_out.download((n,))
            d_th.free()
    assert np.array_equal(st, rst), (seed, mode, nd, n)
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(lp), fin)
    if fin.any():
        assert rel_err(lp[fin], ref[fin]) <= REL, (seed, mode, nd, n)
    assert np.array_equal(lp, lp2, equal_nan=True)
    for layout in res:
        # the generic (SoA / ragged) kernel and the fast kernel run the same arithmetic in a different order of
        # loads only: identical results
        assert np.array_equal(res[layout], lp, equal_nan=True), (seed, layout)
    if mode == "SM_GAUSS":
        ok = st == _lib.GF_ST_OK
        assert np.abs(fr[ok] - rfr[ok]).max() <= ABS_FR if ok.any() else True


@pytest.mark.parametrize("seed", range(int(os.environ.get("GF_FUZZ_SEEDS_BSM", "12"))))
def test_random_bsm_configurations(oracle, seed):
    """The BSM kernel on random (operator dimension, texture, source, binning, 7- or 12-column paramset, scale
    window) configurations.  Bar as in test_gpu_parity.test_bsm_random_vs_oracle: 1e-10 plus the oracle's own
    80-bit unitarity defect per walker; the status verdict is the oracle's outside half a decade around the threshold."""
    from golemflavor_amd import configs as Cf
    from golemflavor_amd.enums import Texture
    rng = np.random.default_rng(7000 + seed)
    dim = int(rng.integers(3, 9))
    tex = [Texture.OEU, Texture.OET, Texture.OUT][int(rng.integers(0, 3))]
    src = rng.dirichlet((1, 1, 1)) if rng.random() < 0.5 else np.eye(3)[int(rng.integers(0, 3))]
    nbins = int(rng.choice([1, 2, 5, 20, 33, 64]))
    lo_e, hi_e = 10 ** rng.uniform(4, 5), 10 ** rng.uniform(6, 7.5)
    edges = np.logspace(np.log10(lo_e), np.log10(hi_e), nbins + 1)
    twelve = rng.random() < 0.5
    ps = Cf.fr_paramsets(dim, (0.4444, 0.0))[1] if twelve else Cf.texture_paramset(dim)
    bf = tuple(rng.dirichlet((3, 3, 3)))
    kw = dict(texture=tex, dimension=dim, binning=edges, source_ratio=src, bestfit_fr=bf, smearing=float(rng.choice([0.02, 0.2])))
    okw = dict(kw, texture=tex.name)
    om = oracle.make_model(ps, "BSM_GAUSS", **okw)
    n = int(rng.choice([64, 700, 3000, 9000]))
    box = np.array(ps.seeds, dtype=float)
    th = rng.uniform(box[:, 0], box[:, 1], size=(n, len(ps)))
    lo, hi = Cf.SCALE_BOUNDARIES[dim]
    th[:, -1] = rng.uniform(lo, lo + rng.uniform(0.3, 1.0) * (hi - lo), n)
    wild = rng.random(n) < 0.01                                      # NaN / +-inf coordinates: out of prior, -inf
    th[wild, rng.integers(0, len(ps), wild.sum())] = rng.choice([np.nan, np.inf, -np.inf], wild.sum())
    ref, ref_fr, ref_st = oracle.lnprob_batch(om, th, want_fr=True, want_status=True)
    with Model(compile_model(ps, "BSM_GAUSS", **kw)) as m:
        lp, fr, st = m.lnprob(th, want_fr=True)
        pfr, pst = m.propagate(th)
        # device entry point from an SoA buffer (the batch sizes here cover 16, 4 and 1 lanes per walker)
        soa = np.ascontiguousarray(th.T)
        d_th, d_out = m.alloc(soa.nbytes).upload(soa), m.alloc(8 * n)
        m.lnprob_device(d_th.ptr, n, d_out.ptr, None, None, layout=GF_LAYOUT_SOA)
        m.sync()
        lp_soa = d_out.download((n,))
    same = st != _lib.GF_ST_NON_UNITARY                        # (without a status array a flagged walker keeps its value)
    assert np.array_equal(lp_soa[same], lp[same], equal_nan=True), (seed, n)
    assert np.all(st[wild] == _lib.GF_ST_OUT_OF_PRIOR) and np.all(ref_st[wild] == 1) and np.all(np.isneginf(lp[wild]))
    r80 = oracle.unitarity_residual_batch(om, th)
    inbox = (st != _lib.GF_ST_OUT_OF_PRIOR) & (ref_st != 1)
    # the unitarity verdict: the device replays the reference's operations in emulated x87 arithmetic for every (walker,
    # bin) whose fp64 estimate is not clear-cut, so it is the oracle's verdict outside half a decade around the threshold
    # -- for every operator dimension (the fp64 estimate alone over-flagged dimensions 7-8 at the top of their range)
    clear = ((r80 < 10 ** -7.25) | (r80 > 10 ** -6.75)) & inbox
    flagged, ref_flagged = st == _lib.GF_ST_NON_UNITARY, ref_st == 2
    differ = np.flatnonzero(clear & (flagged != ref_flagged))
    if differ.size:
        # The oracle is the HOST's long double arithmetic, x87 microcode included (acosl is fpatan), and the microcode of two CPU
        # vendors differs in last bits: for a rare walker that sits on a branch point of the cubic that bit decides the verdict
        # (seed 254, walker 1991: residual 1.9e-9 on an Intel Xeon, 1.8e-6 on an AMD EPYC 9575F -- profiles/r04/fuzz_extended.txt).
        # Arbitrate with the host build of the device's own chain (gf_x87.hpp: correctly rounded functions, the same on every
        # host; pinned to the x87 unit by tests/test_x87_emulation.py): the device must agree with THAT, and such walkers stay rare.
        assert differ.size <= 2, (seed, dim, tex, nbins, differ)
        desc = compile_model(ps, "BSM_GAUSS", **kw)
        for i in differ:
            assert _emulated_host_verdict(oracle, desc, th[i], edges, dim, tex) == bool(flagged[i]), (seed, dim, tex, nbins, int(i), float(r80[i]))
    assert np.mean(flagged[inbox] == ref_flagged[inbox]) >= 0.998, (seed, dim, tex, nbins)
    good = (ref_st == 0) & (st == 0)
    if good.any():
        tol = ABS_FR + 10.0 * r80
        err = np.abs(fr - ref_fr).max(axis=1)
        over = np.flatnonzero(good & (err > tol))
        if over.size:
            # the long-double closed form is itself off by more than its unitarity defect suggests on a few walkers
            # (hierarchical spectra): arbitrate with 60-digit arithmetic -- the kernel must be the accurate one
            from exact_mp import exact_flux_avg
            pick = over[np.argsort(err[over])[::-1][:6]]
            exact = exact_flux_avg(th[pick], tex.name, dim, src, edges)
            assert np.abs(fr[pick] - exact).max() <= 1e-11, (seed, dim, tex, nbins, np.abs(fr[pick] - exact).max())
            assert np.abs(ref_fr[pick] - exact).max() > np.abs(fr[pick] - exact).max()
            assert over.size <= 0.02 * good.sum(), (seed, over.size, good.sum())
        clean = good & (r80 < 1e-13) & np.isfinite(ref)
        if clean.any():
            assert rel_err(lp[clean], ref[clean]) <= REL, (seed, dim, tex, nbins)
        # propagate (no priors, no likelihood) gives the same composition as the lnprob kernel, bit for bit
        assert np.array_equal(pfr[good], fr[good])


@pytest.mark.parametrize("seed", range(int(os.environ.get("GF_FUZZ_SEEDS_SAMPLER", "10"))))
def test_random_sampler_shapes_are_launch_shape_independent(monkeypatch, seed):
    """Device sampler on random (posterior structure, walkers, chains, steps, thinning): the one-workgroup-per-
    ensemble launch shape and the per-half-step grid launch shape give bitwise the same chains, acceptance counts
    and final state; stored lnprob is the batch kernel's evaluation of the stored position."""
    from golemflavor_amd import mcmc as mcmc_utils
    rng = np.random.default_rng(9000 + seed)
    ps, mode = _random_paramset(rng)
    nd = len(ps)
    kw = dict(source_ratio=tuple(rng.dirichlet((1, 1, 1))))
    if mode == "SM_GAUSS":
        kw.update(bestfit_fr=tuple(rng.dirichlet((4, 3, 3))), smearing=float(rng.choice([0.05, 0.3])))
    nwalkers = 2 * int(rng.integers(nd, 4 * nd + 40))
    nchains = int(rng.choice([1, 2, 5]))
    nsteps, thin = int(rng.integers(3, 50)), int(rng.choice([1, 1, 2, 5]))
    box = np.array(ps.ranges, dtype=float)
    p0 = rng.uniform(box[:, 0], box[:, 1], size=(nchains, nwalkers, nd))
    out = {}
    chain_seed = int(rng.integers(1, 2 ** 62))
    with Model(compile_model(ps, mode, **kw)) as m:
        for shape in ("1", "0"):
            monkeypatch.setenv("GF_SAMPLER_PERSIST", shape)
            s = mcmc_utils.DeviceEnsembleSampler(nwalkers, nd, m, nchains=nchains, seed=chain_seed)
            s.run_mcmc(p0 if nchains > 1 else p0[0], nsteps, thin=thin)
            out[shape] = (s.chain, s.lnprobability, s.acceptance_fraction, s.state[0], s.state[1])
            s.close()
        monkeypatch.delenv("GF_SAMPLER_PERSIST")
        for x, y in zip(out["1"], out["0"]):
            assert np.array_equal(x, y, equal_nan=True), (seed, mode, nd, nwalkers, nchains, nsteps, thin)
        ch, lp = out["1"][0], out["1"][1]
        again = m.lnprob(ch.reshape(-1, nd), want_status=False).reshape(lp.shape)
        assert np.array_equal(again, lp, equal_nan=True)


@pytest.mark.parametrize("seed", range(int(os.environ.get("GF_FUZZ_SEEDS_MULTI", "6"))))
def test_random_stacked_posteriors_follow_the_reference_stretch_move(oracle, seed):
    """One ensemble per posterior in one sampler (random posterior structure, 2-5 posteriors differing in best fit,
    smearing and source, random walkers / steps) against the numpy restatement of the stretch move that evaluates
    every chain with the ORACLE on the same Philox stream."""
    from golemflavor_amd import mcmc as mcmc_utils
    from test_gpu_sampler import _reference_stretch
    rng = np.random.default_rng(11000 + seed)
    ps, mode = _random_paramset(rng)
    nd = len(ps)
    nmodels = int(rng.integers(2, 6))
    kws = []
    for _ in range(nmodels):
        kw = dict(source_ratio=tuple(rng.dirichlet((1, 1, 1))))
        if mode == "SM_GAUSS":
            kw.update(bestfit_fr=tuple(rng.dirichlet((4, 3, 3))), smearing=float(rng.choice([0.1, 0.3, 0.6])))
        kws.append(kw)
    models = [Model(compile_model(ps, mode, **kw)) for kw in kws]
    oms = [oracle.make_model(ps, mode, **kw) for kw in kws]
    nwalkers = 2 * int(rng.integers(nd, nd + 12))
    nsteps = int(rng.integers(4, 14))
    box = np.array(ps.ranges, dtype=float)
    p0 = rng.uniform(box[:, 0], box[:, 1], size=(nmodels, nwalkers, nd))
    chain_seed = int(rng.integers(1, 2 ** 62))
    s = mcmc_utils.DeviceEnsembleSampler(nwalkers, nd, models, seed=chain_seed)
    s.run_mcmc(p0, nsteps)
    ref_chain, ref_lnp, ref_acc = _reference_stretch(oracle, oms, p0, nsteps, chain_seed)
    got = s.chain.transpose(0, 2, 1, 3)
    assert np.abs(got - ref_chain).max() < 1e-11, (seed, mode, nd, nmodels, nwalkers, nsteps)
    assert np.array_equal(np.round(s.acceptance_fraction * nsteps).astype(int), ref_acc)
    s.close()
    for m in models:
        m.close()
