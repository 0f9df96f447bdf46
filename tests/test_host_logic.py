"""CPU: host-side logic -- data model, descriptor compiler, C-ABI loading, sampler, mcmc()."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from common import BIN_EDGES, notebook_sets
from golemflavor_amd import _lib
from golemflavor_amd import configs as Cf
from golemflavor_amd import fr as fr_utils
from golemflavor_amd import llh as llh_utils
from golemflavor_amd import mcmc as mcmc_utils
from golemflavor_amd.descriptor import compile_model, log_gauss_mass
from golemflavor_amd.enums import ParamTag, PriorsCateg, Texture
from golemflavor_amd.param import Param, ParamSet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- data model (param.py:24-214)
def test_param_defaults_and_nominal_value():
    p = Param(name="x", value=0.3, ranges=[0., 1.], std=0.1)
    assert p.prior is PriorsCateg.UNIFORM and p.tag is ParamTag.NONE and p.seed == (0., 1.)
    p.value = 0.9
    assert p.nominal_value == 0.3                      # frozen at construction
    q = Param(name="y", value=1, ranges=[0, 2], seed=[0.5, 1.5], prior=PriorsCateg.GAUSSIAN, tag=ParamTag.NUISANCE)
    assert q.seed == (0.5, 1.5) and q.tex == r"{\rm y}"
    with pytest.raises(AssertionError):
        Param(name="z", value=0, ranges=[0, 1], prior="gaussian")


def test_paramset_access_and_from_tag():
    _, ps = Cf.fr_paramsets(6, (0.4, 0.0))
    assert len(ps) == 12
    assert ps.names == ('s_12_2', 'c_13_4', 's_23_2', 'dcp', 'm21_2', 'm3x_2', 'convNorm', 'promptNorm', 'muonNorm',
                        'astroNorm', 'astroDeltaGamma', 'logLam')
    assert ps['logLam'].ranges == (-56, -30) and ps[3].name == 'dcp'
    assert ps.from_tag(ParamTag.NUISANCE).names == ('convNorm', 'promptNorm', 'muonNorm', 'astroNorm', 'astroDeltaGamma')
    assert ps.from_tag([ParamTag.SCALE, ParamTag.MMANGLES], values=True) == (-43.0,)
    assert ps.from_tag(ParamTag.SM_ANGLES, index=True) == (0, 1, 2, 3, 4, 5)
    assert len(ps.from_tag(ParamTag.SM_ANGLES, invert=True)) == 6
    with pytest.raises(ValueError):
        ParamSet(list(ps) + [Param(name='dcp', value=0, ranges=[0, 1])])
    assert ps.remove_params(ps.from_tag(ParamTag.NUISANCE)).names[-1] == 'logLam'
    assert ps.to_dict()['s_12_2'] == 0.307
    assert len(ps.extend(Param(name='new', value=0, ranges=[0, 1]))) == 13


# ---------------------------------------------------------------- host fr utilities vs golden
def test_host_fr_utilities_match_reference(golden):
    assert np.allclose(fr_utils.angles_to_fr((0.3, 0.4)), golden["ka_angles_to_fr_03_04"], rtol=0, atol=2e-16)
    U = fr_utils.NUFIT_U
    assert np.abs(np.asarray(U.real, float) - golden["ka_nufit_re"]).max() < 1e-15
    assert np.abs(np.asarray(U.imag, float) - golden["ka_nufit_im"]).max() < 1e-15
    for s, want in zip(golden["g3_src"], golden["g3_fr_nufit"]):
        assert np.abs(np.asarray(fr_utils.u_to_fr(s, U), float) - want).max() < 1e-15
    ang = fr_utils.fr_to_angles(fr_utils.u_to_fr((1, 0, 0), U))
    assert np.allclose([float(a) for a in ang], golden["g6_asimov_angles"], rtol=0, atol=1e-15)
    for f, want in zip(golden["g2_fr_in"], golden["g2_angles_back"]):
        assert np.allclose([float(a) for a in fr_utils.fr_to_angles(f)], want, rtol=0, atol=1e-14)
    asimov, _ = notebook_sets(golden)
    assert np.allclose(fr_utils.angles_to_fr(asimov.values), golden["g6_bestfit_fr"], rtol=0, atol=1e-15)


# ---------------------------------------------------------------- descriptor
def test_log_gauss_mass_matches_scipy_truncnorm(oracle):
    # scipy's own normalisation of the frozen truncnorm the reference builds (llh.py:25-29)
    from scipy.stats._continuous_distns import _log_gauss_mass
    for a, b in [(-23.6, 53.3), (-0.5, 2.0), (0.0, 8.3), (1.5, 4.0), (-6.0, -2.0), (-40.0, -30.0), (-1e-3, 1e-3)]:
        want = float(_log_gauss_mass(np.float64(a), np.float64(b)))
        assert log_gauss_mass(a, b) == pytest.approx(want, rel=1e-13, abs=1e-300)
        assert oracle.log_gauss_mass(a, b) == pytest.approx(want, rel=1e-13, abs=1e-300)


def test_compile_model_notebook(golden):
    asimov, ps = notebook_sets(golden)
    d = compile_model(ps, "SM_GAUSS", bestfit_fr=golden["g6_bestfit_fr"], smearing=0.02)
    assert d.ndim == 6 and d.mode == _lib.GF_MODE_SM_GAUSS
    assert list(d.idx_sm) == [0, 1, 2, 3] and list(d.idx_src) == [4, 5] and d.idx_scale == -1
    assert list(d.prior_kind)[:6] == [3, 3, 3, 1, 1, 1]
    assert d.loc[1] == (1 - 0.02206) ** 2 and d.sigma[2] == 0.069 and d.hi[3] == 2 * np.pi
    assert d.offset == -320.0 and C.sizeof(d) % 8 == 0


def test_compile_model_bsm_rules():
    _, ps = Cf.fr_paramsets(6, (0.4, 0.0))
    d = compile_model(ps, "BSM_GAUSS", bestfit_fr=(1 / 3,) * 3, smearing=0.02, texture=Texture.OET, dimension=6,
                      binning=BIN_EDGES, source_ratio=(0, 1, 0))
    assert d.nbins == 20 and d.texture == 2 and d.idx_scale == 11 and d.idx_gamma == 10
    assert list(d.idx_sm) == [0, 1, 2, 3] and list(d.idx_mass) == [4, 5]
    # fr.py:425-435: without the two mass params the SM part falls back to NuFIT defaults
    ps7 = ParamSet([p for p in Cf.texture_paramset(3) if p.name not in ("m21_2", "m3x_2")])
    d = compile_model(ps7, "BSM_GAUSS", bestfit_fr=(1 / 3,) * 3, smearing=0.02, texture=Texture.OUT, dimension=3,
                      binning=BIN_EDGES)
    assert list(d.idx_sm) == [-1] * 4 and list(d.idx_mass) == [-1, -1]
    assert d.sm_fixed[0] == 0.307 and d.mass_fixed[1] == 2.515e-21
    with pytest.raises(ValueError):
        compile_model(ps, "BSM_GAUSS", bestfit_fr=(1 / 3,) * 3, smearing=0.02, texture=Texture.NONE, dimension=6,
                      binning=BIN_EDGES)                               # texture NONE needs MMANGLES
    with pytest.raises(ValueError):
        compile_model(ps, "SM_GAUSS")                                  # Gaussian modes need bestfit + smearing
    with pytest.raises(ValueError):
        compile_model(ParamSet([Param(name="p%d" % i, value=0, ranges=[0, 1]) for i in range(17)]), "PRIOR_ONLY")


# ---------------------------------------------------------------- C ABI (no compute calls: no GPU here)
def test_cabi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "golemflavor_hip.h")).read()
    declared = set(re.findall(r"^(?:int|void|size_t|int64_t|const char\*)\s+(gf_[a-z0-9_]+)\s*\(", hdr, re.M))
    assert len(declared) >= 39
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), "libgolemhip.so does not export %s" % name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert L.gf_abi_version() == _lib.GF_ABI_VERSION
    assert L.gf_strerror(_lib.GF_ERR_NO_DEVICE).decode().startswith("no gfx950")


def test_descriptor_struct_layout_matches_header():
    # the C side static-asserts nothing about Python; cross-check the size from the header's field list
    n_i32 = 6 + 4 + 2 + 2 + 1 + 4 + 1 + 2 + 16
    n_f64 = 16 * 5 + 4 + 2 + 3 + 1 + 4 + 1 + 3 + 1 + 1 + 1 + 65
    assert C.sizeof(_lib.GfModelDesc) == 4 * n_i32 + 8 * n_f64
    assert _lib.lib().gf_sizeof_model_desc() == C.sizeof(_lib.GfModelDesc)


def test_model_create_validates_before_touching_the_device(golden):
    L = _lib.lib()
    _, ps = notebook_sets(golden)
    d = compile_model(ps, "SM_GAUSS", bestfit_fr=golden["g6_bestfit_fr"], smearing=0.02)
    h = C.c_void_p()
    d.ndim = 0
    assert L.gf_model_create(C.byref(d), 0, C.byref(h)) == _lib.GF_ERR_INVALID_ARG
    d.ndim = 6
    d.abi_version = 99
    assert L.gf_model_create(C.byref(d), 0, C.byref(h)) == _lib.GF_ERR_INVALID_ARG
    d.abi_version = _lib.GF_ABI_VERSION
    d.sigma[0] = -1.0
    assert L.gf_model_create(C.byref(d), 0, C.byref(h)) == _lib.GF_ERR_INVALID_ARG
    d.sigma[0] = 0.013
    # CP phase range beyond the kernels' argument reduction: refused, not silently NaN
    d.hi[3] = 4.0e6
    assert L.gf_model_create(C.byref(d), 0, C.byref(h)) == _lib.GF_ERR_UNSUPPORTED
    assert b"phase" in L.gf_last_hip_error()
    d.hi[3] = 2 * np.pi
    if _lib.device_count() == 0:
        # no GPU in this container: the product path fails loudly, there is no CPU fallback
        rc = L.gf_model_create(C.byref(d), 0, C.byref(h))
        assert rc == _lib.GF_ERR_NO_DEVICE
        from golemflavor_amd.model import Model
        with pytest.raises(_lib.GolemHipError):
            Model(d)


# ---------------------------------------------------------------- sampler + mcmc()
class _GaussLnProb:
    """Vectorised toy posterior with known moments (test double for golemflavor_amd.llh.LnProb)."""
    vectorized = True

    def __init__(self, mu, sig):
        self.mu, self.sig, self.calls, self.shapes = np.asarray(mu), np.asarray(sig), 0, set()

    def __call__(self, th):
        th = np.atleast_2d(th)
        self.calls += 1
        self.shapes.add(th.shape)
        out = -0.5 * np.sum(((th - self.mu) / self.sig) ** 2, axis=1)
        out[np.any(np.abs(th) > 50, axis=1)] = -np.inf
        return out


def test_stretch_move_samples_the_target_and_batches_half_ensembles():
    np.random.seed(26)
    f = _GaussLnProb([1.0, -2.0, 0.5], [0.5, 2.0, 1.0])
    s = mcmc_utils.EnsembleSampler(64, 3, f, seed=11)
    p0 = np.random.normal(0, 1, size=(64, 3))
    pos = s.run_mcmc(p0, 300)[0]
    s.reset()
    s.run_mcmc(pos, 2500)
    assert f.shapes == {(64, 3), (32, 3)}                     # one call per half-ensemble
    assert f.calls == 2 + 2 * 2800
    chain = s.chain
    assert chain.shape == (64, 2500, 3) and s.lnprobability.shape == (64, 2500)
    flat = s.flatchain
    assert np.allclose(flat.mean(axis=0), [1.0, -2.0, 0.5], atol=0.1)
    assert np.allclose(flat.std(axis=0), [0.5, 2.0, 1.0], rtol=0.1)
    acc = s.acceptance_fraction
    assert acc.shape == (64,) and 0.4 < acc.mean() < 0.8       # 3-dim Gaussian: ~0.6
    tau = s.acor
    assert tau.shape == (3,) and np.all((tau > 3) & (tau < 50))


def test_sampler_accepts_plain_per_walker_callables_like_emcee():
    calls = []

    def ln_prob(theta):
        calls.append(np.shape(theta))
        return -0.5 * float(np.sum(np.square(theta)))

    s = mcmc_utils.EnsembleSampler(8, 2, ln_prob, seed=3)
    s.run_mcmc(np.random.default_rng(0).normal(size=(8, 2)), 5)
    assert set(calls) == {(2,)} and len(calls) == 8 + 5 * 8
    with pytest.raises(AssertionError):
        mcmc_utils.EnsembleSampler(7, 2, ln_prob)              # odd number of walkers
    with pytest.raises(AssertionError):
        mcmc_utils.EnsembleSampler(2, 2, ln_prob)              # fewer than 2*dim walkers


def test_minus_inf_proposals_are_never_accepted():
    f = _GaussLnProb([0.0], [100.0])                           # wide: proposals beyond |50| happen
    s = mcmc_utils.EnsembleSampler(10, 1, f, seed=5)
    s.run_mcmc(np.linspace(-45, 45, 10)[:, None], 400)
    assert np.all(np.abs(s.flatchain) <= 50) and np.all(np.isfinite(s.lnprobability))


def test_mcmc_driver_signature_prints_and_return(capsys, golden):
    _, ps = notebook_sets(golden)
    np.random.seed(26)
    p0 = mcmc_utils.flat_seed(ps, nwalkers=20)
    seeds = np.array(ps.seeds)
    assert p0.shape == (20, 6) and np.all(p0 >= seeds[:, 0]) and np.all(p0 <= seeds[:, 1])
    np.random.seed(26)
    assert np.array_equal(p0, mcmc_utils.flat_seed(ps, nwalkers=20))     # reproducible from np.random.seed
    f = _GaussLnProb(np.array(ps.values, dtype=float), np.array([0.01, 0.001, 0.05, 1.0, 0.3, 0.3]))
    samples = mcmc_utils.mcmc(p0=p0, ln_prob=f, ndim=6, nwalkers=20, burnin=20, nsteps=50, threads=4)
    assert samples.shape == (20 * 50, 6)
    out = capsys.readouterr().out
    for line in ("Running burn-in", "Finished burn-in", "Running", "Finished", "acceptance fraction",
                 "sum of acceptance fraction", "np.unique(samples[:,0]).shape", "WARNING : NEED TO RUN MORE SAMPLES"):
        assert line in out
    g = mcmc_utils.gaussian_seed(Cf.texture_paramset(6), 12)
    assert g.shape == (12, 7)


def test_save_chains(tmp_path):
    of = mcmc_utils.save_chains(np.arange(6.).reshape(3, 2), str(tmp_path / "sub" / "chains_x"))
    assert of.endswith("chains_x.npy") and np.array_equal(np.load(of), np.arange(6.).reshape(3, 2))


def test_integrated_time_on_ar1():
    rng = np.random.default_rng(1)
    rho, n = 0.9, 200000
    x = np.empty(n)
    x[0] = 0
    e = rng.normal(size=n)
    for i in range(1, n):
        x[i] = rho * x[i - 1] + e[i]
    tau = mcmc_utils.integrated_time(x[:, None])[0]
    assert tau == pytest.approx((1 + rho) / (1 - rho), rel=0.1)       # 19
    with pytest.raises(mcmc_utils.AutocorrError):
        mcmc_utils.integrated_time(x[:200, None])


def test_chain_identifier_matches_reference_naming():
    """misc.py:34-51 gen_identifier / solve_ratio."""
    import argparse
    from golemflavor_amd.enums import DataType
    a = argparse.Namespace(dimension=6, source_ratio=np.array([1., 2., 0.]) / 3, injected_ratio=np.array([1., 1., 1.]) / 3,
                           texture=Texture.OET, data=DataType.ASIMOV)
    assert mcmc_utils.chain_identifier(a) == "_DIM6_sfr_1_2_0_mfr_1_1_1_OET"
    a.data = DataType.REAL
    a.texture = Texture.NONE
    a.source_ratio = np.array([0.3, 0.7, 0.0])
    assert mcmc_utils.chain_identifier(a) == "_DIM6_sfr_0.30_0.70_0.00"
    assert mcmc_utils.solve_ratio([0., 1., 0.]) == "0_1_0" and mcmc_utils.solve_ratio([2., 4., 0.]) == "1_2_0"


def test_host_multi_gaussian_matches_golden(golden):
    """llh.multi_gaussian (host closed form, llh.py:32-54) on the G5 vectors: near the mode, far tail,
    subnormal band, -inf wall."""
    got = llh_utils.multi_gaussian(golden["g5_fr"], golden["g5_bf"], 0.02)
    ref = golden["g5_llh"]
    assert np.array_equal(np.isinf(got), np.isinf(ref))
    fin = np.isfinite(ref)
    assert np.abs(got[fin] - ref[fin]).max() <= 1e-12 * np.abs(ref[fin]).max()
    assert np.isinf(ref).sum() > 10 and fin.sum() > 100


def test_header_is_plain_c99(tmp_path):
    """include/golemflavor_hip.h is consumed by C, not only by the C++ translation units that implement it:
    the C consumer under tests/cabi must compile warning-free as strict C99."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    obj = str(tmp_path / "smoke.o")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-D_GNU_SOURCE", "-I", os.path.join(root, "include"),
                        "-c", os.path.join(root, "tests", "cabi", "smoke.c"), "-o", obj], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_host_sampler_thinning_counts_every_iteration():
    """emcee-2 semantics (what golemflavor/mcmc.py:45 prints): `iterations` counts every step, the chain keeps every
    `thin`-th one, acceptance_fraction = naccepted / iterations -- also across several sample() calls."""
    from golemflavor_amd import mcmc

    def lnp(x):
        return -0.5 * float(np.sum(np.asarray(x) ** 2))

    s = mcmc.EnsembleSampler(8, 2, lnp, seed=3)
    p0 = np.random.default_rng(0).normal(size=(8, 2))
    pos = s.run_mcmc(p0, 30, thin=3)[0]
    assert s.iterations == 30 and s.chain.shape == (8, 10, 2) and s.lnprobability.shape == (8, 10)
    s.run_mcmc(pos, 10, thin=3)                          # stores steps 0, 3, 6, 9 of this call
    assert s.iterations == 40 and s.chain.shape == (8, 14, 2)
    assert np.all(s.acceptance_fraction <= 1.0) and np.all(s.naccepted <= 40)
    unthinned = mcmc.EnsembleSampler(8, 2, lnp, seed=3)
    unthinned.run_mcmc(p0, 30)
    assert np.array_equal(unthinned.chain[:, ::3], s.chain[:, :10])       # same random stream, every third step kept
    assert np.array_equal(unthinned.naccepted / 30, unthinned.acceptance_fraction)


# ---------------------------------------------------------------- round 3: no_bsm, mcmc_argparse, override record
def test_no_bsm_compiles_to_the_standard_propagation():
    """args.no_bsm (fr.py:437-438; the reference's branch cannot run): defined as u_to_fr(source_ratio, sm_u)
    with the flux average's own column rule -- mixing angles from theta only when all six oscillation parameters are scanned."""
    ps12 = Cf.fr_paramsets(6, (0.5, 0.0))[1]
    kw = dict(bestfit_fr=(1 / 3,) * 3, smearing=0.02, source_ratio=(0.0, 1.0, 0.0), texture=Texture.OET, dimension=6,
              binning=BIN_EDGES)
    d = compile_model(ps12, "BSM_GAUSS", no_bsm=True, **kw)
    assert d.mode == _lib.GF_MODE_SM_GAUSS and d.ndim == 12 and d.nbins == 0
    assert list(d.idx_sm) == [0, 1, 2, 3] and list(d.idx_src) == [-1, -1] and d.idx_src_x == -1
    assert list(d.source_ratio) == [0.0, 1.0, 0.0]
    # without the mass splittings among the columns the angles are NOT taken from theta (fr.py:433-435: NuFIT)
    ps4 = Cf.mcx_paramset()
    d4 = compile_model(ps4, "BSM_GAUSS", no_bsm=True, **kw)
    assert list(d4.idx_sm) == [-1] * 4 and d4.idx_src_x == -1 and list(d4.sm_fixed) == list(Cf.NUFIT_ANGLES)
    with pytest.raises(ValueError):
        compile_model(ps12, "SM_GAUSS", no_bsm=True, bestfit_fr=(1 / 3,) * 3, smearing=0.02)
    # the BSM descriptor itself is unchanged by the new keyword's default
    assert bytes(memoryview(compile_model(ps12, "BSM_GAUSS", **kw))) == bytes(memoryview(compile_model(ps12, "BSM_GAUSS", no_bsm=False, **kw)))


def test_mcmc_argparse_has_the_reference_flags_and_defaults():
    """golemflavor/mcmc.py:56-85."""
    import argparse
    from golemflavor_amd import mcmc as mcmc_utils
    from golemflavor_amd.enums import MCMCSeedType
    p = mcmc_utils.mcmc_argparse(argparse.ArgumentParser())
    a = p.parse_args([])
    assert (a.run_mcmc, a.burnin, a.nwalkers, a.nsteps, a.mcmc_seed_type, a.plot_angles, a.plot_elements) == \
        (True, 100, 60, 2000, MCMCSeedType.UNIFORM, False, False)
    a = p.parse_args("--run-mcmc False --burnin 200 --nwalkers 2048 --nsteps 1000 --mcmc-seed-type gaussian --plot-angles True".split())
    assert (a.run_mcmc, a.burnin, a.nwalkers, a.nsteps, a.mcmc_seed_type, a.plot_angles) == \
        (False, 200, 2048, 1000, MCMCSeedType.GAUSSIAN, True)
    with pytest.raises(SystemExit):
        p.parse_args(["--mcmc-seed-type", "cauchy"])


def test_result_changing_overrides_need_gf_diagnostics(tmp_path):
    """A stray GF_UNI_* variable must not change verdicts silently: without GF_DIAGNOSTICS=1 it is ignored (and listed as
    ignored); with it, it is honoured and listed.  Child processes: the record is per process."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from golemflavor_amd import _lib\n"
            "import ctypes as C\n"
            "L = _lib.lib()\n"
            "L.gf_internal_env.restype = C.c_char_p; L.gf_internal_env.argtypes = [C.c_char_p, C.c_int]\n"
            "v = L.gf_internal_env(b'GF_UNI_A_OK', 1); w = L.gf_internal_env(b'GF_SAMPLER_LPW', 0)\n"
            "print(repr(v), repr(w), '|' + _lib.diagnostic_overrides())\n" % ROOT)
    env = dict(os.environ, GF_UNI_A_OK="1e-9", GF_SAMPLER_LPW="4")
    env.pop("GF_DIAGNOSTICS", None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=60)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "None b'4' |GF_UNI_A_OK(ignored) GF_SAMPLER_LPW=4" and "ignored" in out.stderr
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, GF_DIAGNOSTICS="1"), timeout=60)
    assert out.stdout.strip() == "b'1e-9' b'4' |GF_UNI_A_OK=1e-9 GF_SAMPLER_LPW=4"


def test_scan_point_descriptors_are_the_compiled_ones():
    """A grid scan compiles one descriptor per (paramset, dimension) and patches source ratio / texture / fixed scale per
    point (scan._patched).  Every point's patched descriptor must be, byte for byte, what the reference-shaped entry points
    compile for that point on their own: llh.bsm_ln_prob's descriptor (C5), mc_texture.py's two models (C4)."""
    import argparse
    from golemflavor_amd import scan, fr as fr_utils
    from golemflavor_amd.enums import ParamTag
    for point in scan.sens_grid():
        dim, tex, source, scale = point
        ps, box, got = scan._SensPoint.descriptor(point, 0.02)
        asimov, ps_ref = Cf.fr_paramsets(dim, fr_utils.fr_to_angles((1, 1, 1)))
        bf = fr_utils.angles_to_fr(asimov.from_tag(ParamTag.BESTFIT, values=True))
        want = compile_model(ps_ref, "BSM_GAUSS", bestfit_fr=bf, smearing=0.02, source_ratio=np.array(source), texture=tex,
                             dimension=dim, binning=Cf.default_bin_edges(), no_bsm=False)
        assert bytes(got) == bytes(want), point
        assert np.array_equal(box, np.array(ps_ref.seeds, dtype=float))
    for tex in (Texture.OET, Texture.OEU):
        for point in scan.texture_grid(6):
            scale, source = point
            ps6, box, prior, post = scan._TexturePoint.descriptors(point, 6, tex)
            ref6 = Cf.ParamSet(list(Cf.texture_paramset(6))[:6])
            assert bytes(prior) == bytes(compile_model(ref6, "PRIOR_ONLY", flat_llh=1.0))
            want = compile_model(ref6, "BSM_GAUSS", texture=tex, dimension=6, binning=Cf.default_bin_edges(), source_ratio=source,
                                 scale_fixed=scale, bestfit_fr=(1 / 3,) * 3, smearing=0.02)
            assert bytes(post) == bytes(want), point
    with pytest.raises(ValueError):
        scan._patched(post, texture=Texture.NONE)
    with pytest.raises(KeyError):
        scan._patched(post, dimension=4)


def test_host_prepare_keeps_the_content_and_tolerates_a_concurrent_writer():
    """gf_host_prepare maps the pages of a result buffer from several threads with an atomic OR of zero per page: what the
    buffer holds stays, also while another thread is filling it (a scan maps its result array during the run that fills it)."""
    import threading
    L = _lib.lib()
    n = 64 << 20
    a = np.arange(n // 8, dtype=np.float64)
    assert L.gf_host_prepare(a.ctypes.data_as(C.c_void_p), a.nbytes) == _lib.GF_OK
    assert np.array_equal(a, np.arange(n // 8, dtype=np.float64))
    b = np.empty(n // 8)
    src = np.random.default_rng(0).random(n // 8)
    t = threading.Thread(target=lambda: L.gf_host_prepare(b.ctypes.data_as(C.c_void_p), b.nbytes))
    t.start()
    b[:] = src                                                     # races with the page touches
    t.join()
    assert np.array_equal(b, src)
    assert L.gf_host_prepare(None, 0) == _lib.GF_OK
    # ... and it really maps them (an elided read-modify-write would leave the shared zero page in place: "12.6 GB in 1 ms")
    def rss():
        with open("/proc/self/statm") as f:
            return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE")
    c = np.empty((256 << 20) // 8)
    before = rss()
    assert L.gf_host_prepare_n(c.ctypes.data_as(C.c_void_p), c.nbytes, 4) == _lib.GF_OK
    assert rss() - before > 200 << 20


def test_host_prepare_poke_path_maps_pages_and_keeps_the_content(tmp_path):
    """The fallback of gf_host_prepare where madvise(MADV_POPULATE_WRITE) is refused (kernels before 5.14): a locked
    read-modify-write of one byte per page.  Forced with GF_PREPARE_FORCE_POKE in a child process (the switch is read once):
    the pages must become resident and what the buffer held must stay."""
    import subprocess
    import sys
    code = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from golemflavor_amd import _lib
L = _lib.lib()
def rss():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE")
c = np.empty((192 << 20) // 8)
before = rss()
assert L.gf_host_prepare_n(c.ctypes.data_as(C.c_void_p), c.nbytes, 3) == _lib.GF_OK
grown = rss() - before
a = np.arange((32 << 20) // 8, dtype=np.float64)
assert L.gf_host_prepare(a.ctypes.data_as(C.c_void_p), a.nbytes) == _lib.GF_OK
print(grown, bool(np.array_equal(a, np.arange(a.size, dtype=np.float64))), "GF_PREPARE_FORCE_POKE" in _lib.diagnostic_overrides())
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GF_PREPARE_FORCE_POKE="1", PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code, root], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    grown, same, echoed = out.stdout.split()
    assert int(grown) > 150 << 20 and same == "True" and echoed == "True"


def test_result_arena_without_a_gpu_is_still_an_arena():
    """scan.ResultArena (ABI 5): registration fails where there is no device -- reported, not raised --, the block's pages are mapped
    instead, `result_array` serves from it what fits (views that ALIAS the arena) and fresh memory otherwise."""
    from golemflavor_amd import scan
    arena = scan.ResultArena(8 << 20)
    try:
        assert arena.nbytes == 8 << 20 and arena.seconds >= 0.0
        if not arena.registered:
            assert arena.register_error                                     # "GolemHipError: gf_host_register failed: ..." or no library
        old = scan.set_result_arena(arena)
        try:
            a = scan.result_array((4, 1000, 8))
            b = scan.result_array((2, 100))
            assert np.shares_memory(a, arena.array) and np.shares_memory(b, a)      # the next result overwrites the previous one
            big = scan.result_array((3 << 20,))                                  # 24 MB: does not fit
            assert not np.shares_memory(big, arena.array) and big.shape == (3 << 20,)
        finally:
            scan.set_result_arena(old)
        assert not np.shares_memory(scan.result_array((16,)), arena.array)
    finally:
        arena.close()
    assert arena.take((1,)) is None and not arena.registered
