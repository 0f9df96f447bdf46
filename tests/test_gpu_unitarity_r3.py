"""GPU, round 3: the arbitration kernel's three-lane distribution of the emulated-x87 chain against the serial chain
(gf_x87.hpp, the statement tests/test_x87_emulation.py checks against the CPU's x87 unit) -- bit for bit -- and against the
oracle's residual (= the reference's); the walker-centric queue (highest bin first, stop at the first failure) against the
verdict of every bin."""
import ctypes as C

import numpy as np
import pytest

from common import BIN_EDGES, uniform_theta
from golemflavor_amd import _lib
from golemflavor_amd import configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model

pytestmark = pytest.mark.gpu


def _residuals(m, d_th, n, walkers, bins, which):
    L = _lib.lib()
    L.gf_internal_uni_residuals.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                            C.c_void_p]
    w = np.ascontiguousarray(walkers, dtype=np.int64)
    b = np.ascontiguousarray(bins, dtype=np.int32)
    d_w, d_b, d_o = m.alloc(w.nbytes).upload(w), m.alloc(b.nbytes).upload(b), m.alloc(8 * len(w))
    _lib.check(L.gf_internal_uni_residuals(m._h, d_th.ptr, 0, n, d_w.ptr, d_b.ptr, len(w), which, d_o.ptr), "uni residuals")
    out = d_o.download((len(w),))
    for d in (d_w, d_b, d_o):
        d.free()
    return out


CASES = [  # dimension, texture, 12 columns (SM angles sampled), NP angles sampled
    (6, Texture.OEU, True, False), (6, Texture.OET, False, False), (3, Texture.OUT, False, False), (4, Texture.OEU, True, False),
    (8, Texture.OUT, False, False), (6, Texture.NONE, False, True), (5, Texture.NONE, True, True)]


@pytest.mark.parametrize("dim,tex,twelve,np_sampled", CASES)
def test_three_lane_chain_equals_the_serial_chain_bit_for_bit(dim, tex, twelve, np_sampled, oracle):
    """Every (walker, bin) residual of the three-lane distribution is the serial chain's to the last bit -- the same
    operations in the same order, only on other lanes -- over the whole scale range (residuals from 1e-19 to order one),
    with the SM matrix sampled or fixed, fixed textures and sampled NP angles."""
    from test_oracle_golden import _mm_paramset
    rng = np.random.default_rng(100 + dim)
    if np_sampled:
        ps = _mm_paramset(dim, twelve)
    else:
        ps = Cf.fr_paramsets(dim, (0.4, 0.0))[1] if twelve else Cf.texture_paramset(dim)
    lo, hi = Cf.SCALE_BOUNDARIES[dim]
    n = 3000
    if np_sampled:                                                  # the NP angles over their ranges (they have no seed box)
        box = np.array([p.seed if p.seed is not None else p.ranges for p in ps], dtype=float)
        th = rng.uniform(box[:, 0], box[:, 1], size=(n, len(ps)))
    else:
        th = uniform_theta(ps, n, rng, seeds=True)
    th[:, -1] = rng.uniform(lo, hi, n)
    kw = dict(dimension=dim, binning=BIN_EDGES, source_ratio=(0., 1., 0.), bestfit_fr=(1 / 3,) * 3, smearing=0.02)
    with Model(compile_model(ps, "BSM_GAUSS", texture=tex, **kw)) as m:
        d_th = m.alloc(th.nbytes).upload(th)
        walkers = np.repeat(np.arange(n), 20)
        bins = np.tile(np.arange(20), n)
        serial = _residuals(m, d_th, n, walkers, bins, 0)
        group = _residuals(m, d_th, n, walkers, bins, 1)
        nine = _residuals(m, d_th, n, walkers, bins, 100)          # the nine-lane distribution (the sampler's settle step)
    assert np.array_equal(serial.view(np.uint64), group.view(np.uint64)), \
        "%d of %d residuals differ" % (np.sum(serial.view(np.uint64) != group.view(np.uint64)), len(serial))
    assert np.array_equal(serial.view(np.uint64), nine.view(np.uint64)), \
        "nine lanes: %d of %d residuals differ" % (np.sum(serial.view(np.uint64) != nine.view(np.uint64)), len(serial))
    assert np.isfinite(serial).mean() > 0.99
    # and the chain itself against the oracle (long double: the reference's arithmetic): worst bin per walker, verdict equal
    # outside half a decade around the threshold
    om = oracle.make_model(ps, "BSM_GAUSS", texture=tex.name, **kw)
    want = oracle.unitarity_residual_batch(om, th)
    got = group.reshape(n, 20).max(axis=1)
    clear = (want < 10 ** -7.25) | (want > 10 ** -6.75)
    assert np.array_equal((got >= 1e-7)[clear], (want >= 1e-7)[clear])
    # the residual is amplified rounding noise: where the emulated asin / acos / sin / cos differ from libm's in a last bit
    # (faithful, not correctly rounded) it changes by a factor of order one -- never by decades
    big = want > 1e-12
    if big.any():                                                   # (random NP angles keep every residual below 1e-12)
        off = np.abs(np.log10(got[big] / want[big]))
        assert np.mean(off < 0.5) > 0.99 and off.max() < 1.5, (np.mean(off < 0.5), off.max())
    near = want > 1e-9
    if near.any():
        assert np.abs(np.log10(got[near] / want[near])).max() < 0.6


def test_walker_centric_verdict_equals_all_bins_verdict():
    """The queue hands the kernel walkers, whose bins it takes from the top down until one fails.  The status must be what
    evaluating EVERY bin gives: a walker is non-unitary iff any bin's residual reaches 1e-7.  Every walker is sent to
    arbitration here (tier 1 off, band of 12 decades: needs GF_DIAGNOSTICS), in its own process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import numpy as np
from test_gpu_unitarity_r3 import _residuals
from common import BIN_EDGES, uniform_theta
from golemflavor_amd import _lib, configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model
bad = 0
for dim, tex, twelve in ((6, Texture.OEU, True), (6, Texture.OUT, False), (3, Texture.OET, False)):
    ps = Cf.fr_paramsets(dim, (0.4, 0.0))[1] if twelve else Cf.texture_paramset(dim)
    lo, hi = Cf.SCALE_BOUNDARIES[dim]
    rng = np.random.default_rng(7)
    n = 20000
    th = uniform_theta(ps, n, rng, seeds=True)
    th[:, -1] = rng.uniform(lo, hi, n)
    kw = dict(dimension=dim, binning=BIN_EDGES, source_ratio=(0., 1., 0.), bestfit_fr=(1 / 3,) * 3, smearing=0.02)
    with Model(compile_model(ps, "BSM_GAUSS", texture=tex, **kw)) as m:
        lp, st = m.lnprob(th)
        d_th = m.alloc(th.nbytes).upload(th)
        res = _residuals(m, d_th, n, np.repeat(np.arange(n), 20), np.tile(np.arange(20), n), 1).reshape(n, 20)
    want = ~(res < 1e-7).all(axis=1)
    got = st == _lib.GF_ST_NON_UNITARY
    bad += int(np.sum(want != got))
    print(dim, tex.name, "non-unitary", int(want.sum()), "mismatches", int(np.sum(want != got)), "nan lnprob", int(np.isnan(lp).sum()))
    assert np.array_equal(np.isnan(lp), got)
print("BAD", bad, "|" + _lib.diagnostic_overrides())
''' % (root, root)
    env = dict(os.environ, GF_DIAGNOSTICS="1", GF_UNI_BAND_DECADES="12", GF_UNI_NO_WEIGHT_GATE="1", PYTHONDONTWRITEBYTECODE="1")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    last = res.stdout.strip().splitlines()[-1]
    assert last.startswith("BAD 0 |") and "GF_UNI_BAND_DECADES=12" in last, res.stdout


_TIER_CHILD = r'''
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
from test_gpu_unitarity_r3 import tier_cases, tier_statuses
out = {}
for name, st in tier_statuses(tier_cases()):
    out[name] = st
np.savez(sys.argv[2], **out)
from golemflavor_amd import _lib
print("OVERRIDES", _lib.diagnostic_overrides())
'''


def tier_cases():
    # the failing region of the reference's assert is reached with OEU from d = 5 and OUT from d = 6; the others stay unitary
    return [(6, Texture.OEU, True), (5, Texture.OEU, True), (7, Texture.OEU, True), (8, Texture.OEU, True), (6, Texture.OUT, True),
            (8, Texture.OUT, True), (6, Texture.OET, False), (3, Texture.OUT, False), (4, Texture.OEU, False)]


def tier_statuses(cases, n=100000):
    """Statuses of n walkers per case, the scale over the whole SCALE_BOUNDARIES range (so that every tier sees walkers:
    SM-weight acquittals at low scales, the fp64 estimate's two cut-offs, the band in between)."""
    for dim, tex, twelve in cases:
        ps = Cf.fr_paramsets(dim, (0.4, 0.0))[1] if twelve else Cf.texture_paramset(dim)
        lo, hi = Cf.SCALE_BOUNDARIES[dim]
        rng = np.random.default_rng(1000 + dim)
        th = uniform_theta(ps, n, rng, seeds=True)
        th[:, -1] = rng.uniform(lo, hi, n)
        kw = dict(dimension=dim, binning=BIN_EDGES, source_ratio=(0., 1., 0.), bestfit_fr=(1 / 3,) * 3, smearing=0.02)
        with Model(compile_model(ps, "BSM_GAUSS", texture=tex, **kw)) as m:
            _, st = m.lnprob(th)
        yield "d%d_%s_%d" % (dim, tex.name, twelve), st


def test_the_cheap_tiers_never_overrule_the_x87_chain(tmp_path):
    """Tier 1 (SM-weight bound: acquits without evaluating anything) and tier 2 (fp64 estimate: acquits below the band,
    condemns above it) rest on margins measured on samples.  Here 900 000 walkers over the whole scale
    range of nine (dimension, texture) cases get their status twice: as shipped, and in a child process where every walker
    is sent through the emulated-x87 chain (GF_UNI_NO_WEIGHT_GATE + a 12-decade band, result-changing overrides that need
    GF_DIAGNOSTICS=1 and are echoed).  The two must be identical: no tier settles a walker the exact chain would settle the
    other way.  (Before the arbitration took every bin above the lowest undecided one -- uni_arbitration_mask -- 21 of
    2.4 M walkers differed here, all with one failing bin at 1.03e-7 ... 1.42e-7: profiles/r03/tier2_pairs.txt.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script, out = tmp_path / "child.py", tmp_path / "st.npz"
    script.write_text(_TIER_CHILD)
    env = dict(os.environ, GF_DIAGNOSTICS="1", GF_UNI_BAND_DECADES="12", GF_UNI_NO_WEIGHT_GATE="1", PYTHONDONTWRITEBYTECODE="1")
    res = subprocess.run([sys.executable, str(script), root, str(out)], capture_output=True, text=True, env=env, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    assert "GF_UNI_BAND_DECADES=12" in res.stdout and "GF_UNI_NO_WEIGHT_GATE=1" in res.stdout
    exact = np.load(out)
    assert "GF_UNI_" not in _lib.diagnostic_overrides()            # this process runs the tiers as shipped
    nbad = nnon = 0
    for name, st in tier_statuses(tier_cases()):
        nbad += int(np.sum(st != exact[name]))
        nnon += int(np.sum(exact[name] == _lib.GF_ST_NON_UNITARY))
        assert np.array_equal(st, exact[name]), (name, int(np.sum(st != exact[name])))
    assert nnon > 100000                                           # the failing region is well represented
