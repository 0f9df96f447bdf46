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
