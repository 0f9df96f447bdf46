"""Per (walker, bin) pair: how low can the fp64 estimate of tier 2 sit while the emulated-x87 chain's residual is at or above
the reference's threshold?  Tier 2 acquits a PAIR when its estimate is 2.7 decades below 1e-7, so this is the distribution
its lower band must cover -- pair level, not walker level (a walker's largest estimate may come from another bin).
The estimate of bin k alone comes from a one-bin model (binning = that bin's two edges) with GF_UNI_DUMP (fr[0] <- estimate);
the residuals are the arbitration kernel's own chain.  Usage: python tools/uni_estimate_tail.py [walkers per case]"""
import os, sys
os.environ["GF_DIAGNOSTICS"] = "1"; os.environ["GF_UNI_DUMP"] = "1"; os.environ["GF_UNI_NO_WEIGHT_GATE"] = "1"
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from common import BIN_EDGES, uniform_theta
from test_gpu_unitarity_r3 import _residuals
from golemflavor_amd import _lib, configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
cuts = (-2.0, -2.2, -2.45, -2.7, -2.95, -3.2, -3.7, -4.2)
tot_fail = tot_clear = 0
tot_counts_fail, tot_counts_clear = np.zeros(len(cuts), int), np.zeros(len(cuts), int)
for dim, tex in ((6, Texture.OEU), (5, Texture.OEU), (7, Texture.OEU), (8, Texture.OEU), (6, Texture.OUT), (7, Texture.OUT), (8, Texture.OET)):
    ps = Cf.fr_paramsets(dim, (0.4, 0.0))[1]
    lo, hi = Cf.SCALE_BOUNDARIES[dim]
    rng = np.random.default_rng(3000 + dim)
    th = uniform_theta(ps, n, rng, seeds=True)
    th[:, -1] = rng.uniform(lo, hi, n)
    kw = dict(dimension=dim, source_ratio=(0., 1., 0.), bestfit_fr=(1 / 3,) * 3, smearing=0.02)
    with Model(compile_model(ps, "BSM_GAUSS", texture=tex, binning=BIN_EDGES, **kw)) as m:
        d_th = m.alloc(th.nbytes).upload(th)
        res = np.empty((n, 20))
        for a in range(0, n, 100000):
            b = min(n, a + 100000)
            res[a:b] = _residuals(m, d_th, n, np.repeat(np.arange(a, b), 20), np.tile(np.arange(20), b - a), 1).reshape(b - a, 20)
        _, st0 = m.lnprob(th[:8])
    est = np.zeros((n, 20))
    for k in range(20):
        with Model(compile_model(ps, "BSM_GAUSS", texture=tex, binning=BIN_EDGES[k:k + 2], **kw)) as mk:
            for a in range(0, n, 50000):                              # small batches: the evaluation kernel runs tier 2 itself (and dumps)
                fr, st = mk.propagate(th[a:a + 50000])
                est[a:a + 50000, k] = np.where(st == 1, np.nan, fr[:, 0])
    fail = np.isfinite(est) & (res >= 1e-7)
    clear = np.isfinite(est) & (res >= 10 ** -6.75)
    marg = np.log10(np.maximum(est, 1e-300) / 1e-7)
    cf = np.array([(marg[fail] < c).sum() for c in cuts]); cc = np.array([(marg[clear] < c).sum() for c in cuts])
    tot_fail += fail.sum(); tot_clear += clear.sum(); tot_counts_fail += cf; tot_counts_clear += cc
    print("d=%d %-3s: %8d pairs at or above 1e-7 (%d at or above 10^-6.75); lowest estimate / 1e-7: 10^%.2f (10^%.2f); pairs with the estimate below 10^c x 1e-7, c = %s: %s (%s)" %
          (dim, tex.name, fail.sum(), clear.sum(), marg[fail].min() if fail.any() else np.nan, marg[clear].min() if clear.any() else np.nan,
           list(cuts), cf.tolist(), cc.tolist()), flush=True)
print("all: %d pairs at or above 1e-7, %d at or above 10^-6.75; below the cuts %s: %s (%s)" % (tot_fail, tot_clear, list(cuts), tot_counts_fail.tolist(), tot_counts_clear.tolist()))
