"""Which walkers does a cheap unitarity tier settle differently from the emulated-x87 chain?  Statuses as shipped against
the residual of every (walker, bin) pair from the arbitration kernel's own chain (gf_internal_uni_residuals).
Usage: python tools/tier_mismatch.py [n_per_case]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from common import BIN_EDGES, uniform_theta
from test_gpu_unitarity_r3 import _residuals, tier_cases
from golemflavor_amd import _lib, configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.model import Model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
from golemflavor_amd.enums import Texture
CASES = [(6, Texture.OEU, True), (5, Texture.OEU, True), (7, Texture.OEU, True), (8, Texture.OEU, True), (6, Texture.OUT, True),
         (7, Texture.OUT, True), (8, Texture.OUT, True), (4, Texture.OEU, True)]
tot = [0, 0, 0, 0]
for dim, tex, twelve in CASES:
    ps = Cf.fr_paramsets(dim, (0.4, 0.0))[1] if twelve else Cf.texture_paramset(dim)
    lo, hi = Cf.SCALE_BOUNDARIES[dim]
    rng = np.random.default_rng(1000 + dim)
    th = uniform_theta(ps, n, rng, seeds=True)
    th[:, -1] = rng.uniform(lo, hi, n)
    kw = dict(dimension=dim, binning=BIN_EDGES, source_ratio=(0., 1., 0.), bestfit_fr=(1 / 3,) * 3, smearing=0.02)
    with Model(compile_model(ps, "BSM_GAUSS", texture=tex, **kw)) as m:
        _, st = m.lnprob(th)
        d_th = m.alloc(th.nbytes).upload(th)
        res = np.empty((n, 20))
        for a in range(0, n, 100000):
            b = min(n, a + 100000)
            res[a:b] = _residuals(m, d_th, n, np.repeat(np.arange(a, b), 20), np.tile(np.arange(20), b - a), 1).reshape(b - a, 20)
    want = ~(res < 1e-7).all(axis=1)
    got = st == _lib.GF_ST_NON_UNITARY
    inbox = st != 1
    bad = np.flatnonzero((want != got) & inbox)
    worst = np.nanmax(res, axis=1)
    inband = (worst > 10 ** -7.25) & (worst < 10 ** -6.75)
    tot[0] += int(inbox.sum()); tot[1] += int((want & inbox).sum()); tot[2] += int(inband[bad].sum()); tot[3] += int((~inband[bad]).sum())
    print("d=%d %s twelve=%d: %d non-unitary by the chain, %d mismatches (%d with the worst residual inside the half-decade band, %d outside)" %
          (dim, tex.name, twelve, int((want & inbox).sum()), len(bad), int(inband[bad].sum()), int((~inband[bad]).sum())), flush=True)
    for i in bad[:6]:
        r = res[i]
        print("   walker %d: status %d, chain says %s; logLam %.6f; worst bin %d residual %.4e; residuals by bin:" %
              (i, st[i], "NON-unitary" if want[i] else "unitary", th[i, -1], int(np.nanargmax(r)), np.nanmax(r)))
        print("     ", " ".join("%.2e" % v for v in r))
        print("      theta", repr(th[i].tolist()))
print("total: %d walkers in the box, %d non-unitary by the chain; mismatches inside the band %d, outside %d" % tuple(tot))
