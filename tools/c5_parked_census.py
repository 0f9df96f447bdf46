"""How many walkers of the C5 scan (256 chains x 512 walkers) need the x87 arbitration at a typical step, and how many
undecided bins they bring: the positions after 150 steps are evaluated chain by chain with status, and the arbitration
queue of each call is dumped (gf_internal_uni_dump).  A half-step of the device sampler parks about half of that."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from golemflavor_amd import _lib, scan, mcmc as mcmc_utils
L = _lib.lib()
L.gf_internal_uni_dump.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint32)]
pts = scan.sens_grid()
jobs = [scan._SensPoint(p, g, nwalkers=512, device=0) for g, p in enumerate(pts)]
s = mcmc_utils.DeviceEnsembleSampler(512, 12, [j.f for j in jobs], seed=25, stream_ids=list(range(len(jobs))))
s.on_nonunitary = "-inf"
s.run_mcmc(np.stack([j.p0 for j in jobs]), 150, storechain=False)
pos, lnp = s.state
tot_w = tot_b = 0
per_chain = []
hist = np.zeros(21, int)
for g, j in enumerate(jobs):
    m = j.f.model
    lp, st = m.lnprob(pos[g], want_status=True)
    items = np.zeros((512, 2), dtype=np.uint64); cnt = C.c_uint32(0)
    L.gf_internal_uni_dump(m._h, items.ctypes.data_as(C.POINTER(C.c_uint64)), len(items), C.byref(cnt))
    it = items[:min(cnt.value, 512)]
    nb = np.array([bin(int(x)).count("1") for x in it[:, 1]], dtype=int)
    tot_w += len(it); tot_b += int(nb.sum()); per_chain.append(len(it))
    for x in nb: hist[x] += 1
per_chain = np.array(per_chain)
print("walkers queued over all chains: %d of %d (%.2f %%), undecided bins %d (%.1f per queued walker)" % (tot_w, 256 * 512, 100.0 * tot_w / (256 * 512), tot_b, tot_b / max(tot_w, 1)))
print("chains with queued walkers: %d of 256; the ten busiest: %s" % ((per_chain > 0).sum(), sorted(per_chain)[-10:]))
print("bins per queued walker, histogram 0..20:", hist.tolist())
print("a half-step parks about %d walkers with %d bins; the settle grid has 256 x 42 = 10752 groups" % (tot_w // 2, tot_b // 2))
