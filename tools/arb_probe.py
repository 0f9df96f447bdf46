"""Where the arbitration time goes: per workload, the (walker, bin) pairs that reach k_uni_resolve, pairs per walker, and the
time of a status launch against a no-status launch.  GPU.  usage: python tools/arb_probe.py [n]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from golemflavor_amd import _lib, configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model
from common import BIN_EDGES, uniform_theta

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
L = _lib.lib()
L.gf_internal_uni_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]


def stats(m):
    out = (C.c_uint32 * 3)()
    L.gf_internal_uni_stats(m._h, out)
    return list(out)


for label, dim, tex, twelve, span, fixed_sm in (("bench failing region", 6, Texture.OEU, True, None, False),
                                                  ("OET d6 full range", 6, Texture.OET, True, None, False),
                                                  ("OUT d6 full range", 6, Texture.OUT, False, None, False),
                                                  ("OUT d3 full range", 3, Texture.OUT, False, None, False)):
    ps = Cf.fr_paramsets(dim, (0.4, 0.0))[1] if twelve else Cf.texture_paramset(dim)
    lo, hi = Cf.SCALE_BOUNDARIES[dim]
    rng = np.random.default_rng(1)
    th = uniform_theta(ps, n, rng, seeds=True)
    th[:, -1] = rng.uniform(lo, hi, n)
    kw = dict(dimension=dim, binning=BIN_EDGES, source_ratio=(0., 1., 0.), bestfit_fr=(1 / 3,) * 3, smearing=0.02)
    with Model(compile_model(ps, "BSM_GAUSS", texture=tex, **kw)) as m:
        d_th = m.alloc(th.nbytes).upload(th); d_out = m.alloc(8 * n); d_st = m.alloc(4 * n)
        for rep in range(2):
            m.lnprob_device(d_th.ptr, n, d_out.ptr, None, d_st.ptr); m.sync()
        s0 = stats(m)
        t0 = time.perf_counter()
        for rep in range(3):
            m.lnprob_device(d_th.ptr, n, d_out.ptr, None, d_st.ptr)
        m.sync(); dt = (time.perf_counter() - t0) / 3
        s1 = stats(m)
        st = d_st.download((n,), dtype=np.int32)
        m.lnprob_device(d_th.ptr, n, d_out.ptr, None, None); m.sync()
        t0 = time.perf_counter()
        for rep in range(3):
            m.lnprob_device(d_th.ptr, n, d_out.ptr, None, None)
        m.sync(); dt0 = (time.perf_counter() - t0) / 3
    pairs = (s1[1] - s0[1]) / 3.0
    print("%-22s n %d  status %.3f ms (%.2e evals/s)  no status %.3f ms  pairs/launch %.0f (%.3f per walker)  arbitration ~%.3f ms -> %.2e pairs/s  nonunitary %.4f"
          % (label, n, 1e3 * dt, n / dt, 1e3 * dt0, pairs, pairs / n, 1e3 * (dt - dt0), pairs / max(dt - dt0, 1e-9), np.mean(st == 2)), flush=True)

# ---- who reaches arbitration: pairs per queued walker, and the outcome -----------------------------------------------
L.gf_internal_uni_dump.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint32)]
for label, dim, tex, twelve in (("bench failing region", 6, Texture.OEU, True), ("OUT d6 full range", 6, Texture.OUT, False)):
    ps = Cf.fr_paramsets(dim, (0.4, 0.0))[1] if twelve else Cf.texture_paramset(dim)
    lo, hi = Cf.SCALE_BOUNDARIES[dim]
    rng = np.random.default_rng(1)
    th = uniform_theta(ps, n, rng, seeds=True)
    th[:, -1] = rng.uniform(lo, hi, n)
    kw = dict(dimension=dim, binning=BIN_EDGES, source_ratio=(0., 1., 0.), bestfit_fr=(1 / 3,) * 3, smearing=0.02)
    with Model(compile_model(ps, "BSM_GAUSS", texture=tex, **kw)) as m:
        d_th = m.alloc(th.nbytes).upload(th); d_out = m.alloc(8 * n); d_st = m.alloc(4 * n)
        m.lnprob_device(d_th.ptr, n, d_out.ptr, None, d_st.ptr); m.sync()
        st = d_st.download((n,), dtype=np.int32)
        items = np.zeros((n, 2), dtype=np.uint64)
        cnt = C.c_uint32(0)
        L.gf_internal_uni_dump(m._h, items.ctypes.data_as(C.POINTER(C.c_uint64)), len(items), C.byref(cnt))
    it = items[:min(cnt.value, len(items))]
    uw = it[:, 0].astype(np.int64)
    bits = (it[:, 1:2] >> np.arange(20, dtype=np.uint64)[None, :]) & np.uint64(1)
    counts = bits.sum(axis=1).astype(np.int64)
    w = np.repeat(uw, counts)
    b = np.nonzero(bits)[1]
    failed = st[uw] == 2
    print(label, "pairs", int(counts.sum()), "walkers queued", len(uw), "(%.3f of all)" % (len(uw) / n), "of them non-unitary %.3f" % failed.mean())
    print("  pairs per queued walker: histogram", np.bincount(counts, minlength=21)[1:21].tolist())
    print("  ... of the walkers that FAIL:      ", np.bincount(counts[failed], minlength=21)[1:21].tolist())
    print("  ... of the walkers that pass:      ", np.bincount(counts[~failed], minlength=21)[1:21].tolist())
    hb = np.zeros(20, dtype=np.int64)
    np.add.at(hb, b, 1)
    print("  pairs by energy bin:", hb.tolist())
    print("  non-unitary walkers NOT queued (condemned by tier 2):", int(np.sum(st == 2) - failed.sum()), "of", int(np.sum(st == 2)))
