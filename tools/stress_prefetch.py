#!/usr/bin/env python3
"""Stress of the big-batch BSM kernel (one lane per walker, LDS-DMA tile prefetch, deferred tier 2) against the small-batch
instances (several lanes per walker, synchronous staging, tiers inline): random windows of a 2 M-row batch, with and without
status, must reproduce the reference rows bit for bit.  usage: stress_prefetch.py [windows]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from golemflavor_amd import configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model

windows = int(sys.argv[1]) if len(sys.argv) > 1 else 150
N = 2_000_003
rng = np.random.default_rng(11)
bad = 0
for name, ps, tex in (("7-col OEU", Cf.texture_paramset(6), Texture.OEU), ("12-col OET", Cf.fr_paramsets(6, (0.4444, 0.0))[1], Texture.OET)):
    nd = len(ps)
    box = np.array(ps.seeds, dtype=float)
    th = rng.uniform(box[:, 0], box[:, 1], size=(N, nd))
    lo, hi = Cf.SCALE_BOUNDARIES[6]
    th[:, -1] = rng.uniform(lo, hi - (0 if tex == Texture.OEU else 4), N)
    th[::977, 0] = 5.0                                            # rows outside the prior box
    desc = compile_model(ps, "BSM_GAUSS", texture=tex, dimension=6, binning=Cf.default_bin_edges(), source_ratio=(0., 1., 0.),
                         bestfit_fr=(1 / 3,) * 3, smearing=0.02)
    with Model(desc) as m:
        d_th = m.alloc(th.nbytes).upload(th)
        d_out, d_fr, d_st = m.alloc(8 * N), m.alloc(24 * N), m.alloc(4 * N)
        # reference: chunks of 5000 rows (16 / 4 lanes per walker, no prefetch, tiers inline)
        t0 = time.time()
        for off in range(0, N, 5000):
            n = min(5000, N - off)
            m.lnprob_device(d_th.at(8 * nd * off), n, d_out.at(8 * off), d_fr.at(24 * off), d_st.at(4 * off))
        m.sync()
        ref = (d_out.download((N,)), d_fr.download((N, 3)), d_st.download((N,), dtype=np.int32))
        print(name, "reference in %.1f s; non-unitary %.3f, out of box %.4f" % (time.time() - t0, np.mean(ref[2] == 2), np.mean(ref[2] == 1)), flush=True)
        for w in range(windows):
            n = int(rng.integers(65536, 1_500_000))
            off = int(rng.integers(0, N - n)) & ~1                # 16-B aligned rows for every width
            with_status = bool(w & 1)
            m.lnprob_device(d_th.at(8 * nd * off), n, d_out.at(0), d_fr.at(0), d_st.at(0) if with_status else None)
            m.sync()
            got = (d_out.download((n,)), d_fr.download((n, 3)))
            want_lp, want_fr = ref[0][off:off + n].copy(), ref[1][off:off + n]
            if not with_status:                                   # without status non-unitary walkers keep their value
                sel = ref[2][off:off + n] == 2
                ok = np.array_equal(got[0][~sel], want_lp[~sel], equal_nan=True) and np.array_equal(got[1], want_fr, equal_nan=True)
            else:
                st = d_st.download((n,), dtype=np.int32)
                ok = np.array_equal(got[0], want_lp, equal_nan=True) and np.array_equal(got[1], want_fr, equal_nan=True) and np.array_equal(st, ref[2][off:off + n])
            if not ok:
                bad += 1
                print("MISMATCH", name, "window", w, "off", off, "n", n, "status", with_status, flush=True)
        print(name, windows, "windows done, mismatches so far:", bad, flush=True)
print("stress_prefetch:", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
