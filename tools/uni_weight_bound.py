#!/usr/bin/env python3
"""The 80-bit unitarity residual of the reference (fr.py:489-494, evaluated by the CPU oracle per energy bin) against the
weight `a` of the SM term in the trace-normalised Hamiltonian, over random walkers of every operator dimension and
texture (incl. Texture.NONE with random NP angles): the bound behind tier 1 of the kernels' unitarity verdict
(gf_bsm_device.hpp).  CPU only (test infrastructure: uses oracle/)."""
import os, sys, ctypes as C, numpy as np, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import oracle as O
from golemflavor_amd import configs as Cf
from common import BIN_EDGES
LO = O.lib()
LD=np.longdouble
Zz=1e-9
TEX = {1:(0.5,1.0,Zz,Zz), 2:(Zz,0.25,Zz,Zz), 3:(Zz,1.0,0.5,Zz)}
centres = np.sqrt(BIN_EDGES[:-1]*BIN_EDGES[1:])
def arr(x): return (C.c_double*len(x))(*x)
rng=np.random.default_rng(11)
rec=[]
for trial in range(6000):
    dim = int(rng.integers(3,9))
    texk = int(rng.integers(1,5))
    ps = Cf.texture_paramset(dim)
    box=np.array(ps.seeds,float)
    th = rng.uniform(box[:,0],box[:,1])
    if rng.random()<0.3:
        rr=np.array(ps.ranges,float); th[:4]=rng.uniform(rr[:4,0],rr[:4,1])
    lo,hi = Cf.SCALE_BOUNDARIES[dim]
    th[6] = rng.uniform(lo,hi)
    npang = TEX[texk] if texk<4 else (rng.uniform(0,1),rng.uniform(0,1),rng.uniform(0,1),rng.uniform(0,2*np.pi))
    sc2=10.0**th[6]
    for e in centres[[0,6,12,16,19]]:
        out = np.zeros(96, dtype=LD)
        LO.orc_debug_bsmu_ld(arr(npang), C.c_double(th[6]), dim, C.c_double(e), arr(th[4:6]), arr(th[:4]), out.ctypes.data_as(C.c_void_p))
        r80=float(out[54])
        sm=(th[4]+th[5])/(2*e); npt=1.01*sc2*e**(dim-3)
        a=sm/(sm+npt)
        rec.append((dim,texk,a,r80))
rec=np.array(rec)
a=rec[:,2]; r=rec[:,3]; ok=np.isfinite(r)
print('n',len(rec))
for texk in (1,2,3,4):
    s=ok&(rec[:,1]==texk)
    prod=(r*a)[s]
    print('tex',texk,'max r80*a = %.3g'%prod.max(), ' max r80 with a>=1e-12: %.3g'%r[s&(a>=1e-12)].max(), ' with a>=1e-11: %.3g'%r[s&(a>=1e-11)].max(), ' a>=1e-10: %.3g'%r[s&(a>=1e-10)].max(), 'min a with r80>=1e-7:', a[s&(r>=1e-7)].max() if (s&(r>=1e-7)).any() else None)
print('max over everything of r80 * a for a < 1e-8: %.3g' % (r * a)[ok & (a < 1e-8)].max())
for texk in (1, 2, 3, 4):
    for lo, hi in ((1e-11, 1e-8), (1e-14, 1e-11), (1e-17, 1e-14), (1e-22, 1e-17)):
        s = ok & (rec[:, 1] == texk) & (a >= lo) & (a < hi) & (r > 0)
        if s.sum() < 5:
            continue
        q = np.log10((r * a)[s])
        print('tex', texk, 'a in [%g,%g)' % (lo, hi), 'n', s.sum(), 'log10(r80*a): min %.2f 1%% %.2f med %.2f 99%% %.2f max %.2f' % (
            q.min(), np.quantile(q, .01), np.median(q), np.quantile(q, .99), q.max()), ' frac fail %.3f' % np.mean(r[s] >= 1e-7))
