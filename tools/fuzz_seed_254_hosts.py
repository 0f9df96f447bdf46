"""Seed 254 of the extended BSM fuzz, walker 1991: the ORACLE on this host -- per-bin residuals (long double), the host build of the emulated chain,
and the batch verdict -- so that two hosts can be compared (CPU only).  The verdict of this walker depends on the CPU the oracle runs on:
profiles/r04/fuzz_extended.txt."""
import ctypes as C, math, os, subprocess, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from golemflavor_amd import configs as Cf
from golemflavor_amd.enums import Texture
from oracle import oracle
LD = np.longdouble
seed = 254
rng = np.random.default_rng(7000 + seed)
dim = int(rng.integers(3, 9))
tex = [Texture.OEU, Texture.OET, Texture.OUT][int(rng.integers(0, 3))]
src = rng.dirichlet((1, 1, 1)) if rng.random() < 0.5 else np.eye(3)[int(rng.integers(0, 3))]
nbins = int(rng.choice([1, 2, 5, 20, 33, 64]))
lo_e, hi_e = 10 ** rng.uniform(4, 5), 10 ** rng.uniform(6, 7.5)
edges = np.logspace(np.log10(lo_e), np.log10(hi_e), nbins + 1)
print(dim, tex, nbins, edges)
th = np.array([3.3749342911919472e-01, 9.5709743952288273e-01, 6.1962979700647303e-01, 2.6252481956155558e+00, 7.3807511925658526e-23, 2.4620710022010561e-21, -4.2165923496203476e+01])
out = "/tmp/libx87host_probe.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-fPIC", "-shared", "-ffp-contract=off", "-o", out, "/root/repo/tests/x87/x87_host.cpp"])
L = C.CDLL(out)
L.x87t_bin_residual.restype = C.c_double
L.x87t_bin_residual.argtypes = [C.POINTER(C.c_double)] * 2 + [C.c_double] * 4 + [C.c_int, C.c_void_p, C.c_void_p]
LO = oracle.lib()
Z = 1e-9
TEX = {1: (0.5, 1.0, Z, Z), 2: (Z, 0.25, Z, Z), 3: (Z, 1.0, 0.5, Z)}
arr = lambda x: (C.c_double * len(x))(*[float(v) for v in x])
t = int(tex.value)
centres = np.sqrt(edges[:-1] * edges[1:])
sc2 = math.pow(10., th[6])
smu, npu = np.zeros(18, dtype=LD), np.zeros(18, dtype=LD)
LO.orc_angles_to_u_ldout(arr(th[:4]), smu.ctypes.data_as(C.c_void_p))
LO.orc_angles_to_u_ldout(arr(TEX[t]), npu.ctypes.data_as(C.c_void_p))
for e in centres:
    o = np.zeros(96, dtype=LD)
    LO.orc_debug_bsmu_ld(arr(TEX[t]), C.c_double(th[6]), dim, C.c_double(e), arr(th[4:6]), arr(th[:4]), o.ctypes.data_as(C.c_void_p))
    ro = float(o[54])
    ra = L.x87t_bin_residual(arr(th[:4]), arr(TEX[t]), th[4], th[5], sc2, e, dim, smu.ctypes.data_as(C.c_void_p), npu.ctypes.data_as(C.c_void_p))
    rb = L.x87t_bin_residual(arr(th[:4]), arr(TEX[t]), th[4], th[5], sc2, e, dim, None, npu.ctypes.data_as(C.c_void_p))
    print("E %.4e oracle residual %.4e  emulation (host matrices) %.4e  emulation (emulated sm angles) %.4e" % (e, ro, ra, rb))
twelve = rng.random() < 0.5
ps = Cf.fr_paramsets(dim, (0.4444, 0.0))[1] if twelve else Cf.texture_paramset(dim)
bf = tuple(rng.dirichlet((3, 3, 3)))
kw = dict(texture=tex, dimension=dim, binning=edges, source_ratio=src, bestfit_fr=bf, smearing=float(rng.choice([0.02, 0.2])))
om = oracle.make_model(ps, "BSM_GAUSS", **dict(kw, texture=tex.name))
print("columns", [p.name for p in ps], "twelve", twelve)
r = oracle.unitarity_residual_batch(om, th[None, :])
ref, ref_fr, ref_st = oracle.lnprob_batch(om, th[None, :], want_fr=True, want_status=True)
print("oracle batch r80", r, "status", ref_st, "lnprob", ref)
n = int(rng.choice([64, 700, 3000, 9000]))
box = np.array(ps.seeds, dtype=float)
TH = rng.uniform(box[:, 0], box[:, 1], size=(n, len(ps)))
lo, hi = Cf.SCALE_BOUNDARIES[dim]
TH[:, -1] = rng.uniform(lo, lo + rng.uniform(0.3, 1.0) * (hi - lo), n)
wild = rng.random(n) < 0.01
TH[wild, rng.integers(0, len(ps), wild.sum())] = rng.choice([np.nan, np.inf, -np.inf], wild.sum())
print("n", n, "walker 1991 equals printed theta:", np.array_equal(TH[1991], th), TH[1991] - th)
for rep in range(3):
    R = oracle.unitarity_residual_batch(om, TH)
    ref, ref_fr, ST = oracle.lnprob_batch(om, TH, want_fr=True, want_status=True)
    print("batch: r80[1991] %.4e status %d; flagged total %d" % (R[1991], ST[1991], int((ST == 2).sum())))
