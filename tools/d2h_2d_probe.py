"""Is a PITCHED device-to-host copy slower than a linear one on this box?  The C5 scan's chain leaves the device as blocks of 16 steps x
256 chains: 256 rows of 786 KB, one per chain, a capacity stride apart (gf_internal_d2h_2d: hipMemcpy2DAsync into the pinned ring).
Same bytes, same destination kind (fresh), linear (gf_internal_d2h) against pitched with the scan's geometry.   python tools/d2h_2d_probe.py"""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from golemflavor_amd import _lib, configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.model import Model
L = _lib.lib()
m = Model(compile_model(Cf.unitary_paramset(), "PRIOR_ONLY", source_ratio=(1, 2, 0)), device=0)
L.gf_internal_borrow_stream.restype, L.gf_internal_borrow_stream.argtypes = C.c_int, [C.c_int, C.POINTER(C.c_void_p)]
L.gf_internal_d2h_2d.restype = C.c_int
L.gf_internal_d2h_2d.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]
L.gf_internal_d2h.restype, L.gf_internal_d2h.argtypes = C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
st = C.c_void_p()
assert L.gf_internal_borrow_stream(0, C.byref(st)) == 0
nch, row = 256, 16 * 512 * 12 * 8          # one block of the C5 scan: 256 rows of 786 432 B
cap_stride = 1024 * 512 * 12 * 8           # capacity 1024 stored steps
d = m.alloc(nch * cap_stride)
for rep in range(3):
    for kind in ("linear", "pitched", "pitched, 8 blocks back to back"):
        nblk = 8 if kind.endswith("back") else 1
        a = np.empty((nch, 1000 * 512 * 12))       # the destination: rows = chains, pitch = the whole chain
        t0 = time.perf_counter()
        for b in range(8):
            if kind == "linear":
                rc = L.gf_internal_d2h(0, st, C.c_void_p(a.ctypes.data + b * nch * row), d.ptr, nch * row)
            else:
                rc = L.gf_internal_d2h_2d(0, st, C.c_void_p(a.ctypes.data + b * row), a.strides[0], C.c_void_p(d.ptr.value + b * row), cap_stride, row, nch)
            assert rc == 0
        dt = time.perf_counter() - t0
        print(json.dumps({"copy": kind, "bytes": 8 * nch * row, "seconds": round(dt, 4), "GBps": round(8 * nch * row / dt / 1e9, 1)}), flush=True)
        del a
