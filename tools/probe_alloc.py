#!/usr/bin/env python3
"""How long do hipMalloc / hipFree of large blocks take on this box?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from golemflavor_amd import configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.model import Model
with Model(compile_model(Cf.unitary_paramset(), "PRIOR_ONLY", source_ratio=(1, 2, 0))) as m:
    for rep in range(2):
        for mb in (64, 256, 1024, 2048):
            t0 = time.perf_counter(); b = m.alloc(mb << 20); t1 = time.perf_counter(); b.free(); t2 = time.perf_counter()
            print("rep %d: %5d MiB  hipMalloc %.2f ms  hipFree %.2f ms" % (rep, mb, 1e3 * (t1 - t0), 1e3 * (t2 - t1)), flush=True)
