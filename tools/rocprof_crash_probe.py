import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from golemflavor_amd import dist as gdist
mode = sys.argv[1]
keep = []
if mode == "dummy":
    keep.append(np.empty(12_582_912_000 // 8))          # an untouched 12.6 GB host array, as run_mcmc_to_host allocates
for cfg, b, n in (("C4", 100, 200), ("C5", 100, 200), ("C4", 200, 1000), ("C5", 200, 1000)):
    rec = bench.extra_scan(0, cfg, b, n)
    print(cfg, b, n, rec["seconds"], flush=True)
