import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for _ in range(2):
    r = bench.extra_scan(0, "C5", 100, 200)
print(os.environ.get("GF_SAMPLER_LPW", "auto"), r["seconds"], r["phases"], flush=True)
