import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for _ in range(2):
    r = bench.extra_scan(0, "C4", 100, 200)
    print(r["seconds"], r["phases"], flush=True)
