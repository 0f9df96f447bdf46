import sys, os, json
sys.path.insert(0, os.getcwd())
import bench
r = bench.extra_c3(0)
print(json.dumps({k: r[k] for k in ("kernel_ms", "draws_per_s", "frac_of_hbm_peak")}), json.dumps(r["with_angles"]))
