#!/usr/bin/env python3
"""bench.py's C4 / C5 scan sub-records, each twice in one process (first-launch effects vs steady state)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
for cfg in ("C4", "C4", "C5", "C5"):
    r = bench.extra_scan(0, cfg)
    print(json.dumps({k: r[k] for k in ("workload", "seconds", "phases", "sampling_evals_per_s")}), flush=True)
