#!/usr/bin/env python3
"""cProfile of bench.py's C4 scan sub-record (second run in the process: steady state): where does the host time go?"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
bench.extra_scan(0, cfg)
pr = cProfile.Profile()
pr.enable()
r = bench.extra_scan(0, cfg)
pr.disable()
print(r["seconds"], r["phases"])
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
