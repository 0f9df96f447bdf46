"""The C5 scan at the reference's length into the registered result arena (ABI 5): the chain read back during the run as one pitched DMA per
block of steps against one linear DMA per chain row (GF_PIPE_ROWS_1D=1), and after the run.  Each variant in a process of its own
(the variants free and map 12.6 GB: one process would time its neighbour's munmap).  python tools/arena_scan_ab.py [variant]"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = {"pitched": {}, "rows_1d": {"GF_PIPE_ROWS_1D": "1"}, "after_the_run": {"GF_SCAN_NO_STREAMED_CHAIN": "1"}, "ring": {"GF_NO_DIRECT_D2H": "1"}}
if len(sys.argv) > 1:
    import bench  # noqa: E402
    from golemflavor_amd import scan  # noqa: E402
    arena = scan.ResultArena(12582912000)
    scan.set_result_arena(arena)
    for rep in range(4):
        r = bench.extra_scan(0, "C5", 200, 1000)
        print(json.dumps({"variant": sys.argv[1], "rep": rep, "seconds": round(r["seconds"], 4), "sampling_s": round(r["sampling_s"], 4),
                          "d2h_s": round(r["d2h_s"], 4), "us_per_half_step": round(1e6 * r["sampling_s"] / 2400, 1)}), flush=True)
        del r
    r = bench.extra_scan(0, "C4", 200, 1000)
    print(json.dumps({"variant": sys.argv[1], "C4 200+1000 seconds": round(r["seconds"], 4), "d2h_s": round(r["d2h_s"], 4), "GBps": round(9.437184 / r["d2h_s"], 1)}), flush=True)
else:
    for rnd in range(2):
        for name, env in VARIANTS.items():
            subprocess.run([sys.executable, __file__, name], env=dict(os.environ, **env), check=False)
