"""The C5 sampler by the grid of k_stretch_settle (GF_SETTLE_BLOCKS; the library launches one block per CU): each value in a process of its
own (the switch is read once).  python tools/c5_settle_blocks_ab.py"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import bench  # noqa: E402
    os.environ["GF_SAMPLER_CHAIN"] = "0"
    for rep in range(3):
        r = bench.extra_c5_sampler(0)
        print(json.dumps({"GF_SETTLE_BLOCKS": sys.argv[1], "rep": rep, "us_per_half_step": round(r["us_per_half_step_stored_run"], 2), "nonunitary": r["nonunitary_proposals"]}), flush=True)
else:
    for b in ("0", "32", "64", "96", "128", "192", "256", "512"):
        subprocess.run([sys.executable, __file__, b], env=dict(os.environ, **({"GF_SETTLE_BLOCKS": b} if b != "0" else {})), check=False)
