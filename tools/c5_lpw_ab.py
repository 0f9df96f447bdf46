"""The C5 sampler (grid kernels) by lanes per walker of the half-step kernel (GF_SAMPLER_LPW = 1 / 4 / 16; the library picks 4 for
256 chains x 512 walkers): sampling only, chain kept on the device.  python tools/c5_lpw_ab.py [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
os.environ["GF_SAMPLER_CHAIN"] = "0"
for rep in range(reps):
    for lpw in ("", "1", "2", "4", "16"):
        if lpw:
            os.environ["GF_SAMPLER_LPW"] = lpw
        else:
            os.environ.pop("GF_SAMPLER_LPW", None)
        r = bench.extra_c5_sampler(0)
        print(json.dumps({"rep": rep, "GF_SAMPLER_LPW": lpw or "(library's choice)", "us_per_half_step": round(r["us_per_half_step_stored_run"], 2),
                          "burnin_s": round(r["burnin_s"], 4), "stored_run_s": round(r["stored_run_s"], 4), "nonunitary": r["nonunitary_proposals"]}), flush=True)
