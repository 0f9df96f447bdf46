"""A/B of the C5 scan's sampling phase on one GPU: one workgroup per chain (k_stretch_chain, round 4) against the per-half-step grid
kernels + k_stretch_settle (GF_SAMPLER_CHAIN=0), with the chain read back after the run (GF_SCAN_NO_STREAMED_CHAIN=1: the sampling
alone) and during it (the default).  python tools/c5_sampler_ab.py [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rows = []
for rep in range(reps):
    for chain in ("1", "0"):
        for streamed in (False, True):
            os.environ["GF_SAMPLER_CHAIN"] = chain
            if streamed:
                os.environ.pop("GF_SCAN_NO_STREAMED_CHAIN", None)
            else:
                os.environ["GF_SCAN_NO_STREAMED_CHAIN"] = "1"
            for burn, n in ((100, 200), (200, 1000)):
                r = bench.extra_scan(0, "C5", burn, n)
                row = {"rep": rep, "per_chain_workgroups": chain == "1", "read_back_during_the_run": streamed, "burnin": burn, "nsteps": n,
                       "seconds": round(r["seconds"], 4), "sampling_s": round(r["sampling_s"], 4), "d2h_s": round(r["d2h_s"], 4),
                       "us_per_half_step": round(1e6 * r["sampling_s"] / (2 * (burn + n)), 2) if not streamed else None,
                       "nonunitary": r.get("nonunitary_proposals", {}).get("nonunitary_proposals_rejected")}
                rows.append(row)
                print(json.dumps(row), flush=True)
