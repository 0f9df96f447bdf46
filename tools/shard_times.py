"""What each rank of an 8-rank job would do, run one after the other on ONE GPU: the reference-length scans on the shard
`dist.shard(n_points, r, 8)` of every rank r (same seeds, same chains as the whole grid's), into the registered arena.  The slowest shard
is the prediction for N = 8 (nothing is shared between the ranks of a scan but the host's memory system).  python tools/shard_times.py [world]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from golemflavor_amd import configs as Cf, dist as gdist, scan  # noqa: E402
from golemflavor_amd.descriptor import compile_model  # noqa: E402
from golemflavor_amd.model import Model  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
arena = scan.ResultArena(12582912000 // world + (64 << 20))
scan.set_result_arena(arena)
stage = Model(compile_model(Cf.unitary_paramset(), "PRIOR_ONLY", source_ratio=(1, 2, 0)), device=0)
for cfg in ("C4", "C5"):
    pts, nw, make, evals = bench.scan_setup(cfg, 0)
    rows = []
    for r in range(world):
        mine = gdist.shard(len(pts), r, world)
        for rep in range(2):                                   # the second run of a shard: buffers cached, kernels loaded
            sub = [pts[g] for g in mine]                       # (the shard as a grid of its own: local random streams, the same posteriors)
            chains = scan.run_points(sub, list(range(len(sub))), make, 200, 1000, stacked=True, gather=scan.DeviceGather(None, 0, 1, stage))
        ph = dict(scan.PHASES)
        total = ph["setup"] + ph["sampling"] + ph.get("gather", 0.0)
        rows.append(total)
        print(json.dumps({"scan": cfg, "rank": r, "of": world, "points": len(mine), "seconds": round(total, 4), "setup": round(ph["setup"], 4),
                          "sampling": round(ph["sampling"], 4), "delivery": round(ph.get("gather", 0.0), 4),
                          "launch_shape": (scan.LAST_NONUNITARY.get("launch_shape") or {}).get("shape")}), flush=True)
        del chains
    print(json.dumps({"scan": cfg, "predicted_seconds_at_N=%d" % world: round(max(rows), 4), "fastest_shard": round(min(rows), 4)}), flush=True)
