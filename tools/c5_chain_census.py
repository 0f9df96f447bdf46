"""Where the per-chain sampler's time goes on the C5 scan (256 chains x 512 walkers): per-chain census of k_stretch_chain.
python tools/c5_chain_census.py [burnin nsteps]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from golemflavor_amd import scan, mcmc as M

burn, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 200)
pts, nw, make, evals = bench.scan_setup("C5", 0)
jobs = [make(p, g) for g, p in enumerate(pts)]
s = M.DeviceEnsembleSampler(nw, 12, [j.f for j in jobs], seed=25, stream_ids=list(range(len(jobs))))
s.on_nonunitary = "-inf"
import time
s.run_mcmc(np.stack([j.p0 for j in jobs]), burn, storechain=False)
s.reset()
t0 = time.perf_counter(); s.run_mcmc(None, n); dt = time.perf_counter() - t0
st = s.chain_stats()
tot = st["propose_s"] + st["settle_s"] + st["bulk_s"]
order = np.argsort(-tot)
print(json.dumps({"steps": n, "wall_s": dt, "us_per_half_step_wall": 1e6 * dt / (2 * n),
                  "chains_that_waited": int(np.count_nonzero(st["waited_for"])), "chains_with_bulk": int(np.count_nonzero(st["settled_in_bulk"])),
                  "waited_for_total": int(st["waited_for"].sum()), "settled_in_bulk_total": int(st["settled_in_bulk"].sum()),
                  "median_chain_us_per_half_step": 1e6 * float(np.median(tot)) / (2 * n),
                  "propose_us_per_pass_median": 1e6 * float(np.median(st["propose_s"] / np.maximum(st["passes"], 1)))}))
for c in order[:12]:
    print(json.dumps({"chain": int(c), "point": str(pts[c]), "total_ms": round(1e3 * float(tot[c]), 2), "propose_ms": round(1e3 * float(st["propose_s"][c]), 2),
                      "settle_ms": round(1e3 * float(st["settle_s"][c]), 2), "bulk_ms": round(1e3 * float(st["bulk_s"][c]), 2),
                      "waited_for": int(st["waited_for"][c]), "passes_that_waited": int(st["passes_that_waited"][c]), "passes": int(st["passes"][c]),
                      "us_per_waiting_pass": round(1e6 * float(st["settle_s"][c]) / max(int(st["passes_that_waited"][c]), 1), 1),
                      "settled_in_bulk": int(st["settled_in_bulk"][c]), "bulk_settlements": int(st["bulk_settlements"][c])}))
