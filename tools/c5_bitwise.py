"""The C5 scan's chains (256 x 512 walkers), bit for bit, between the sampler's launch shapes: one workgroup per chain, the
per-half-step grid kernels, and whichever of the two the sampler's own probe picks.
python tools/c5_bitwise.py [burnin nsteps [npoints]]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from golemflavor_amd import mcmc as M

burn, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (40, 60)
npts = int(sys.argv[3]) if len(sys.argv) > 3 else 256
pts, nw, make, evals = bench.scan_setup("C5", 0)
pts = pts[256 - npts:]                              # the high-scale end of the grid is where proposals are parked
out = {}
for mode, env in (("per chain", {"GF_SAMPLER_CHAIN": "1"}),
                  ("grid", {"GF_SAMPLER_CHAIN": "0"}), ("auto", {})):
    os.environ.pop("GF_SAMPLER_CHAIN", None)
    os.environ.update(env)
    jobs = [make(p, 256 - npts + g) for g, p in enumerate(pts)]
    s = M.DeviceEnsembleSampler(nw, 12, [j.f for j in jobs], seed=25, stream_ids=list(range(256 - npts, 256)))
    s.on_nonunitary = "-inf"
    s.run_mcmc(np.stack([j.p0 for j in jobs]), burn, storechain=False)
    s.reset()
    t0 = time.perf_counter(); s.run_mcmc(None, n); dt = time.perf_counter() - t0
    c = s._fetch(chain=True, lnprob=True, naccepted=True)
    out[mode] = (c[0], c[1], c[2], s.nonunitary_proposals)
    print(json.dumps({"mode": mode, "launch_shape": s.launch_shape(), "seconds": round(dt, 4), "us_per_half_step": round(1e6 * dt / (2 * n), 1), "nonunitary": int(out[mode][3])}), flush=True)
    s.close()
    for j in jobs:
        j.close()
ref = out["grid"]
for mode in ("per chain", "auto"):
    same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out[mode][:3], ref[:3])) and out[mode][3] == ref[3]
    first = None
    if not same:
        d = np.argwhere(out[mode][0] != ref[0])
        first = d[0].tolist() if len(d) else "counters only"
    print(json.dumps({"mode": mode, "bitwise_equal_to_grid": bool(same), "first_difference [chain, step, walker, dim]": first}), flush=True)
