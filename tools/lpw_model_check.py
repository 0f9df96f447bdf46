"""Does `lanes_per_walker`'s cost model pick well?  Stacked C5 samplers of 16 / 32 / 64 / 128 / 256 chains x 512 walkers (every second ... grid
point, so that each size has its share of chains in the failing region), 100 + 200 steps, lanes per walker forced to 1 / 2 / 4 / 16 and
left to the library.  One process per setting (the switch is read per run, but the graph is captured once)."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import time
    import numpy as np
    from golemflavor_amd import scan, mcmc as mcmc_utils
    nch = int(sys.argv[1])
    pts = scan.sens_grid()
    pts = pts[:: len(pts) // nch][:nch]
    jobs = [scan._SensPoint(p, g, nwalkers=512, device=0) for g, p in enumerate(pts)]
    os.environ["GF_SAMPLER_CHAIN"] = "0"
    s = mcmc_utils.DeviceEnsembleSampler(512, 12, [j.f for j in jobs], seed=25, stream_ids=list(range(len(jobs))))
    s.on_nonunitary = "-inf"
    s.run_mcmc(np.stack([j.p0 for j in jobs]), 100, storechain=False)
    t0 = time.perf_counter(); s.run_mcmc(None, 200, storechain=False); dt = time.perf_counter() - t0
    print(json.dumps({"chains": nch, "GF_SAMPLER_LPW": os.environ.get("GF_SAMPLER_LPW", "(library)"), "us_per_half_step": round(1e6 * dt / 400, 1)}), flush=True)
else:
    for nch in (16, 32, 64, 128, 256):
        for lpw in ("", "1", "2", "4", "16"):
            env = dict(os.environ)
            if lpw:
                env["GF_SAMPLER_LPW"] = lpw
            subprocess.run([sys.executable, __file__, str(nch)], env=env, check=False)
