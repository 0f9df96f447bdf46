"""Would a per-chain choice of lanes per walker pay?  The C5 grid's chains at the two top scales (64 chains: all that ever park a proposal)
and the other 192, each set as a stacked sampler of its own, lanes per walker forced: the half-step of each (propose + settle).  If the
heavy set at 16 lanes and the light set at 2 are both well below the whole grid's 80 us, two concurrent launches would be too."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import time
    import numpy as np
    from golemflavor_amd import scan, mcmc as mcmc_utils
    which = sys.argv[1]
    pts = scan.sens_grid()
    heavy_idx = set(json.load(open("/tmp/gf_heavy_chains.json")))
    sel = [p for g, p in enumerate(pts) if (g in heavy_idx) == (which == "heavy")]
    jobs = [scan._SensPoint(p, g, nwalkers=512, device=0) for g, p in enumerate(sel)]
    os.environ["GF_SAMPLER_CHAIN"] = "0"
    s = mcmc_utils.DeviceEnsembleSampler(512, 12, [j.f for j in jobs], seed=25, stream_ids=list(range(len(jobs))))
    s.on_nonunitary = "-inf"
    s.run_mcmc(np.stack([j.p0 for j in jobs]), 100, storechain=False)
    s.run_mcmc(None, 200, storechain=False)                      # (clocks up)
    t0 = time.perf_counter(); s.run_mcmc(None, 400, storechain=False); dt = time.perf_counter() - t0
    print(json.dumps({"set": which, "chains": len(sel), "GF_SAMPLER_LPW": os.environ.get("GF_SAMPLER_LPW", "(library)"), "us_per_half_step": round(1e6 * dt / 800, 1),
                      "nonunitary": int(s.nonunitary_proposals)}), flush=True)
elif True:
    # which chains ever wait for a settled proposal: a census run of the whole grid with one workgroup per chain
    code = '''
import json, os, sys
sys.path.insert(0, %r)
import numpy as np
from golemflavor_amd import scan, mcmc as mcmc_utils
os.environ["GF_SAMPLER_CHAIN"] = "1"
pts = scan.sens_grid()
jobs = [scan._SensPoint(p, g, nwalkers=512, device=0) for g, p in enumerate(pts)]
s = mcmc_utils.DeviceEnsembleSampler(512, 12, [j.f for j in jobs], seed=25, stream_ids=list(range(len(jobs))))
s.on_nonunitary = "-inf"
s.run_mcmc(np.stack([j.p0 for j in jobs]), 100, storechain=False)
s.reset()
s.run_mcmc(None, 100, storechain=False)
st = s.chain_stats()
heavy = [int(i) for i in np.flatnonzero((st["waited_for"] > 0) | (st["settled_in_bulk"] > 0))]
json.dump(heavy, open("/tmp/gf_heavy_chains.json", "w"))
print(json.dumps({"chains that parked a proposal in 100 steps": len(heavy)}))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-c", code], check=True)
    for which, lpws in (("heavy", ("2", "4", "16")), ("light", ("1", "2", "4"))):
        for lpw in lpws:
            subprocess.run([sys.executable, __file__, which], env=dict(os.environ, GF_SAMPLER_LPW=lpw), check=False)
