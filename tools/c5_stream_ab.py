"""The C5 scan at the reference's length with the chain read back DURING the run: how good is the overlap, and what changes it?
One process per setting (the copy pool is created once per process).  python tools/c5_stream_ab.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import bench
    out = {"C5": [], "C4": []}
    for rep in range(4):
        for cfg in ("C5",):
            r = bench.extra_scan(0, cfg, bench.REF_BURNIN, bench.REF_NSTEPS)
            if rep:                                  # (the first scan of a process pays the pinned ring, the chain buffer, cold caches)
                out[cfg].append(round(r["seconds"], 3))
    print(json.dumps(out))
    sys.exit(0)
for label, env in (("one pipe for the run", {}), ("a pipeline per block", {"GF_RUN_TO_HOST_NO_PIPE": "1"}), ("one pipe for the run", {}),
                   ("a pipeline per block", {"GF_RUN_TO_HOST_NO_PIPE": "1"}), ("one pipe, 16 copy threads", {"GF_D2H_THREADS": "16"}),
                   ("per block, 16 copy threads", {"GF_RUN_TO_HOST_NO_PIPE": "1", "GF_D2H_THREADS": "16"})):
    e = dict(os.environ, GF_SAMPLER_CHAIN="0")
    e.update(env)
    r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=e, timeout=600)
    print(label, r.stdout.strip(), r.stderr.strip()[-200:], flush=True)
