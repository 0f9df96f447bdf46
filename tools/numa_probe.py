"""Does the NUMA placement of the read-back's host side matter on this box?  D2H of 4.7 GB into fresh memory with the process (and so
the library's copy threads and the pages they first touch) confined to each NUMA node in turn.  python tools/numa_probe.py"""
import ctypes as C, glob, json, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def cpulist(s):
    out = []
    for part in s.strip().split(","):
        if "-" in part:
            a, b = part.split("-"); out += list(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out

if len(sys.argv) > 1 and sys.argv[1] == "child":
    cpus = cpulist(sys.argv[2])
    if cpus:
        os.sched_setaffinity(0, cpus)
    from golemflavor_amd import _lib, configs as Cf
    from golemflavor_amd.descriptor import compile_model
    from golemflavor_amd.model import Model
    L = _lib.lib()
    m = Model(compile_model(Cf.unitary_paramset(), "PRIOR_ONLY", source_ratio=(1, 2, 0)), device=0)
    n = 4_718_592_000
    d = m.alloc(n)
    rates = []
    for rep in range(4):
        a = np.empty(n // 8)
        t0 = time.perf_counter()
        _lib.check(L.gf_memcpy_d2h(m._h, a.ctypes.data_as(C.c_void_p), d.ptr, a.nbytes), "d2h")
        rates.append(round(n / (time.perf_counter() - t0) / 1e9, 1))
        del a
    print(json.dumps({"cpus": sys.argv[2][:40], "GBps_fresh_destination": rates}))
    sys.exit(0)

nodes = {}
for p in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    nodes[p.split("/")[-2]] = open(p).read().strip()
gpu_nodes = {}
for p in glob.glob("/sys/class/drm/card*/device/numa_node"):
    gpu_nodes[p.split("/")[4]] = open(p).read().strip()
print(json.dumps({"numa_nodes": nodes, "gpu_numa_node": gpu_nodes, "allowed_cpus": len(os.sched_getaffinity(0))}))
for name, cl in [("all", "")] + list(nodes.items()):
    r = subprocess.run([sys.executable, __file__, "child", cl], capture_output=True, text=True, timeout=300)
    print(name, r.stdout.strip(), r.stderr.strip()[-200:])
