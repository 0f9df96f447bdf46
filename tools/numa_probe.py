import os, sys, time, glob
sys.path.insert(0, os.getcwd())
mode = sys.argv[1]
def cpulist(s):
    out = set()
    for part in s.strip().split(","):
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out
# which NUMA node is the visible GPU on?  (via HIP's PCI bus id)
import ctypes as C
hip = C.CDLL("libamdhip64.so")
buf = C.create_string_buffer(64)
hip.hipDeviceGetPCIBusId(buf, 64, 0)
bdf = buf.value.decode().lower()
node = open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip()
local = cpulist(open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read())
allc = set(range(os.cpu_count()))
print("GPU", bdf, "numa node", node, "local cpus", len(local))
if mode == "local": os.sched_setaffinity(0, local)
elif mode == "remote": os.sched_setaffinity(0, allc - local)
import numpy as np
from golemflavor_amd import configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.model import Model
n = 1887436800
with Model(compile_model(Cf.unitary_paramset(), "PRIOR_ONLY")) as m:
    d = m.alloc(n)
    keep = []
    out = None
    for rep in range(6):
        if rep % 2 == 1:
            t0 = time.perf_counter(); out = None; print("   freeing the previous array: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
        t0 = time.perf_counter(); new = d.download((n // 8,)); dt = time.perf_counter() - t0
        out = new
        keep.append(out) if rep % 2 == 0 else None
        print("%s: download %d: %.1f ms = %.1f GB/s" % (mode, rep, 1e3 * dt, n / dt / 1e9))
