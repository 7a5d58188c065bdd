"""Host-side pack against NUMA placement: the source buffer is filled by threads of one node, the pack runs on the cores of
the same / the other / all nodes, with 32 / 64 / 96 worker threads.  Development harness (gpurun)."""
import os, sys, time, glob
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lrge_amd import engine

def cpus(node):
    s = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
    out = []
    for part in s.split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out

nodes = sorted(int(p.rsplit("node", 1)[1]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
print("nodes:", nodes, [len(cpus(n)) for n in nodes], flush=True)
allc = sorted(os.sched_getaffinity(0))
n, L = 100000, 7200
rng = np.random.Generator(np.random.PCG64(1))
bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n * L, dtype=np.uint8)]
offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
for src_node in nodes[:2]:
    os.sched_setaffinity(0, set(cpus(src_node)) & set(allc))
    c0 = engine.Context(0)
    pin = c0.host_alloc(bases.size); pin.array[:] = bases          # first touch on src_node
    for where in ["same", "other", "all"]:
        if where == "other" and len(nodes) < 2: continue
        tgt = set(cpus(src_node)) if where == "same" else (set(cpus(nodes[1 - nodes.index(src_node)])) if where == "other" else set(allc))
        os.sched_setaffinity(0, tgt & set(allc))
        for th in (32, 64, 96):
            c = engine.Context(0)
            c.set_option("HOST_PACK_THREADS", str(th))
            best = 1e9
            for it in range(4):
                t0 = time.perf_counter(); S = c.upload(pin, offs, None, wait=True); dt = time.perf_counter() - t0; S.free(); best = min(best, dt)
            print("source on node %d, pack on %s cores, %d threads: %.2f ms" % (src_node, where, th, best * 1e3), flush=True)
            c.close()
    os.sched_setaffinity(0, set(allc))
    pin.free(); c0.close()
