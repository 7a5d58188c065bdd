import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from lrge_amd import engine
ctx = engine.Context(0)
n, L = 100000, 7200
rng = np.random.Generator(np.random.PCG64(1))
bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n * L, dtype=np.uint8)]
offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
pin = ctx.host_alloc(bases.size); pin.array[:] = bases
for opt in (None, ("HOST_PACK_THREADS", "16"), ("HOST_PACK_THREADS", "64"), ("NO_HOST_PACK", "1")):
    c = engine.Context(0)
    if opt: c.set_option(*opt)
    p2 = c.host_alloc(bases.size); p2.array[:] = bases
    for it in range(4):
        t0 = time.perf_counter(); S = c.upload(p2, offs, None, wait=True); dt = time.perf_counter() - t0; S.free()
    print(opt, "upload of %.0f Mbases: %.2f ms" % (bases.size / 1e6, dt * 1e3), flush=True)
    p2.free(); c.close()
