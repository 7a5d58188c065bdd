#!/usr/bin/env python
"""Host-side wall time of every call of one bench step (C4 by default): where the milliseconds between the GPU stages go."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lrge_amd import engine, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4_dmel_twoset"
g, q, t = synth.make_config(cfg)
ctx = engine.Context(0)
qr, tr = engine.name_ranks(q.names, t.names)
dq, dt = torch.from_numpy(q.bases).cuda(), torch.from_numpy(t.bases).cuda()
torch.cuda.synchronize()
avg = np.float32(t.lens().sum()) / np.float32(t.n)
acc = {}
def tick(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
N = 12
for it in range(N + 2):
    if it == 2: acc.clear()
    t0 = time.perf_counter()
    Td = ctx.upload(int(dt.data_ptr()), t.offsets, tr, wait=False); t0 = tick("upload T", t0)
    Qd = ctx.upload(int(dq.data_ptr()), q.offsets, qr, wait=False); t0 = tick("upload Q", t0)
    Qd.presketch(0); t0 = tick("presketch hint", t0)
    ix = engine.Index(ctx, Td, 0); t0 = tick("index build", t0)
    counts, has = ix.overlap_twoset(Qd); t0 = tick("overlap", t0)
    tm = ctx.timings(); cn = ctx.counters(); st = ix.stats(); t0 = tick("introspection", t0)
    est = ctx.estimates(counts, q.lens(), float(avg), t.n, 100); t0 = tick("estimates", t0)
    ix.free(); Qd.free(); Td.free(); t0 = tick("free", t0)
    med = engine.median(est, True, 0.15, 0.65); t0 = tick("median", t0)
tot = sum(acc.values())
for k, v in acc.items():
    print("%-16s %7.3f ms" % (k, v / N * 1e3))
print("%-16s %7.3f ms   (index stage total %.2f, overlap total %.2f)" % ("sum", tot / N * 1e3, ix.build_timings["total"], tm["total"]))
