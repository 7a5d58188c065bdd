import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lrge_amd import engine, synth
cfg = synth.CONFIGS["c2_bact_twoset"]
genome = synth.random_genome(cfg["genome"], cfg["seed"])
t = synth.sample_reads(genome, cfg["T"], "ont", seed=cfg["seed"] + 1, name_prefix="t")
q = synth.sample_reads(genome, cfg["Q"], "ont", seed=cfg["seed"] + 101, name_prefix="q0_")
ctx = engine.Context(0)
qr, tr = engine.name_ranks(q.names, t.names)
Qd = ctx.upload(q.bases, q.offsets, qr); Td = ctx.upload(t.bases, t.offsets, tr)
qlens = q.lens(); avg_t = np.float32(t.lens().sum()) / np.float32(t.n)
acc = {}
def T(name, f):
    t0 = time.perf_counter(); r = f(); acc[name] = acc.get(name, 0) + time.perf_counter() - t0; return r
for it in range(23):
    if it == 3: acc.clear(); t_all = time.perf_counter()
    ix = T("index", lambda: engine.Index(ctx, Td, 0))
    tb = T("misc", lambda: (dict(ix.build_timings), dict(ix.build_counters)))
    counts, has = T("overlap", lambda: ix.overlap_twoset(Qd))
    tm = T("misc", lambda: (ctx.timings(), ctx.counters(), ix.stats()))
    est = T("estimates", lambda: ctx.estimates(counts, qlens, float(avg_t), t.n, 100))
    T("free", lambda: ix.free())
    med = T("median", lambda: engine.median(est, True, 0.15, 0.65))
tot = time.perf_counter() - t_all
print("per step ms: total %.3f" % (tot / 20 * 1e3), {k: round(v / 20 * 1e3, 3) for k, v in acc.items()})
print("gpu stage totals: index %.3f overlap %.3f" % (tb[0].get("total", 0), tm[0][0].get("total", 0) if isinstance(tm[0], tuple) else 0))
print(tb[0]); print(tm[0])
