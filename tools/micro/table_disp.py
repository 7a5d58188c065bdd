import numpy as np, sys
sys.path.insert(0, '.')
from lrge_amd import engine, synth
g, q, t = synth.make_config("c4_dmel_twoset")
ctx = engine.Context(0)
qr, tr = engine.name_ranks(q.names, t.names)
T = ctx.upload(t.bases, t.offsets, tr)
for fix in (None, "1"):
    ctx.set_option("HT_NO_FIX", fix)
    ix = engine.Index(ctx, T, 0)
    print("HT_NO_FIX", fix, "mean displacement", ix.build_counters["table_disp_sum"] / ix.stats()["n_keys"], "keys", ix.stats()["n_keys"])
    ix.free()
