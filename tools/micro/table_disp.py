import numpy as np, sys
sys.path.insert(0, '.')
from lrge_amd import engine, synth
g, q, t = synth.make_config("c4_dmel_twoset")
ctx = engine.Context(0)
qr, tr = engine.name_ranks(q.names, t.names)
T = ctx.upload(t.bases, t.offsets, tr)
for fix, pw in ((None, None), (None, "3"), (None, "4"), (None, "5"), (None, "6"), ("1", None)):
    ctx.set_option("HT_NO_FIX", fix); ctx.set_option("HT_POWER", pw)
    ix = engine.Index(ctx, T, 0)
    print("HT_NO_FIX", fix, "HT_POWER", pw, "mean displacement", ix.build_counters["table_disp_sum"] / ix.stats()["n_keys"], "keys", ix.stats()["n_keys"])
    ix.free()
