#!/usr/bin/env python3
"""Round-4 diagnosis: forward counts of one cb configuration on one GPU under option toggles; where do they differ?"""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lrge_amd import engine, synth_cb

cfg = sys.argv[1] if len(sys.argv) > 1 else "c5_human_half"
spec, Q, T = synth_cb.spec_of(cfg)
dq, dt = spec.device_reads(0, Q), spec.device_reads(Q, T)
res = {}
variants = [("default", {}), ("nofilter", {"NO_GROUP_FILTER": "1"}), ("noprune", {"LPG_NO_PRUNE": "1"}), ("neither", {"NO_GROUP_FILTER": "1", "LPG_NO_PRUNE": "1"}),
            ("nolocal", {"NO_LOCAL_SORT": "1"}), ("hw", {"CHAIN": "hw"})]
if len(sys.argv) > 2:
    variants = [v for v in variants if v[0] in sys.argv[2].split(",")]
for name, opts in variants:
    ctx = engine.Context(0)
    for k, v in opts.items():
        ctx.set_option(k, v)
    Qd = ctx.upload(dq.ptr, dq.offsets, dq.name_ranks())
    Td = ctx.upload(dt.ptr, dt.offsets, dt.name_ranks())
    ix = engine.Index(ctx, Td, 1)
    c, h = ix.overlap_twoset(Qd)
    cn = ctx.counters()
    res[name] = c
    print(name, "sum", int(c.sum()), "batches", cn["batches"], "anchors", cn["anchors"], "kept", cn["anchors_kept"], "chained groups", cn["groups_chained"], "chain anchors", cn["chain_anchors"], flush=True)
    ix.free(); Qd.free(); Td.free(); ctx.close()
base = res[variants[-1][0]] if "neither" not in res else res["neither"]
for name in res:
    d = np.nonzero(res[name] != base)[0]
    print(name, "differs from reference variant on", len(d), "queries", d[:10].tolist(), (d[-3:].tolist() if len(d) else []), flush=True)
