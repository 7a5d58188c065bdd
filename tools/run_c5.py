#!/usr/bin/env python
"""Run a counter-based configuration (lrge_amd.synth_cb.CONFIGS; default: the full H. sapiens-scale BASELINE configs[4])
on one MI355X: the reads are generated straight into HBM by the device twin of the generator, the forward and / or the
inverse (--use-min-ref) strategy are timed, and samples are checked against the CPU oracle fed by the host twin.

  python tools/run_c5.py --config c5_human_twoset --forward --inverse --check-queries 64 --check-targets 20000
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def log(*a):
    print("[run_c5 %7.1f s]" % (time.perf_counter() - T0), *a, file=sys.stderr, flush=True)


T0 = time.perf_counter()


def gb(x):
    return round(x / 2**30, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c5_human_twoset")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--preset", default="pb", choices=["ont", "pb"])
    ap.add_argument("--forward", action="store_true")
    ap.add_argument("--inverse", action="store_true")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--check-queries", type=int, default=0, help="forward: first N queries against the oracle's index of ALL targets (slow at full size)")
    ap.add_argument("--check-targets", type=int, default=0, help="inverse: a range of N streamed targets against the oracle's index of the queries")
    ap.add_argument("--twin-check", type=int, default=2000, help="reads of each set compared between the two generator twins")
    ap.add_argument("--keep-ascii", action="store_true", help="leave the ASCII reads resident beside the packed ones (the bench's situation)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from lrge_amd import engine, synth_cb
    preset = 1 if a.preset == "pb" else 0
    spec, Q, T = synth_cb.spec_of(a.config, a.scale)
    out = {"config": a.config, "scale": a.scale, "preset": "ava-pb" if preset else "ava-ont", "generator": "counter-based (tools/synth)",
           "genome_size_true": spec.gsize, "n_query": Q, "n_target": T}
    f0, tot = synth_cb.mem_info()
    log("device memory: %.1f GB free of %.1f" % (gb(f0), gb(tot)))
    t1 = time.perf_counter()
    dq = spec.device_reads(0, Q)
    dt_ = spec.device_reads(Q, T)
    out["data_gen_s"] = round(time.perf_counter() - t1, 2)
    out["query_bases"], out["target_bases"] = dq.total_bases, dt_.total_bases
    log("generated %.2f + %.2f Gbases in HBM in %.2f s; free %.1f GB" % (dq.total_bases / 1e9, dt_.total_bases / 1e9, out["data_gen_s"], gb(synth_cb.mem_info()[0])))
    if a.twin_check:
        n = min(a.twin_check, Q, T)
        for name, dr, first in (("queries", dq, 0), ("targets", dt_, Q)):
            for lo in (0, dr.n - n):
                h = spec.host_reads(first=first + lo, n=n)
                same = np.array_equal(h.offsets, dr.offsets[lo:lo + n + 1] - dr.offsets[lo]) and np.array_equal(h.bases, dr.to_host(lo, lo + n))
                out.setdefault("twins_identical", {})["%s[%d:%d]" % (name, lo, lo + n)] = bool(same)
        log("twins:", out["twins_identical"])
    ctx = engine.Context(0)
    qr, tr = dq.name_ranks(), dt_.name_ranks()
    t1 = time.perf_counter()
    Qd = ctx.upload(dq.ptr, dq.offsets, qr)
    Td = ctx.upload(dt_.ptr, dt_.offsets, tr)
    out["pack_s"] = round(time.perf_counter() - t1, 3)
    if not a.keep_ascii:
        dt_.free(); dq.free()
    log("packed in %.2f s; free %.1f GB" % (out["pack_s"], gb(synth_cb.mem_info()[0])))
    qlens, tlens = dq.lens(), dt_.lens()
    avg = np.float32(tlens.sum()) / np.float32(T)
    res = {}
    for strat in (["forward"] if a.forward else []) + (["inverse"] if a.inverse else []):
        best = None
        for _ in range(a.repeat):
            t1 = time.perf_counter()
            if strat == "inverse":
                ix = engine.Index(ctx, Qd, preset)
                tb = dict(ix.build_timings)
                log("inverse: index built (%.0f ms); free %.1f GB" % (tb["total"], gb(synth_cb.mem_info()[0])))
                counts = ix.overlap_inverse(Td)
                has = (counts > 0).astype(np.uint32)
            else:
                ix = engine.Index(ctx, Td, preset)
                tb = dict(ix.build_timings)
                log("forward: index built (%.0f ms); free %.1f GB" % (tb["total"], gb(synth_cb.mem_info()[0])))
                counts, has = ix.overlap_twoset(Qd)
            tm, cn, st = ctx.timings(), ctx.counters(), ix.stats()
            est = ctx.estimates(counts, qlens, float(avg), T, 100)
            med = engine.median(est, True, 0.15, 0.65)
            dt = time.perf_counter() - t1
            log("%s: step %.3f s (index %.0f ms, overlap %.0f ms); free %.1f GB" % (strat, dt, tb["total"], tm["total"], gb(synth_cb.mem_info()[0])))
            if best is None or dt < best[0]:
                best = (dt, tb, tm, cn, st, med, counts.copy(), has.copy())
            if _ + 1 < a.repeat:
                ix.free()
        dt, tb, tm, cn, st, med, counts, has = best
        r = dict(step_s=round(dt, 4), reads_per_s=round(Q / dt, 1), index_ms=round(tb["total"], 1), overlap_ms=round(tm["total"], 1),
                 mid_occ=st["mid_occ"], n_minimizers_index=st["n_minimizers"], n_keys_index=st["n_keys"], anchors=cn["anchors"], batches=cn["batches"],
                 groups=cn["groups"], groups_chained=cn["groups_chained"], no_mapping=int((has == 0).sum()),
                 counts_crc32=int(__import__("zlib").crc32(np.ascontiguousarray(counts).tobytes())), counts_sum=int(counts.astype(np.int64).sum()),
                 estimate=None if med[1] is None else float(med[1]), rel_err=None if med[1] is None else abs(float(med[1]) - spec.gsize) / spec.gsize,
                 q15_q65=[None if med[0] is None else float(med[0]), None if med[2] is None else float(med[2])],
                 stage_ms={**{"index_" + k: round(v, 1) for k, v in tb.items() if v and k != "total"}, **{k: round(v, 1) for k, v in tm.items() if v}})
        res[strat] = (counts, ix)
        out[strat] = r
        log(strat, json.dumps(r))
        # ---- oracle checks on samples (host twin -> oracle) ----
        if strat == "inverse" and a.check_targets:
            from oracle import oracle as O
            opt = O.make_opt(O.PRESET_AVA_PB if preset else O.PRESET_AVA_ONT, dual=True)
            n = min(a.check_targets, T)
            lo = (T - n) // 2
            t1 = time.perf_counter()
            hq = spec.host_reads(first=0, n=Q)
            ixo = O.Index(O.ReadSet(hq.seqs(), hq.names), opt)
            ht = spec.host_reads(first=Q + lo, n=n)
            rc, einv = ixo.inverse_counts(O.ReadSet(ht.seqs(), ht.names), threads=0)
            t_or = time.perf_counter() - t1
            # the same range on the device: a set of its own (device twin again), streamed against the same index
            dsub = spec.device_reads(Q + lo, n)
            Sd = ctx.upload(dsub.ptr, dsub.offsets, dsub.name_ranks())
            dsub.free()
            csub = ix.overlap_inverse(Sd)
            Sd.free()
            r["oracle_check"] = {"streamed_targets": [lo, lo + n], "counts_equal": bool(np.array_equal(einv, csub)), "mid_occ_equal": bool(ixo.mid_occ == st["mid_occ"]),
                                 "overlaps_in_sample": int(einv.sum()), "oracle_s": round(t_or, 1)}
            log("inverse oracle check:", r["oracle_check"])
        if strat == "forward" and a.check_queries:
            from oracle import oracle as O
            opt = O.make_opt(O.PRESET_AVA_PB if preset else O.PRESET_AVA_ONT, dual=True)
            t1 = time.perf_counter()
            ht = spec.host_reads(first=Q, n=T)
            ixo = O.Index(O.ReadSet(ht.seqs(), ht.names), opt)
            n = min(a.check_queries, Q)
            hq = spec.host_reads(first=0, n=n)
            rc, ec, eh = ixo.twoset_counts(O.ReadSet(hq.seqs(), hq.names), threads=0)
            r["oracle_check"] = {"queries": n, "counts_equal": bool(np.array_equal(ec, counts[:n])), "mid_occ_equal": bool(ixo.mid_occ == st["mid_occ"]),
                                 "n_keys_equal": bool(ixo.n_keys == st["n_keys"]) if hasattr(ixo, "n_keys") else None, "oracle_s": round(time.perf_counter() - t1, 1)}
            log("forward oracle check:", r["oracle_check"])
        ix.free()
    print(json.dumps(out))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
