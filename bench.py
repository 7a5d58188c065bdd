#!/usr/bin/env python
"""bench.py -- reads overlapped/sec of the liblrge overlap hot path on MI355X.

One "step" = one full pass of the hot path over one batch of synthetic reads that are already
resident (2-bit packed) in HBM: minimizer index build over the target set, sketch + seed + chain
of every query, distinct-target counts, per-read estimates, median.  Workload at N=1 is
BASELINE.json configs[1]: the 4.4 Mbp bacterial ONT set, two-set -Q 5000 -T 10000 (preset ava-ont,
which is what the reference CLI always runs: lrge/src/main.rs:56-85 never forwards -P).

N > 1 (launched by torch.distributed.run, one rank per GPU): the path shards by query read, so
every rank holds the full target index (built redundantly, no data-path collective) and its own
Q query reads ("weak": per-GPU work fixed); one RCCL all_gather over xGMI collects the per-read
estimate vectors.  value = reads all ranks processed / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2_bact_twoset")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the config (debug only; invalid as a result)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def cpu_baseline(q, t, budget_s):
    """The oracle ("port" of the liblrge/minimap2-2.30 path) timed on the host cores, on a bounded
    sample: the full target index is built once (timed), then as many query reads as fit in the
    budget are mapped with all cores; the index cost is pro-rated over the sampled fraction."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    opt = O.make_opt(O.PRESET_AVA_ONT, dual=True)
    T = O.ReadSet(t.seqs(), t.names)
    t0 = time.perf_counter()
    ix = O.Index(T, opt)
    t_index = time.perf_counter() - t0
    done, t_map, chunk = 0, 0.0, max(64, 8 * cores)
    counts = []
    while done < q.n and t_map < budget_s:
        hi = min(q.n, done + chunk)
        sub = q.slice(done, hi)
        Q = O.ReadSet(sub.seqs(), sub.names)
        t1 = time.perf_counter()
        rc, c, _ = ix.twoset_counts(Q, threads=cores)
        t_map += time.perf_counter() - t1
        assert rc == 0
        counts.append(c)
        done = hi
    frac = done / q.n
    reads_per_s = done / (t_map + t_index * frac)
    return dict(value=reads_per_s, unit="reads/s", cores=cores, kind="port",
                sample="first %d of %d query reads mapped on %d threads (%.1f s) against the full %d-read target index "
                       "(built in %.1f s: sketch + bucket sort on the same threads, scatter serial; pro-rated x%.3f)"
                       % (done, q.n, cores, t_map, t.n, t_index, frac),
                map_only_reads_per_s=done / t_map), np.concatenate(counts) if counts else np.zeros(0, np.uint32), ix.mid_occ


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
    import torch
    import torch.distributed as dist
    from lrge_amd import engine, synth

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("LRGE_BENCH_FORCE_DIST") == "1"   # the switch lets a 1-GPU box exercise RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # ---- synthetic inputs (untimed) ----
    cfg = synth.CONFIGS[a.config]
    assert cfg["mode"] == "twoset", "bench.py times the two-set forward path"
    gsize = int(cfg["genome"] * a.scale)
    Qn, Tn = max(1, int(cfg["Q"] * a.scale)), max(1, int(cfg["T"] * a.scale))
    genome = synth.random_genome(gsize, cfg["seed"])
    t = synth.sample_reads(genome, Tn, cfg["platform"], seed=cfg["seed"] + 1, name_prefix="t")
    # weak scaling: every rank draws its own Q query reads (rank 0's are the BASELINE set)
    q = synth.sample_reads(genome, Qn, cfg["platform"], seed=cfg["seed"] + 101 + 7 * rank, name_prefix="q%d_" % rank)

    ctx = engine.Context(local_rank)
    qr, tr = engine.name_ranks(q.names, t.names)
    Qd = ctx.upload(q.bases, q.offsets, qr)      # resident in HBM, 2-bit packed, before the timed region
    Td = ctx.upload(t.bases, t.offsets, tr)
    qlens = q.lens()
    avg_t = np.float32(t.lens().sum()) / np.float32(t.n)

    if use_dist:
        pin_in = torch.empty(Qn, dtype=torch.float32).pin_memory()
        pin_out = torch.empty(Qn * world, dtype=torch.float32).pin_memory()
        d_mine = torch.empty(Qn, dtype=torch.float32, device="cuda")
        d_all = torch.empty(Qn * world, dtype=torch.float32, device="cuda")

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        if not os.environ.get("LRGE_BENCH_NO_PRESKETCH"):
            Qd.presketch(0)     # the queries are sketched beside the index's sort / table passes (still once per step)
        ix = engine.Index(ctx, Td, 0)
        tb = dict(ix.build_timings); cb_ = dict(ix.build_counters)
        counts, has = ix.overlap_twoset(Qd)
        tm = ctx.timings(); cn = ctx.counters(); st = ix.stats()
        est = ctx.estimates(counts, qlens, float(avg_t), t.n, 100)
        ix.free()
        if use_dist:    # the one collective of the path: per-read estimate vectors over RCCL/xGMI
            pin_in.numpy()[:] = est                       # pinned staging: the copies queue up behind each other,
            d_mine.copy_(pin_in, non_blocking=True)       # one host wait per step
            dist.all_gather_into_tensor(d_all, d_mine)
            pin_out.copy_(d_all, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            est_all = pin_out.numpy().copy()
        else:
            est_all = est
        med = engine.median(est_all, True, 0.15, 0.65)
        for k_ in ("rs_scatter_launches", "rs_scatter_items", "rs_scatter_bytes"):   # the index build sorts too
            cn[k_] = cn.get(k_, 0) + cb_.get(k_, 0)
        return counts, est_all, med, tb, tm, cn, st

    for _ in range(a.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    acc_tb, acc_tm, acc_cn = {}, {}, {}
    for _ in range(a.steps):
        counts, est_all, med, tb, tm, cn, st = step()
        for k, v in tb.items(): acc_tb[k] = acc_tb.get(k, 0.0) + v
        for k, v in tm.items(): acc_tm[k] = acc_tm.get(k, 0.0) + v
        for k, v in cn.items(): acc_cn[k] = v if k == "lpg_split" else acc_cn.get(k, 0) + v
    sync_all()
    elapsed = time.perf_counter() - t0
    # one instrumented step AFTER the timed region: an event pair around every k_rs_scatter launch (timer level 2 costs
    # ~2 % of a step in host work between launches, so the timed steps run at the default level)
    ctx.set_timer_level(2)
    _, _, _, tb2, tm2, cn2, _ = step()
    ctx.set_timer_level(1)
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        K = a.steps
        ms_per_step = elapsed * 1e3 / K
        value = world * Qn * K / elapsed
        # ---- roofline of the dominant kernel against the HBM roof ----
        # The kernel with the most time per step is k_chain_lpg (mg_lchain_dp + backtrack, 64 groups per
        # wavefront).  Algorithmic bytes per launch = 16 B per anchor it chains (8 B key + 8 B value; SURVEY.md
        # 8(d): the "16*H anchor in for chaining" term of B_q restricted to what the launch covers).  Its avg
        # launch duration is measured live with a HIP event pair on the side stream the kernel runs on.
        # It is an integer DP bound by VALU issue, not by memory, so frac is small by construction; the
        # memory-bound kernel with the most time (k_rs_scatter: 32 B per (key, value) pair, 16 B per packed key, 24 B in the unpacking pass) is reported next to it.
        pmc = {}
        pj = os.path.join(ROOT, "profiles", "chain_pmc.json")
        if os.path.exists(pj):
            try:
                pmc = json.load(open(pj))
            except Exception:
                pmc = {}

        def roof(kernel, ms_total, launches, bytes_total, traffic_key):
            launches = max(1, launches)
            avg_ms = ms_total / launches
            alg = bytes_total / launches
            ach = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBPS, "traffic": pmc.get(traffic_key), "alg_bytes_per_launch": alg,
                    "avg_launch_ms": avg_ms, "launches_per_step": launches / K}

        mode = os.environ.get("LRGE_HIP_CHAIN", "")
        if acc_cn.get("lpg_launches", 0):
            r_dom = roof("k_chain_lpg", acc_tm.get("chain_lpg", 0.0), acc_cn["lpg_launches"], 16.0 * acc_cn.get("lpg_anchors", 0),
                         "k_chain_lpg_hbm_bytes_per_launch")
        else:   # a single-kernel mode was forced (LRGE_HIP_CHAIN=hw|reg|lds)
            r_dom = roof("k_chain_" + (mode or "hw"), acc_tm.get("chain", 0.0), acc_cn.get("chain_launches", 0),
                         16.0 * acc_cn.get("chain_anchors", 0), "k_chain_%s_hbm_bytes_per_launch" % (mode or "hw"))
        r_stage = roof("chain stage: k_chain_hw beside k_chain_lpg (fork..join)", acc_tm.get("chain", 0.0),
                       acc_cn.get("batches", 0), 16.0 * acc_cn.get("chain_anchors", 0), "chain_stage_hbm_bytes_per_step")
        r_sc = roof("k_rs_scatter", tm2.get("rs_scatter", 0.0) + tb2.get("rs_scatter", 0.0), cn2.get("rs_scatter_launches", 0),
                    float(cn2.get("rs_scatter_bytes", 0)), "k_rs_scatter_hbm_bytes_per_launch")
        r_sc["launches_per_step"] = float(cn2.get("rs_scatter_launches", 0))
        r_sc["measured"] = "one instrumented step after the timed region (event pair around every launch)"
        # What actually bounds the chain stage: VALU issue.  A wave64 VALU instruction occupies its SIMD for 4 cycles
        # (MI355X_MICROARCH.md); the instruction counts per launch come from the SQ counter pass in profiles/
        # (SQ_INSTS_VALU, same command), the stage time is the live fork..join figure of this run.
        issue = None
        vi = [pmc.get("k_chain_lpg_valu_insts_per_launch"), pmc.get("k_chain_hw_valu_insts_per_launch")]
        if all(vi) and acc_cn.get("batches", 0) and acc_tm.get("chain", 0.0) > 0:
            n_simd, clk = 256 * 4, 2.4e9
            issue_ms = sum(vi) * 4.0 / (n_simd * clk) * 1e3
            stage_ms = acc_tm["chain"] / acc_cn["batches"]
            issue = {"bound": "valu_issue", "valu_insts_per_step": sum(vi), "cycles_per_inst": 4, "simds": n_simd, "clock_ghz": 2.4,
                     "issue_ms": issue_ms, "stage_ms": stage_ms, "frac": issue_ms / stage_ms,
                     "note": "chain stage = k_chain_lpg + k_chain_hw side by side; integer DP, not a memory stream"}
        # whole-path algorithmic bytes: B_q summed over queries + B_idx, per step
        L = float(q.lens().sum()); M = acc_cn.get("query_minimizers", 0) / K; H = acc_cn.get("anchors", 0) / K
        B_q = L / 4 + 32 * M + 8 * H + 32 * H + 4 * Qn
        B_idx = float(t.lens().sum()) / 4 + 16 * st["n_minimizers"]
        e2e_gbps = (B_q + B_idx) / (ms_per_step * 1e-3) / 1e9
        out = {
            "metric": "reads overlapped/sec (whole node)", "value": value, "unit": "reads/s",
            "n_gpus": world, "steps": K, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 keys / i32 chain scores / f32 gap penalty", "data": "synthetic",
            "config": {"workload": "%s: %.1f Mbp genome, ONT reads, two-set forward -Q %d -T %d per GPU, preset ava-ont, dual=yes"
                                   % (a.config, gsize / 1e6, Qn, Tn),
                       "query_reads_per_gpu": Qn, "target_reads": Tn, "parallelism": "query-sharded x%d, index replicated" % world,
                       "scale": a.scale, "presketch_hint": not os.environ.get("LRGE_BENCH_NO_PRESKETCH")},
            "genome_size_true": gsize,
            "genome_size_estimate": None if med[1] is None else float(med[1]),
            "genome_size_abs_error": None if med[1] is None else abs(float(med[1]) - gsize),
            "estimate_q15_q65": [None if med[0] is None else float(med[0]), None if med[2] is None else float(med[2])],
            "mid_occ": st["mid_occ"],
            "roofline": {**r_dom, "whole_path_alg_GBps": e2e_gbps, "whole_path_frac": e2e_gbps / HBM_PEAK_GBPS},
            "roofline_other": [r_stage, r_sc],
            "chain_stage_issue": issue,
            "stage_ms_per_step": {**{"index_" + k: v / K for k, v in acc_tb.items() if v and k != "total"},
                                  **{k: v / K for k, v in acc_tm.items() if v}},
            "work_per_step": {k: (v if k == "lpg_split" else v / K) for k, v in acc_cn.items()},
        }
        if world == 1 and not a.no_cpu_baseline:
            cb, ccounts, cmid = cpu_baseline(q, t, a.cpu_seconds)
            out["cpu_baseline"] = cb
            out["gpu_vs_cpu"] = value / cb["value"]
            n = len(ccounts)
            out["parity_vs_oracle_sample"] = {"reads": n, "counts_equal": bool(np.array_equal(ccounts, counts[:n])),
                                              "mid_occ_equal": bool(cmid == st["mid_occ"])}
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
