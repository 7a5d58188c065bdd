"""benchlib (beside bench.py, NOT part of the product package lrge_amd: it imports the oracle) -- the legs of bench.py that are not the timed step: the CPU baseline (the oracle port timed on the host cores, test
infrastructure used as the reported baseline only), the committed rocprofv3 counter figures the roofline block quotes, and the
world-of-N emulation on one GPU (every rank timed alone: a projection input, never a bench result).  bench.py keeps the argument
parser, the job of one rank and the timed region."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))


def cpu_baseline(q, t, budget_s, preset):
    """The oracle ("port" of the liblrge/minimap2-2.30 path, NOT the reference binary) timed on the host cores, on a
    bounded sample: the target index is built over the first `frac_t` of the target reads that fit ~budget seconds
    (index time scales linearly in target bases and is pro-rated to the full set), then as many query reads as fit in the
    budget are mapped with all cores against it."""
    from oracle import oracle as O
    cores = O.default_threads()       # (threads actually used: the CPUs the host grants, oracle.host_cpus)
    opt = O.make_opt(O.PRESET_AVA_PB if preset else O.PRESET_AVA_ONT, dual=True)
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
    return dict(value=reads_per_s, unit="reads/s", cores=cores, kind="port", host_cpus_granted=O.host_cpus(), host_hw_threads=os.cpu_count(),
                sample="first %d of %d query reads mapped on %d threads (%.1f s) against the full %d-read target index "
                       "(built in %.1f s on the same threads: sketch + bucket sort parallel, scatter serial; pro-rated x%.3f)"
                       % (done, q.n, cores, t_map, t.n, t_index, frac),
                map_only_reads_per_s=done / t_map, index_s=t_index), \
        (np.concatenate(counts) if counts else np.zeros(0, np.uint32)), ix.mid_occ


PROFILE_ROUND = "r06"


def _committed(config, inverse, preset_name=None):
    """profiles/r05_hbm_traffic[_<config>][_ont].json (else round 4's): made by tools/summarize_profiles.py from the FETCH_SIZE / WRITE_SIZE passes of
    this same command (tools/profile_round.sh).  A collection counts only if it is of this configuration, strategy AND preset."""
    names = []
    tail = "_inverse" if inverse else ""
    for rnd in (PROFILE_ROUND, "r05", "r04"):       # (the newest committed collection of this configuration; the file's name is reported with the figures)
        names += ["%s_hbm_traffic_%s%s_%s.json" % (rnd, config, tail, (preset_name or "").replace("ava-", "")), "%s_hbm_traffic_%s%s.json" % (rnd, config, tail),
                  "%s_hbm_traffic.json" % rnd]
    for name in names:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            if d.get("config") == config and bool(d.get("inverse", False)) == bool(inverse) and (preset_name is None or ("preset %s" % preset_name) in (d.get("workload") or "")):
                return d, name
        except Exception:      # noqa: BLE001
            pass
    return None, None


def committed_traffic(config, inverse=False, preset_name=None):
    """HBM bytes per step of the whole path from the round's committed rocprofv3 --pmc passes, next to the algorithmic bytes:
    the counters cannot be read inside a timed run."""
    d, name = _committed(config, inverse, preset_name)
    if d is None:
        return None
    out = {k: d[k] for k in ("fetch_GB_per_step", "write_GB_per_step", "algorithmic_GB_per_step", "traffic_over_algorithmic", "source") if k in d}
    out["file"] = "profiles/" + name
    return out


def committed_kernel_traffic(config, inverse, kernel, preset_name=None):
    """(HBM bytes per launch of `kernel` [2 x FETCH_SIZE + WRITE_SIZE, averaged over its launches and template instantiations
    in the committed pass], detail) or (None, None)."""
    d, name = _committed(config, inverse, preset_name)
    if d is None:
        return None, None
    n = f = w = 0.0
    for k, v in d.get("kernels", {}).items():
        if k == kernel or k.startswith(kernel + "<"):
            ln = float(v.get("launches_in_pass", 0))
            n += ln; f += ln * v["fetch_raw_bytes_per_launch"]; w += ln * v["write_raw_bytes_per_launch"]
    if not n:
        return None, None
    return (2.0 * f + w) / n, {"fetch_raw_bytes_per_launch": f / n, "write_raw_bytes_per_launch": w / n, "launches_in_pass": n,
                               "correction": "reads x2 (gfx950 tallies 128-B read requests at 64 B), writes x1", "file": "profiles/" + name}


def committed_kernel_valu(config, inverse, kernel, preset_name=None):
    """wave64 VALU instructions per launch of `kernel` (SQ_INSTS_VALU of the committed SQ pass, averaged over its launches and template
    instantiations) or None."""
    d, name = _committed(config, inverse, preset_name)
    if d is None:
        return None
    n = v = 0.0
    for k, e in d.get("kernels", {}).items():
        if (k == kernel or k.startswith(kernel + "<")) and "valu_insts_per_launch" in e:
            ln = float(e.get("launches_in_pass", 0))
            n += ln; v += ln * e["valu_insts_per_launch"]
    return None if not n else {"valu_wave_insts_per_launch": v / n, "launches_in_pass": n, "file": "profiles/" + name}


VALU_PEAK_WAVE_INSTS_PER_S_PER_CU = 2.4e9     # 4 SIMDs x one wave64 VALU instruction every 4 cycles at 2.4 GHz (MI355X_MICROARCH.md; tools/micro/valu_rate.hip measures 2.4-2.7 cycles for the cheap ones)


def cpu_baseline_sampled(spec, Qn, Tn, budget_s, preset):
    """The same leg for a job whose target set the host cannot index inside a bench run (H. sapiens-scale: 30 Gbases):
    the oracle indexes 1/F of the target reads (host twin of the generator) and maps query reads against that index
    for ~budget seconds.  Pro-rated to the whole job as  time = F x index time + F x map time per read x reads  --
    the seed hits, anchors and chains of a query grow with the number of target reads, its own sketch (a small part)
    does not, so this slightly overstates the CPU's time per read and is labelled as what it is: a sample."""
    from oracle import oracle as O
    cores = O.default_threads()       # (threads actually used: the CPUs the host grants, oracle.host_cpus)
    F = 40
    nt = max(1000, Tn // F)
    opt = O.make_opt(O.PRESET_AVA_PB if preset else O.PRESET_AVA_ONT, dual=True)
    t = spec.host_reads(first=Qn, n=nt)
    t0 = time.perf_counter()
    ix = O.Index(O.ReadSet(t.seqs(), t.names), opt)
    t_index = time.perf_counter() - t0
    done, t_map, chunk = 0, 0.0, max(64, 8 * cores)
    while done < Qn and t_map < budget_s:
        hi = min(Qn, done + chunk)
        sub = spec.host_reads(first=done, n=hi - done)
        Q = O.ReadSet(sub.seqs(), sub.names)
        t1 = time.perf_counter()
        rc, c, _ = ix.twoset_counts(Q, threads=cores)
        t_map += time.perf_counter() - t1
        assert rc == 0
        done = hi
    scale = Tn / nt
    t_job = scale * t_index + scale * (t_map / done) * Qn
    return dict(value=Qn / t_job, unit="reads/s", cores=cores, kind="port", host_cpus_granted=O.host_cpus(), host_hw_threads=os.cpu_count(),
                sample="SAMPLE of the job: oracle index over %d of the %d target reads (%.1f s), %d query reads mapped against it on %d "
                       "threads (%.1f s); pro-rated x%.1f in the target dimension (index time and per-read map time both scale with "
                       "the number of target reads) and to all %d queries" % (nt, Tn, t_index, done, cores, t_map, scale, Qn),
                index_s_sample=t_index, map_s_sample=t_map, reads_mapped=done,
                measured_only={"what": "what was actually timed, nothing pro-rated: %d query reads mapped against an index of %d target reads"
                                       % (done, nt), "reads_per_s_map_only": done / t_map, "index_s": t_index, "map_s": t_map},
                pro_rating={"target_factor": scale, "assumption": "index time and per-read map time linear in the number of target reads "
                                                                  "(the per-read sketch is not: the estimate slightly overstates the CPU's time per read)"})


XGMI_LINK_GBPS = 64.0     # one direction of one xGMI link, what a point-to-point transfer between two GPUs sustains (7 links per GPU)


def emulate_world(a, ctx0, RankJob, Qn, engine, parallel, device, q_lens_all, t_lens_all, forward_mode="tshard"):
    """--emulate-world N (see parse()).  Prints one JSON line."""
    import threading
    N = a.emulate_world
    # the one-GPU job first: the reference time and the reference results
    # clock: as `value` -- the ASCII reads in pinned host memory (a rank's host-side pack and PCIe transfer are part of its busy time),
    # or resident in HBM with --clock resident
    from_host = a.clock == "host"

    def sources(job):
        """-> (src_q, src_t, release): what job.step takes"""
        if not from_host:
            return job.qs.ptr, job.ts.ptr, (lambda: None)
        hq = job.ctx.host_alloc(max(job.qs.nbytes, 1)); ht = job.ctx.host_alloc(max(job.ts.nbytes, 1))
        hq.array[:job.qs.nbytes] = job.qs.host(); ht.array[:job.ts.nbytes] = job.ts.host()
        if hasattr(job.qs, "dev"):          # (counter-based generator: the resident copies make room)
            job.qs.dev.free(); job.ts.dev.free()
        return hq, ht, (lambda: (hq.free(), ht.free()))
    one = RankJob(ctx0, None, 0, 1, device)
    sq, st_, rel = sources(one)
    for _ in range(max(1, a.warmup)):
        one.step(sq, st_)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ref = one.step(sq, st_)
    t_one = (time.perf_counter() - t0) * 1e3 / a.steps
    ref_counts, ref_est, ref_med, _, _, _, ref_st = ref
    rel()
    if hasattr(one.qs, "dev") and one.qs.dev.ptr:
        one.qs.dev.free(); one.ts.dev.free()
    one.qs = one.ts = None
    # the one-GPU job's context goes now: its arena holds most of the HBM (segments are kept until the context dies), and the ranks'
    # contexts need that room
    del one
    ctx0.close()

    def free_gb():
        try:
            from lrge_amd import synth_cb
            f, t = synth_cb.mem_info(device)
            return "%.1f of %.1f GB free" % (f / 1e9, t / 1e9)
        except Exception as e:      # noqa: BLE001
            return "mem_info: %r" % e
    sys.stderr.write("[emulate-world] reference context closed: %s\n" % free_gb())
    grp = parallel.LocalGroup(N)
    big = float(q_lens_all.sum() + t_lens_all.sum()) > 5e9
    if big:     # a presketch keeps worst-case buffers (16 B per streamed base) across the build's collectives: N of them do not fit one GPU
        os.environ["LRGE_BENCH_NO_PRESKETCH"] = "1"
    grp.serialize(2 if big else True)      # big jobs: N arenas sized for a GPU each do not fit one GPU -- idle segments go back between turns (allocator time is kept out of busy_ms)
    res, errs = [None] * N, []

    def rank_main(r):
        try:
            c = engine.Context(device)
            if big:
                c.set_option("POOL_SEG_MAX_MB", "256")      # (every large array gets a segment of its own: what a waiting rank keeps is what it uses)
                c.set_option("HT_SLOTS_X100", "125")        # (N resident tables on one GPU: the load a part of a partitioned index runs at)
                # a rank on a GPU of its own takes its share's anchors in ONE batch (~1.4 G seed hits of a 288-GB device); here the planner
                # sees what the other N - 1 ranks leave free and cut some ranks' shares into 2-5 batches, each paying the chain stage's
                # longest-group latency again -- the "moving outlier" of round 4 (batches 4-5 -> chain 47-50 ms instead of 13: profiles/
                # r05_emulated_world8_fwd_resident_runs.txt).  The batch size a GPU of its own would pick is pinned; a batch that really
                # does not fit is still retried in halves (host_overlap_batch.inl)
                c.set_option("BATCH_ANCHORS", str(1 << 31))
            comm = grp.comm(c, r)
            comm.turn(True)
            job = RankJob(c, comm, r, N, device)
            src_q, src_t, release = sources(job)
            sys.stderr.write("[emulate-world] rank %d holds its reads: %s\n" % (r, free_gb()))
            comm.turn(False)
            for it in range(max(1, a.warmup) + a.steps):
                if it == max(1, a.warmup):
                    comm.busy_ms(reset=True); comm.standin_ms(reset=True)
                comm.turn(True)
                out = job.step(src_q, src_t)
                comm.turn(False)
            busy = comm.busy_ms() / a.steps
            standin = comm.standin_ms() / a.steps
            counts, est_all, med, tb, tm, cn, st = out
            lo, hi = job.bounds[r], job.bounds[r + 1]
            mine = ref_counts if (a.inverse or job.tshard) else ref_counts[lo:hi]       # (inverse / target-sharded forward: the all-reduced vector)
            ok = bool(np.array_equal(counts, mine) and st == ref_st and np.array_equal(est_all.view(np.uint32), ref_est.view(np.uint32)))
            detail = None
            if not ok:
                nd = int((counts != mine).sum()) if len(counts) == len(mine) else -1
                detail = dict(counts_differ=nd, first=[int(x) for x in np.nonzero(counts != mine)[0][:8]] if nd > 0 else [], stats=st, ref_stats=ref_st,
                              estimates_equal=bool(np.array_equal(est_all.view(np.uint32), ref_est.view(np.uint32))))
            res[r] = dict(rank=r, busy_ms_per_step=busy, standin_copy_ms_per_step=standin, results_equal_one_gpu=ok, mismatch=detail, streamed_reads_of_rank=hi - lo, shard=job.shard_stats,
                          work_last_step={k: int(cn.get(k, 0)) for k in ("batches", "lpg_split", "lpg_launches", "lpg_anchors", "chain_launches", "chain_anchors", "chain_glb_anchors",
                                                                         "groups_chained", "anchors", "anchors_kept", "query_minimizers", "index_parts")},
                          stage_ms={**{"index_" + k: round(v, 3) for k, v in tb.items() if v and k != "total"}, **{k: round(v, 3) for k, v in tm.items() if v}})
            release()
            comm.close(); c.close()
        except Exception as e:      # noqa: BLE001 -- reported below
            errs.append((r, repr(e)))
            sys.stderr.write("[emulate-world] rank %d failed (%r): %s\n" % (r, e, free_gb()))
            try:
                comm.turn(False)    # a rank that fails while it holds the GPU must hand it back, or the others wait for ever
            except Exception:       # noqa: BLE001
                pass
    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(N)]
    for t_ in th:
        t_.start()
    deadline = time.perf_counter() + float(os.environ.get("LRGE_BENCH_EMULATE_TIMEOUT", "240"))
    t_err = None
    while any(t_.is_alive() for t_ in th) and time.perf_counter() < deadline:
        for t_ in th:
            t_.join(timeout=0.5)
        if errs and t_err is None:
            t_err = time.perf_counter()
        if t_err is not None and time.perf_counter() - t_err > 15.0:      # a rank has failed: the others may be waiting for it for ever
            break
    if errs or any(t_.is_alive() for t_ in th):
        print(json.dumps({"emulate_world": N, "errors": errs, "ranks_still_running": sum(t_.is_alive() for t_ in th)}))
        sys.stdout.flush()
        os._exit(1)
    grp.close()
    busy = max(r_["busy_ms_per_step"] for r_ in res)
    # (the variable-size all-gather of the local transport copies device to device where a node's links deliver straight into HBM: that
    # time is inside busy_ms AND the link model charges the same bytes, so the link-model projection takes it out first)
    busy_net = max(r_["busy_ms_per_step"] - r_["standin_copy_ms_per_step"] for r_ in res)
    # link model: the all-gather of the key sets as a ring (per-link bound), the two all-to-alls point to point (every peer
    # has its own link; the slowest pair bounds the exchange)
    ss = [r_["shard"] for r_ in res if r_["shard"]]
    link_ms = None
    if ss:
        eb = ss[0]["entry_bytes"]
        ring = (N - 1) * ss[0]["keyset_bytes"] / (XGMI_LINK_GBPS * 1e9) * 1e3
        if forward_mode == "tshard" and not a.inverse:      # + the closing all-reduce of the two u32[Q] vectors (ring)
            ring += 2.0 * (N - 1) / N * 8.0 * Qn / (XGMI_LINK_GBPS * 1e9) * 1e3
        a2a = max((x["entries_recv"] * eb + x["hashes_recv"] * x.get("hash_bytes", 8)) / max(N - 1, 1) for x in ss) / (XGMI_LINK_GBPS * 1e9) * 1e3
        link_ms = {"keyset_allgather_ring_ms": ring, "alltoall_ms": a2a, "per_link_GBps": XGMI_LINK_GBPS,
                   "note": "model: D2D copies stand in for the links inside busy_ms (those run at HBM speed); this is what the links add at best-case even spreading"}
    if not ss and a.inverse:      # inverse: one all-reduce of the u32 count vector over the indexed (query) reads closes the step
        vec = 4.0 * Qn
        link_ms = {"count_allreduce_ring_ms": 2.0 * (N - 1) / N * vec / (XGMI_LINK_GBPS * 1e9) * 1e3, "per_link_GBps": XGMI_LINK_GBPS,
                   "keyset_allgather_ring_ms": 0.0, "alltoall_ms": 0.0, "note": "one ring all-reduce of %d bytes; latency-bound in practice (~0.1 ms)" % int(vec)}
    print(json.dumps({"emulate_world": N, "NOT_A_BENCH_RESULT": "all ranks on one GPU, taking turns; a projection input", "config": a.config,
                      "clock": "host (ASCII reads in pinned host memory: a rank's host-side pack and PCIe transfer are inside its busy time)" if from_host else "resident",
                      "strategy": "inverse (--use-min-ref: index = queries, replicated; streamed targets cut by bases)" if a.inverse else
                                  ("forward, TARGETS sharded (lrge_hip_index_build_tsharded): every rank maps all queries against its share, counts all-reduced" if forward_mode == "tshard"
                                   else "forward, queries sharded, the WHOLE index built by every rank (north_star's literal form: replicated index, one gather)" if forward_mode == "replicated"
                                   else "forward, queries sharded (target sketch sharded, restricted index per rank)"),
                      "one_gpu_ms_per_step": t_one, "max_rank_busy_ms_per_step": busy,
                      "projected_speedup_compute_only": t_one / busy,
                      "max_rank_busy_minus_standin_copies_ms_per_step": busy_net,
                      "projected_speedup_with_link_model": None if not link_ms else t_one / (busy_net + link_ms["keyset_allgather_ring_ms"] + link_ms["alltoall_ms"] + link_ms.get("count_allreduce_ring_ms", 0.0)),
                      "link_model": link_ms, "all_ranks_equal_one_gpu": all(r_["results_equal_one_gpu"] for r_ in res),
                      "exchange_bytes_total": None if not ss else {"keysets": N * (N - 1) * ss[0]["keyset_bytes"],
                                                                   "entries": sum(x["entries_sent"] for x in ss) * ss[0]["entry_bytes"],
                                                                   "hashes": sum(x["hashes_sent"] * x.get("hash_bytes", 8) for x in ss)},
                      "ranks": res}))


HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def roofline_blocks(a, preset, world, K, ms_per_step, acc_tb, acc_tm, acc_cn, tb2, tm2, cn2, lvl2_all, st, Qn, Tn, q_lens, t_lens):
    """The roofline bookkeeping of a bench line from the timed steps' HIP-event times and work counters:
    -> (roofline = the SURVEY 8(d) whole-path block, the single kernels by time per step, the kernel families, the figures quoted from
    committed profiles).  Algorithmic bytes follow SURVEY.md 8(d); see the comments inside."""
    def roof(kernel, ms_total, launches, bytes_total, bytes_note, bound="hbm"):
        # bound = "valu": the kernel is bound by the VALU issue rate (the committed SQ pass says how close it runs to it:
        # from_committed_profiles.valu_issue); its HBM figures are still SURVEY 8(d)'s bytes over its time, for reference
        launches = max(1, launches)
        avg_ms = ms_total / launches
        alg = bytes_total / launches
        ach = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        return {"bound": bound, "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBPS, "traffic": None, "alg_bytes_per_launch": alg, "alg_bytes": bytes_note,
                "avg_launch_ms": avg_ms, "launches_per_step": launches / K, "ms_per_step": ms_total / K}

    # Candidates, each timed with HIP event pairs on the stream it runs on during the TIMED steps.  `roofline` is the SINGLE
    # KERNEL with the most time per step; kernel families (all kernels of a sort, of the sketch) are
    # reported beside it in `roofline_other` with kind = "family".  Algorithmic bytes per launch follow SURVEY.md 8(d):
    #   k_lookup           16 B per query minimizer (one hash/offset entry per lookup)
    #   k_chain_lpg        16 B per anchor it chains (8 B key + 8 B value in)
    #   k_expand           8 B position-list entry in + 8 B anchor out per anchor
    #   k_rs_scatter       8(d) counts the ORDERING of index entries and anchors as implementation overhead (zero algorithmic
    #                      bytes); the kindest honest denominator is what any sort must move -- every index entry and every
    #                      anchor read once and written once per STEP -- dealt over the launches that do the moving: a sort of
    #                      P passes can reach at most 1/P of the roof on this scale.  (`streamed`: the bytes the launches
    #                      actually read + write, i.e. the kernel as a streaming kernel.)
    #   families           index radix sort (k_rs_hist + scan + k_rs_scatter, all passes): one read + one write of every entry;
    #                      index sketch (k_sketch_direct + k_sketch_compact): L/4 B of packed bases in + one entry out
    # traffic: HBM bytes per launch of that kernel from the round's committed rocprofv3 --pmc passes of this same command
    # (profiles/r05_hbm_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950); the counters
    # cannot be read inside a timed run.
    cands, fams = [], []
    if acc_cn.get("lookup_launches", 0):
        cands.append(roof("k_lookup", acc_tm.get("k_lookup", 0.0), acc_cn["lookup_launches"],
                          16.0 * acc_cn.get("query_minimizers", 0), "16 B x query minimizers (SURVEY 8d: one hash/offset entry per lookup)"))
    if acc_cn.get("lpg_launches", 0):
        cands.append(roof("k_chain_lpg", acc_tm.get("chain_lpg", 0.0), acc_cn["lpg_launches"], 16.0 * acc_cn.get("lpg_anchors", 0),
                          "16 B x anchors chained by the launch (SURVEY 8d: anchor in for chaining)", bound="valu"))
    n_sk = acc_cn.get("sketch_launches", 0) + acc_cn.get("sketch_wave_launches", 0)
    if n_sk and acc_tb.get("k_sketch", 0.0) > 0 and world == 1:
        # the one-pass index sketch kernel (k_sketch_wave: the wave-dense form, round 6; k_sketch_direct where that does not apply), event pairs
        # around its launches (timer level 2): SURVEY 8(d)'s B_idx = L_T / 4 + 16 M_T dealt over them
        L_idx_ = float((q_lens if a.inverse else t_lens).sum())
        cands.append(roof("k_sketch_wave" if acc_cn.get("sketch_wave_launches", 0) >= acc_cn.get("sketch_launches", 0) else "k_sketch_direct", acc_tb["k_sketch"], n_sk, (L_idx_ / 4.0 + 16.0 * float(st["n_minimizers"])) * K,
                          "SURVEY 8d's B_idx: L_T / 4 B of packed bases in + 16 B per index minimizer out, dealt over the launches", bound="valu"))
    n_idx = float(st["n_minimizers"]) if not (world > 1 and not a.inverse) else float(acc_cn.get("rs_scatter_items", 0)) / max(1, K) / 4.0
    entry_b = 8.0 if (2 * (19 if preset else 15) + int(np.ceil(np.log2((Qn if a.inverse else Tn) + 1))) + int(np.ceil(np.log2(float((q_lens if a.inverse else t_lens).max()) + 1))) + 1) <= 64 else 16.0
    if acc_tb.get("index_sort", 0.0) > 0 and world == 1:
        fams.append(roof("index radix sort (k_rs_hist + scan + k_rs_scatter, all passes)", acc_tb["index_sort"], K, 2.0 * entry_b * n_idx * K,
                         "%d B in + %d B out per index entry: one read and one write of every entry" % (entry_b, entry_b)))
    if acc_tb.get("sketch", 0.0) > 0 and world == 1:
        L_idx = float((q_lens if a.inverse else t_lens).sum())
        fams.append(roof("index sketch (K1: k_sketch_wave, or k_sketch_direct + k_sketch_compact)", acc_tb["sketch"], K, (L_idx / 4.0 + entry_b * n_idx) * K,
                         "L/4 B of packed bases in + %d B per minimizer out" % entry_b))
    if acc_tm.get("anchor_sort", 0.0) > 0:
        fams.append(roof("anchor sort (k_seg_sort_local / k_rs_hist + scan + k_rs_scatter)", acc_tm["anchor_sort"], K, 24.0 * acc_cn.get("anchors_kept", acc_cn.get("anchors", 0)),
                         "8 B packed anchor in + 16 B (key, value) out per anchor that left the expansion"))
    if acc_tm.get("expand", 0.0) > 0:
        cands.append(roof("k_expand_q", acc_tm["expand"], max(1, acc_cn.get("batches", K)), 16.0 * acc_cn.get("anchors", 0),
                          "8 B position-list entry in + 8 B anchor out per seed hit (SURVEY 8d); one `launch` = the <= 3 size-class launches of a batch, timed together"))
    # k_rs_scatter: in the timed steps when they run at timer level 2 (big jobs), else in the instrumented step behind them
    if lvl2_all:
        sc_ms, sc_n, sc_bytes, sc_how = acc_tm.get("rs_scatter", 0.0) + acc_tb.get("rs_scatter", 0.0), acc_cn.get("rs_scatter_launches", 0), float(acc_cn.get("rs_scatter_bytes", 0)), "event pair around every launch of the timed steps"
        anchors_moved = float(acc_cn.get("anchors_kept", acc_cn.get("anchors", 0)))      # (what leaves the expansion: the sort moves those)
    else:
        sc_ms, sc_n, sc_bytes, sc_how = (tm2.get("rs_scatter", 0.0) + tb2.get("rs_scatter", 0.0)) * K, cn2.get("rs_scatter_launches", 0) * K, float(cn2.get("rs_scatter_bytes", 0)) * K, "one instrumented step after the timed region (event pair around every launch)"
        anchors_moved = float(cn2.get("anchors_kept", cn2.get("anchors", 0))) * K
    if sc_n:
        r_sc = roof("k_rs_scatter", sc_ms, sc_n, 2.0 * entry_b * n_idx * K + 16.0 * anchors_moved,
                    "what any sort must move, once per step: %d B in + %d B out per index entry, 8 B in + 8 B out per anchor -- dealt over this "
                    "kernel's launches (SURVEY 8d counts ordering as zero algorithmic bytes)" % (entry_b, entry_b))
        r_sc["measured"] = sc_how
        r_sc["denominator_is_not_8d"] = True        # (SURVEY 8(d) gives ordering ZERO algorithmic bytes: this block never leads the line)
        r_sc["streamed"] = {"bytes_per_launch": sc_bytes / sc_n, "GBps": sc_bytes / (sc_ms * 1e-3) / 1e9 if sc_ms > 0 else 0.0,
                            "what": "bytes the launches read + write (32 / pair, 16 / packed key, 24 unpacking): the kernel as a streaming kernel"}
        cands.append(r_sc)
    for r_ in cands: r_["kind"] = "kernel"
    for r_ in fams: r_["kind"] = "family"
    cands.sort(key=lambda r: -r["ms_per_step"])
    # figures from the round's committed rocprofv3 counter passes (they cannot be read inside a timed run) are NOT put into the blocks: a
    # line timed by the driver must not change because a committed file changed.  They are returned apart (`committed`), keyed by kernel,
    # and bench.py prints them under ONE key, from_committed_profiles
    committed = {"kernel_traffic": {}, "valu_issue": {}}
    pn = "ava-pb" if preset else "ava-ont"
    for r_ in cands:
        tr, det = committed_kernel_traffic(a.config, a.inverse, r_["kernel"], pn)
        if tr is not None:
            committed["kernel_traffic"][r_["kernel"]] = {"hbm_bytes_per_launch": tr, "over_algorithmic": tr / r_["alg_bytes_per_launch"] if r_["alg_bytes_per_launch"] else None, **det}
        if r_["bound"] == "valu":
            v = committed_kernel_valu(a.config, a.inverse, r_["kernel"], pn)
            if v is not None and r_["avg_launch_ms"] > 0:
                rate = v["valu_wave_insts_per_launch"] / (r_["avg_launch_ms"] * 1e-3)
                peak = VALU_PEAK_WAVE_INSTS_PER_S_PER_CU * 256
                committed["valu_issue"][r_["kernel"]] = {**v, "wave_insts_per_s": rate, "peak_wave_insts_per_s": peak, "issue_frac": rate / peak,
                                                         "what": "committed SQ_INSTS_VALU per launch over THIS run's event-timed launch duration, against the NOMINAL rate of 256 CUs x 4 SIMDs x one wave64 instruction per 4 cycles at 2.4 GHz; "
                                                                 "measured (tools/micro/valu_rate.hip) the cheap instructions issue every 2.4-2.7 cycles, so a fraction at or above 1.0 means the SIMDs are saturated, not a counting error"}
    # whole-path algorithmic bytes: B_q summed over queries + B_idx, per step
    # (streamed set: the queries, or the targets with --inverse; for N > 1 rank 0's counters times the world size)
    # M = minimizers of the streamed set, counted ONCE: against a partitioned index every part looks all of them up, and the
    # library's counter adds them up per part (1.122 G = 3 x 374 M had inflated B_q by 24 GB)
    n_parts = max(1, int(acc_cn.get("index_parts", 0)))
    L = float((t_lens if a.inverse else q_lens).sum()); M = world * acc_cn.get("query_minimizers", 0) / K / n_parts; H = world * acc_cn.get("anchors", 0) / K
    B_q = L / 4 + 32 * M + 8 * H + 32 * H + 4 * Qn
    B_idx = float((q_lens if a.inverse else t_lens).sum()) / 4 + 16 * st["n_minimizers"]
    e2e_gbps = (B_q + B_idx) / (ms_per_step * 1e-3) / 1e9
    whole_path = {"what": "SURVEY 8(d), strictly: sum over queries of B_q = L/4 + 32 M + 40 H + 4, plus B_idx = L_T/4 + 16 M_T, over the step's wall time; "
                          "ordering (every radix pass) counts as ZERO algorithmic bytes",
                  "alg_GB_per_step": (B_q + B_idx) / 1e9, "B_q_GB": B_q / 1e9, "B_idx_GB": B_idx / 1e9, "alg_GBps": e2e_gbps, "peak_GBps": HBM_PEAK_GBPS,
                  "frac": e2e_gbps / HBM_PEAK_GBPS, "frac_of_measured_copy_peak": e2e_gbps / 6290.0,
                  "streamed_minimizers": M, "anchors": H, "index_minimizers": st["n_minimizers"], "index_parts": n_parts}
    wt = committed_traffic(a.config, a.inverse, pn)
    if wt is not None:
        committed["whole_path_traffic"] = wt
    # The path's roofline block = SURVEY 8(d), strictly: the keys of the contract, achieved = the path's algorithmic
    # bytes over the step's wall time.  The single kernels follow, the one with the most time per step first.
    roofline = {"bound": "hbm", "kernel": "whole path (SURVEY 8d: every stage of the step)", "achieved": e2e_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": e2e_gbps / HBM_PEAK_GBPS, "traffic": None, "traffic_note": "from_committed_profiles.whole_path_traffic (counters cannot be read in a timed run)",
                "alg_bytes_per_launch": B_q + B_idx, "avg_launch_ms": ms_per_step, "launches_per_step": 1, **{k: v for k, v in whole_path.items() if k not in ("alg_GBps", "peak_GBps", "frac")}}
    return roofline, cands, fams, committed
