#!/usr/bin/env python
"""bench.py -- reads overlapped/sec of the liblrge overlap hot path on MI355X.

Workload (default): BASELINE.json configs[4], the configuration the >= 10x target is quoted on and that fits ONE GPU since
round 3 -- H. sapiens-scale 3.1 Gbp synthetic PacBio HiFi set, two-set forward -Q 100000 -T 2000000 (31.5 Gbases), preset
ava-pb (liblrge's PacBio semantics; `--preset ont` is what the reference CLI runs: lrge/src/main.rs:56-85 never forwards -P).
The reads come from the counter-based generator (lrge_amd/synth_cb.py; 31.5 Gbases in ~1 s, untimed).  `--config
c4_dmel_twoset` (configs[3], rounds 1-3's line), `c2_bact_twoset` (configs[1]), `c5_human_tenth`, `--inverse` select the others.

One "step" = the whole job on one batch of reads: 2-bit pack (K0) of both sets, minimizer index build over the targets,
sketch + seed + chain of every query, distinct-target counts, per-read estimates, median (twoset.rs:587-606 from "reads in
memory" on).  TWO clocks, both measured in every run:
  * `value` / `ms_per_step` -- SURVEY.md 8(d)'s clock: the ASCII reads are FASTA-parsed reads in (pinned) HOST memory when the
    clock starts; host-side 2-bit pack, the PCIe transfer (targets first, the queries travel while the target index is built)
    and everything behind them are inside.  This is the figure VERDICT r03 credits, and the smaller of the two.
  * `resident` -- the same step with the ASCII reads already resident in HBM (pack on the device, no PCIe): what rounds 1-3
    reported as `value`.  `--clock resident` makes it `value` again (and says so in config.clock).
`value` = query reads / wall time of EXACTLY --steps steps after --warmup warm-up steps, barrier + device synchronise on both sides.

N > 1 (launched by torch.distributed.run, one rank per GPU): STRONG scaling of the one job.  The query set is cut into
N contiguous ranges with equal base counts; every rank owns its range end to end.  The target index a rank needs is
restricted to the minimizers its own queries carry: every rank sketches 1/N of the targets, key sets are
all-gathered, kept entries and owned hashes travel by all-to-all (one small all-reduce fixes mid_occ exactly).  One all-gather of
the per-read estimates closes the step (RCCL through the library's C ABI: lrge_hip_comm_*).  value = all query reads /
max-over-ranks time.
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

def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c5_human_twoset")
    ap.add_argument("--clock", default="host", choices=["host", "resident"],
                    help="which clock `value` is quoted on: host = SURVEY 8(d): ASCII reads in pinned host memory when the clock "
                         "starts (default); resident = ASCII reads already in HBM (rounds 1-3).  The other one is measured too "
                         "(`resident` / `from_host` on the line) unless --no-from-host / --no-resident")
    ap.add_argument("--no-resident", action="store_true")
    ap.add_argument("--parity-sample", type=int, default=256, help="query reads whose FORWARD counts the CPU oracle recomputes at full "
                                                                   "H. sapiens scale (restricted index: oracle/c5_sample.py); 0 = skip")
    ap.add_argument("--preset", default=None, choices=["ont", "pb"], help="minimap2 preset (default: by the configuration's platform: pb for the HiFi sets, ont otherwise)")
    ap.add_argument("--inverse", action="store_true", help="--use-min-ref: index the queries, stream the targets")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the config (debug only; invalid as a result)")
    ap.add_argument("--generator", default="auto", choices=["auto", "pcg", "cb"],
                    help="pcg: lrge_amd.synth (sequential numpy generator, host); cb: lrge_amd.synth_cb (counter-based, reads written "
                         "straight into HBM by the device twin -- what makes c5_human_twoset, 31.5 Gbases, a bench workload); "
                         "auto = cb for c5_human_twoset, pcg otherwise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-from-host", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--emulate-world", type=int, default=0, metavar="N",
                    help="projection on ONE GPU of an N-rank strong-scaling run of the forward strategy with the target sketch sharded: "
                         "all N ranks run for real as threads of this process (one context each, the library's local communicator), "
                         "taking turns on the GPU, so that every rank's share is timed alone; results are checked against the "
                         "one-GPU run.  A projection input (per-rank busy time + exchange volumes), not a bench result")
    ap.add_argument("--emulate-rank", default=None, metavar="R/N",
                    help="timing emulation on ONE GPU of rank R of an N-rank strong-scaling run: this process maps rank R's query "
                         "range against an index restricted to it and counts R's 1/N share of the occurrence statistics; nobody "
                         "supplies the other shares, so mid_occ is incomplete and the RESULTS ARE INVALID -- the line is a "
                         "projection input (DESIGN.md section 7), not a bench result")
    return ap.parse_args()


from benchlib import (cpu_baseline, cpu_baseline_sampled, roofline_blocks, emulate_world)  # noqa: E402


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world == 1 and a.gpus > 1:
        sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
    import torch
    import torch.distributed as dist
    from lrge_amd import engine, parallel, synth

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback)")
    if os.environ.get("LRGE_BENCH_SHARE_GPU") == "1":     # tests on a 1-GPU box: all ranks of a world on the devices there are
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("LRGE_BENCH_FORCE_DIST") == "1"   # the switch lets a 1-GPU box exercise RCCL
    if a.preset is None:
        a.preset = "pb" if a.config.startswith("c5_") else "ont"        # (the C5 configurations are the HiFi ones)
    preset = 1 if a.preset == "pb" else 0

    # ---- synthetic inputs (untimed; every rank draws the same seeded job) ----
    gen = a.generator if a.generator != "auto" else ("cb" if a.config == "c5_human_twoset" else "pcg")
    t_gen = time.perf_counter()
    if gen == "cb":
        from lrge_amd import synth_cb
        spec, Qn, Tn = synth_cb.spec_of(a.config, a.scale)
        cfg = dict(synth_cb.CONFIGS[a.config], mode="twoset")
        gsize = spec.gsize
        q = t = None
        # emitted lengths of every read (one counting pass of the device twin; the bases themselves follow below, per rank)
        all_lens = spec.emitted_lens(0, Qn + Tn, local_rank)
        q_lens, t_lens = all_lens[:Qn].astype(np.int64), all_lens[Qn:].astype(np.int64)
    else:
        cfg = synth.CONFIGS[a.config]
        assert cfg["mode"] == "twoset", "bench.py times the two-set path"
        gsize, q, t = synth.make_config(a.config, a.scale)
        Qn, Tn = q.n, t.n
        q_lens, t_lens = q.lens(), t.lens()
    t_gen = time.perf_counter() - t_gen

    # which side packs the reads on the host clock: the host's CPUs are shared by the ranks of this node (one process per GPU), and a rank
    # with fewer than 8 of them ships ASCII over its own PCIe link and packs on the device (lrge_hip_pack_choice).
    # The library reads LRGE_HIP_RANKS_ON_HOST when the context is created.
    os.environ.setdefault("LRGE_HIP_RANKS_ON_HOST", str(int(os.environ.get("LOCAL_WORLD_SIZE", world))))
    ctx = engine.Context(local_rank)
    import ctypes as _C
    from lrge_amd import _ffi as _ffi_
    _gr = _C.c_double()
    _roh = int(os.environ["LRGE_HIP_RANKS_ON_HOST"])
    _host_side = bool(_ffi_.lib().lrge_hip_pack_choice(_roh, _C.byref(_gr)))
    if os.environ.get("LRGE_HIP_PACK") in ("host", "device"):
        _host_side = os.environ["LRGE_HIP_PACK"] == "host"
    pack_info = {"chosen": "host (AVX2 2-bit pack, packed words over PCIe)" if _host_side else "device (ASCII over the rank's PCIe link, k_pack)",
                 "granted_cpus": _gr.value, "ranks_on_host": _roh, "cpus_per_rank": _gr.value / max(1, _roh),
                 "rule": "one rank: host; several: host only if every rank has 8 granted CPUs (LRGE_HIP_PACK overrides)",
                 "applies_to": "reads handed over in host memory (the `host` clock); resident reads are packed on the device"}
    comm, transport, rccl_thread = None, None, None
    if use_dist:
        # torch.distributed only bootstraps (TCP store over gloo): the data-path collectives are the library's own RCCL
        # communicator behind the C ABI (lrge_hip_comm_*), created from a unique id that rank 0 hands out
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        transport = "rccl"
        if os.environ.get("LRGE_BENCH_TRANSPORT") == "host":
            transport = "host (gloo) -- requested"
        else:
            # ncclCommInitRank blocks until every rank has joined; a mis-set network interface would hang it for good, so the
            # bootstrap runs in a helper thread with a deadline -- past it the collectives travel over gloo on host buffers
            # (and the line says so)
            import threading
            box = {}

            def _boot():
                try:
                    box["comm"] = parallel.RcclComm.bootstrap(ctx, rank, world, dist)
                except Exception as e:      # noqa: BLE001
                    box["err"] = e
            rccl_thread = threading.Thread(target=_boot, daemon=True)
            rccl_thread.start()
            rccl_thread.join(timeout=float(os.environ.get("LRGE_BENCH_RCCL_TIMEOUT", "240")))
            comm = box.get("comm")
            if comm is None:
                transport = "host (gloo) -- RCCL communicator not created: %s" % (str(box.get("err"))[:200] if "err" in box else "timed out")
        # every rank must agree on the transport
        flag = torch.tensor([1 if comm is not None else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if comm is not None:
                comm.close()
                transport = "host (gloo) -- another rank could not create its RCCL communicator"
            comm = parallel.HostComm(ctx, dist)

    emu = None
    if a.emulate_rank:
        er, en = (int(x) for x in a.emulate_rank.split("/"))
        assert world == 1 and 0 <= er < en
        emu = (er, en)
        ctx.set_option("DEBUG_OWN_SHARE", "%d,%d" % (en, er))
    avg_t = np.float32(t_lens.sum()) / np.float32(Tn)
    if gen != "cb":
        qr_all, tr_all = engine.name_ranks(q.names, t.names)
    else:       # names are r%08d of the read index, so the index is the lexicographic rank over the union of both sets
        qr_all, tr_all = np.arange(0, Qn, dtype=np.uint32), np.arange(Qn, Qn + Tn, dtype=np.uint32)
    # forward strategy on more than one GPU: the target sketch is sharded too (lrge_hip_index_build_sharded)
    # unless LRGE_BENCH_REPLICATED_SKETCH asks for round 2's form (every rank sketches all targets: lrge_hip_index_build_for)
    shard_targets = not a.inverse and not os.environ.get("LRGE_BENCH_REPLICATED_SKETCH")
    # ... and since round 4 the default multi-GPU form of the forward strategy shards the TARGETS (lrge_hip_index_build_tsharded): every
    # rank indexes its share of the targets and maps ALL queries, the count vectors add up in one all-reduce; no index entry crosses a
    # link.  LRGE_BENCH_FORWARD=qshard selects the query-sharded form of rounds 2-3.
    # LRGE_BENCH_FORWARD=replicated: north_star's LITERAL form -- the target index replicated (every rank builds the whole of it, no
    # collective inside the build), the queries cut by bases, one all-gather of the per-read estimates -- selectable so that the forms
    # above are measured against it instead of argued (missing 6)
    forward_mode = os.environ.get("LRGE_BENCH_FORWARD", "auto")
    if forward_mode == "auto":
        # every rank of the target-sharded form sketches and looks up ALL queries: it pays when that redundant work is small against a
        # rank's share of the index build -- query bases x ranks <= target bases (H. sapiens scale: 1.5 G x 8 against 30 G: 5.2x at 8
        # ranks against 3.3x query-sharded; C4: 0.36 G x 8 against 0.72 G: 1.7x against 2.3x the other way round)
        n_ranks = max(world, a.emulate_world or 1)
        forward_mode = "tshard" if float(q_lens.sum()) * n_ranks <= float(t_lens.sum()) else "qshard"

    class Src:
        """One read set of one rank's job: offsets, name ranks, and where its ASCII bases live (HBM; host on request)."""

    _shared = {}

    class _SharedDev:
        """A DeviceReads several ranks of an emulated world hold together: free() is the last holder's."""

        def __init__(self, dev):
            self.__dict__["_d"] = dev
            self.__dict__["_n"] = 0

        def __getattr__(self, k):
            return getattr(self._d, k)

        def free(self):
            self.__dict__["_n"] -= 1
            if self._n <= 0 and self._d.ptr:
                self._d.free()
                _shared.pop(self._key, None)

    def _shared_reads(first, n, device):
        key = (first, n, device)
        if key not in _shared or not _shared[key]._d.ptr:
            sd = _SharedDev(spec.device_reads(first, n, device))
            sd.__dict__["_key"] = key
            _shared[key] = sd
        _shared[key].__dict__["_n"] += 1
        return _shared[key]

    class RankJob:
        """What one rank does per step.  my / n_shards: which share of the STREAMED set it owns -- contiguous ranges with equal
        base counts, the queries in the forward strategy (twoset.rs:266-334), the targets with --inverse (twoset.rs:485-565)."""

        def __init__(self, ctx, comm, my, n_shards, device, emulated_share=False):
            self.ctx, self.comm, self.my, self.n_shards = ctx, comm, my, n_shards
            self.bounds = parallel.shard_by_bases(t_lens if a.inverse else q_lens, n_shards)
            lo, hi = self.bounds[my], self.bounds[my + 1]
            q_rng = (0, Qn) if (a.inverse or n_shards == 1) else (lo, hi)
            t_rng = (lo, hi) if (a.inverse and n_shards > 1) else (0, Tn)
            self.tshard = forward_mode == "tshard" and not a.inverse and n_shards > 1 and comm is not None
            self.replicated = forward_mode == "replicated" and not a.inverse and n_shards > 1 and comm is not None
            self.sharded = shard_targets and n_shards > 1 and comm is not None and not self.tshard and not self.replicated
            if self.sharded or self.tshard:
                tb_ = parallel.shard_by_bases(t_lens, n_shards)
                t_rng = (tb_[my], tb_[my + 1])
            if self.tshard:
                q_rng = (0, Qn)
                self.bounds = tb_
            self.restrict = (comm is not None or emulated_share or bool(os.environ.get("LRGE_BENCH_RESTRICT"))) and not a.inverse and not self.replicated
            self.q_rng, self.t_rng = q_rng, t_rng
            self.shard_lens = [self.bounds[i + 1] - self.bounds[i] for i in range(n_shards)]
            self.max_shard = max(self.shard_lens)
            self.qs, self.ts = Src(), Src()
            self.device = device
            if gen == "cb":
                self.load_resident()
            else:
                for S, R, rk, (r0, r1) in ((self.qs, q, qr_all, q_rng), (self.ts, t, tr_all, t_rng)):
                    sub = R if (r0, r1) == (0, R.n) else R.slice(r0, r1)
                    S.n, S.offsets, S.rank, S.nbytes = sub.n, sub.offsets, rk[r0:r1], sub.bases.size
                    S.tensor = torch.from_numpy(sub.bases).to("cuda:%d" % device)     # the ASCII reads resident in HBM (value)
                    S.ptr = S.tensor.data_ptr()
                    S.host = (lambda sub=sub: sub.bases)
            self.qs_lens = q_lens[q_rng[0]:q_rng[1]]
            self.shard_stats = None

        def load_resident(self):
            """(counter-based generator) the ASCII reads of this rank written into HBM by the device twin"""
            for S, (r0, r1), first in ((self.qs, self.q_rng, 0), (self.ts, self.t_rng, Qn)):
                if self.replicated and S is self.ts:      # every rank of an emulated world reads the SAME targets: one copy serves them all
                    S.dev = _shared_reads(first + r0, r1 - r0, self.device)
                else:
                    S.dev = spec.device_reads(first + r0, r1 - r0, self.device)      # ASCII straight into HBM
                S.n, S.offsets, S.rank, S.nbytes, S.ptr = r1 - r0, S.dev.offsets, S.dev.name_ranks(), S.dev.total_bases, S.dev.ptr
                S.host = S.dev.to_host

        def step(self, src_q, src_t):
            """One step; a rank that fails anywhere in it aborts the communicator (lrge_hip_comm_abort) on its way out, so that its
            peers leave the collective they are waiting in with an error instead of waiting for ever."""
            try:
                return self._step(src_q, src_t)
            except BaseException:
                if self.comm is not None:
                    try:
                        self.comm.abort()
                    except Exception:      # noqa: BLE001
                        pass
                raise

        def _step(self, src_q, src_t):
            """src_*: int device pointer (ASCII resident in HBM) or PinnedBuffer (ASCII in pinned host memory)."""
            ctx, comm, qs, ts = self.ctx, self.comm, self.qs, self.ts
            if a.inverse:
                # index = the query set (small), the streamed targets of this rank travel / pack while it is built
                Qd = ctx.upload(src_q, qs.offsets, qs.rank, wait=False)
                Td = ctx.upload(src_t, ts.offsets, ts.rank, wait=False)
                if not os.environ.get("LRGE_BENCH_NO_PRESKETCH"):
                    Td.presketch(preset)
                ix = engine.Index(ctx, Qd, preset)
                tb = dict(ix.build_timings); cb_ = dict(ix.build_counters)
                counts = ix.overlap_inverse(Td)
                tm = ctx.timings(); cn = ctx.counters(); st = ix.stats()
                ix.free(); Qd.free(); Td.free()
                if comm is not None:    # the one collective that closes the step: count vector keyed by indexed read (twoset.rs:520-523)
                    counts = comm.all_reduce_u32(counts)
                est_all = ctx.estimates(counts, q_lens, float(avg_t), Tn, 100)
            else:
                _ta = time.perf_counter()
                qshard = self.tshard and not os.environ.get("LRGE_BENCH_NO_QSHARD")
                if qshard:
                    # target-sharded form (round 6): every rank maps ALL queries, but sketches only ITS share of them -- the queries travel
                    # first (they are the small set), each rank runs K1 over its share while its targets travel, the minimizers are
                    # all-gathered (lrge_hip_seqset_presketch_sharded: 16 B each, once per step and world instead of a sketch per rank)
                    Qd = ctx.upload(src_q, qs.offsets, qs.rank, wait=False)
                    _tb = time.perf_counter()
                    Td = ctx.upload(src_t, ts.offsets, ts.rank, wait=False)
                    Qd.presketch_sharded(preset, comm)
                else:
                    Td = ctx.upload(src_t, ts.offsets, ts.rank, wait=False)      # K0 pack (and PCIe, from the host) on the copy stream
                    _tb = time.perf_counter()
                    Qd = ctx.upload(src_q, qs.offsets, qs.rank, wait=False)      # travels / packs while the index is built
                    if not os.environ.get("LRGE_BENCH_NO_PRESKETCH"):
                        Qd.presketch(preset)     # the queries are sketched beside the index's sort / table passes (still once per step)
                _tr = os.environ.get("LRGE_BENCH_TRACE") and self.my == 0 and comm is not None

                def _busy():        # (trace only) flush the turn so that busy_ms is current
                    comm.turn(False); b = comm.busy_ms(); comm.turn(True); return b
                _b0 = _busy() if _tr else 0.0
                _t0 = time.perf_counter()
                if self.tshard:
                    ix = engine.Index(ctx, Td, preset, comm=comm, tshard=True)
                    self.shard_stats = ix.shard_stats
                elif self.sharded:
                    ix = engine.Index(ctx, Td, preset, streamed=Qd, comm=comm, shard=(t_lens, tr_all, self.t_rng[0]))
                    self.shard_stats = ix.shard_stats
                elif self.replicated:      # the whole index on every rank: the plain single-GPU build, no collective
                    ix = engine.Index(ctx, Td, preset)
                else:
                    ix = engine.Index(ctx, Td, preset, streamed=Qd if self.restrict else None, comm=comm)
                tb = dict(ix.build_timings); cb_ = dict(ix.build_counters)
                _t1 = time.perf_counter()
                _b1 = _busy() if _tr else 0.0
                counts, has = ix.overlap_twoset(Qd)
                _t2 = time.perf_counter()
                _b2 = _busy() if _tr else 0.0
                tm = ctx.timings(); cn = ctx.counters(); st = ix.stats()
                ix.free(); Qd.free(); Td.free()
                _t3 = time.perf_counter()
                if self.tshard:         # the one collective that closes the step: the count vectors over disjoint targets add up
                    counts = comm.all_reduce_u32(counts)
                    est_all = ctx.estimates(counts, q_lens, float(avg_t), Tn, 100)
                    if _tr:
                        _b3 = _busy()
                        sys.stderr.write("[trace rank 0] busy: up to the uploads %.2f | index build %.2f | overlap %.2f | all-reduce + estimates %.2f ms;  wall: index %.2f overlap %.2f rest %.2f\n"
                                         % (_b0, _b1 - _b0, _b2 - _b1, _b3 - _b2, (_t1 - _t0) * 1e3, (_t2 - _t1) * 1e3, (time.perf_counter() - _t3) * 1e3))
                else:
                    est = ctx.estimates(counts, self.qs_lens, float(avg_t), Tn, 100)
                    if comm is not None:    # the one collective that closes the step: per-read estimate vectors over RCCL/xGMI
                        est_all = comm.all_gather_f32(est, self.max_shard, self.shard_lens)
                    else:
                        est_all = est
            _t4 = time.perf_counter()
            med = engine.median(est_all, True, 0.15, 0.65)
            if os.environ.get("LRGE_BENCH_TICKS") and not a.inverse:      # host wall time of the step's calls (where the GPU idles between two steps)
                sys.stderr.write("[ticks] upload T %.2f | upload Q + hint %.2f | index %.2f | overlap %.2f | introspection + free %.2f | estimates %.2f | median %.2f ms\n"
                                 % ((_tb - _ta) * 1e3, (_t0 - _tb) * 1e3, (_t1 - _t0) * 1e3, (_t2 - _t1) * 1e3, (_t3 - _t2) * 1e3, (_t4 - _t3) * 1e3, (time.perf_counter() - _t4) * 1e3))
            for k_ in ("rs_scatter_launches", "rs_scatter_items", "rs_scatter_bytes", "sketch_launches", "sketch_wave_launches"):   # the index build sorts (and sketches) too
                cn[k_] = cn.get(k_, 0) + cb_.get(k_, 0)
            return counts, est_all, med, tb, tm, cn, st

    if a.emulate_world:
        emulate_world(a, ctx, RankJob, Qn, engine, parallel, local_rank, q_lens, t_lens, forward_mode)
        ctx.close()
        return
    job = RankJob(ctx, comm, emu[0] if emu else rank, emu[1] if emu else world, local_rank, emulated_share=bool(emu))
    qs, ts, step = job.qs, job.ts, job.step
    torch.cuda.synchronize()

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(src_q, src_t, warmup, steps):
        for _ in range(warmup):
            step(src_q, src_t)
        sync_all()
        t0 = time.perf_counter()
        acc_tb, acc_tm, acc_cn, last = {}, {}, {}, None
        for _ in range(steps):
            last = step(src_q, src_t)
            _, _, _, tb, tm, cn, _ = last
            for k, v in tb.items(): acc_tb[k] = acc_tb.get(k, 0.0) + v
            for k, v in tm.items(): acc_tm[k] = acc_tm.get(k, 0.0) + v
            for k, v in cn.items(): acc_cn[k] = v if k in ("lpg_split", "index_parts") else acc_cn.get(k, 0) + v
        sync_all()
        elapsed = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, acc_tb, acc_tm, acc_cn, last

    # big jobs (launches of milliseconds): an event pair around every k_rs_scatter launch costs nothing there, so the TIMED steps
    # carry them; small jobs time that kernel in one instrumented step behind the timed region (timer level 2 is host work
    # between launches: ~2 % of a C2 step)
    lvl2_all = float(q_lens.sum()) + float(t_lens.sum()) > 5e9
    if lvl2_all:
        ctx.set_timer_level(2)
    # ---- the two clocks ----
    # host (`value` by default): the ASCII reads sit in pinned host memory when the clock starts.  The pinned copies are
    # made from the resident reads first (untimed); with the counter-based generator the resident copies then make room for
    # the staging blocks of the uploads and are written again by the device twin before the resident clock runs.
    clock_host = a.clock == "host" and not emu
    want_host = clock_host or (not a.no_from_host and not emu)
    want_res = (not clock_host) or not a.no_resident
    res_host = res_res = None
    k_other = a.steps if a.config != "c5_human_twoset" else max(2, a.steps // 3)    # steps of the clock that is not `value`
    hq = ht = None

    def run_host(warmup, steps):
        return timed(hq, ht, warmup, steps)

    def run_resident(warmup, steps):
        return timed(qs.ptr, ts.ptr, warmup, steps)

    if want_host:
        hq = ctx.host_alloc(max(qs.nbytes, 1)); ht = ctx.host_alloc(max(ts.nbytes, 1))
        hq.array[:qs.nbytes] = qs.host(); ht.array[:ts.nbytes] = ts.host()
    if clock_host:
        if gen == "cb":
            qs.dev.free(); ts.dev.free()
        res_host = run_host(a.warmup, a.steps)
        if want_res:
            if gen == "cb":
                ctx.set_option("POOL_TRIM", "1")       # the arena of the steps behind us holds most of the HBM: the ASCII reads need 31.5 GB of it back
                job.load_resident()
            res_res = run_resident(1, k_other)
    else:
        res_res = run_resident(a.warmup, a.steps)
        if want_host:
            if gen == "cb":
                qs.dev.free(); ts.dev.free()
            res_host = run_host(1, k_other)
            if gen == "cb":
                job.load_resident()
    main_res = res_host if clock_host else res_res
    elapsed, acc_tb, acc_tm, acc_cn, last = main_res
    counts, est_all, med, _, _, _, st = last
    # one instrumented step AFTER the timed regions: an event pair around every k_rs_scatter launch (timer level 2 costs
    # host work between launches, so the timed steps run at the default level)
    tb2 = tm2 = cn2 = {}
    if not lvl2_all:
        ctx.set_timer_level(2)
        if clock_host and not want_res:
            _, _, _, tb2, tm2, cn2, _ = step(hq, ht)
        else:
            _, _, _, tb2, tm2, cn2, _ = step(qs.ptr, ts.ptr)
        ctx.set_timer_level(1)

    def clock_block(res, steps, what):
        if res is None:
            return None
        e_, _, tm_, _, last_ = res
        return {"ms_per_step": e_ * 1e3 / steps, "value": Qn * steps / e_, "unit": "reads/s", "steps": steps, "what": what,
                "counts_equal_other_clock": bool(np.array_equal(last_[0], counts))}
    what_host = ("ASCII reads in pinned host memory when the clock starts (SURVEY 8d): %.2f GB handed over per step; host-side 2-bit "
                 "pack + PCIe on the copy stream inside the timed region (targets first, the queries travel while the index is built)"
                 % ((qs.nbytes + ts.nbytes) / 1e9))
    what_res = "ASCII reads already resident in HBM when the clock starts; 2-bit pack on the device inside the step (rounds 1-3's `value`)"
    from_host = None if clock_host else clock_block(res_host, k_other, what_host)
    resident = clock_block(res_res, k_other, what_res) if clock_host else None
    if hq is not None:
        hq.free(); ht.free()

    if emu:
        K = a.steps
        print(json.dumps({"emulated_rank": "%d/%d" % emu, "INVALID_AS_RESULT": "one rank's share of the work on one GPU; mid_occ incomplete",
                          "ms_per_step": elapsed * 1e3 / K, "query_reads_of_rank": qs.n, "config": a.config,
                          "stage_ms_per_step": {**{"index_" + k: v / K for k, v in acc_tb.items() if v and k != "total"},
                                                **{k: v / K for k, v in acc_tm.items() if v}}}))
        ctx.close()
        return
    if rank == 0:
        K = a.steps
        ms_per_step = elapsed * 1e3 / K
        value = Qn * K / elapsed

        roofline, kernels, fams, committed = roofline_blocks(a, preset, world, K, ms_per_step, acc_tb, acc_tm, acc_cn, tb2, tm2, cn2, lvl2_all, st, Qn, Tn, q_lens, t_lens)
        # the single kernel with the most time per step among those SURVEY 8(d) gives bytes (k_rs_scatter's launches may sum to more, but 8(d)
        # counts ordering as zero: its block -- with the builder's own "once in, once out" denominator, labelled -- stays in roofline_other)
        dom = next((k_ for k_ in kernels if not k_.get("denominator_is_not_8d")), None)
        clock_txt = ("ASCII reads in pinned HOST memory when the clock starts (SURVEY 8d): host-side 2-bit pack + PCIe inside the step"
                     if clock_host else "ASCII reads resident in HBM, 2-bit pack inside the step")
        out = {
            "metric": "reads overlapped/sec (whole node)", "value": value, "unit": "reads/s",
            "n_gpus": world, "steps": K, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64 keys / i32 chain scores / f32 gap penalty", "data": "synthetic",
            "config": {"workload": "%s: %.1f Mbp genome, %s reads, two-set %s -Q %d -T %d, preset %s, dual=yes; %s"
                                   % (a.config, gsize / 1e6, cfg["platform"], "--use-min-ref (index = queries, targets streamed)" if a.inverse else "forward",
                                      Qn, Tn, "ava-pb" if preset else "ava-ont", clock_txt),
                       "clock": "host" if clock_host else "resident",
                       "pack": pack_info,
                       "query_reads": Qn, "target_reads": Tn,
                       "parallelism": ("one job, streamed targets cut into %d ranges by bases; query index replicated; counts all-reduced" % world) if a.inverse else
                                      ("one job, TARGETS cut into %d ranges by bases: every rank indexes its range and maps all queries (sketched once per world: every "
                                       "rank sketches 1/%d of them, the minimizers all-gathered); occurrence statistics made "
                                       "global by one all-to-all of (key, count) pairs; the count vectors all-reduced" % (world, world)) if job.tshard else
                                      ("one job, queries cut into %d ranges by bases; the WHOLE target index built on every rank (north_star's literal form: "
                                       "replicated index, one gather of the estimates)" % world) if job.replicated else
                                      ("one job, queries cut into %d ranges by bases; index %s" %
                                       (world, ("restricted to each rank's query minimizers, global occurrence statistics by one all-reduce" +
                                                ("; every rank sketches 1/%d of the targets, key sets all-gathered, kept entries and owned hashes by all-to-all" % world if job.sharded else ""))
                                        if world > 1 else "over all targets")),
                       "collectives": transport if use_dist else None,
                       "rccl_ranks": (comm.rccl_ranks() if (comm is not None and hasattr(comm, "rccl_ranks")) else None),
                       "target_sketch": ("targets sharded, all queries mapped by every rank (lrge_hip_index_build_tsharded)" if job.tshard else
                                         "sharded (lrge_hip_index_build_sharded)" if job.sharded else "replicated (whole index per rank)" if job.replicated else "replicated") if world > 1 and not a.inverse else None,
                       "exchange_per_step_rank0": job.shard_stats,
                       "scale": a.scale, "data_gen_s": round(t_gen, 1)},
            "resident": resident,
            "from_host": from_host,
            "genome_size_true": gsize,
            "genome_size_estimate": None if med[1] is None else float(med[1]),
            "genome_size_abs_error": None if med[1] is None else abs(float(med[1]) - gsize),
            "estimate_q15_q65": [None if med[0] is None else float(med[0]), None if med[2] is None else float(med[2])],
            "mid_occ": st["mid_occ"],
            # `roofline` = the PATH: SURVEY 8(d)'s algorithmic bytes (ordering counts as zero) over the step's wall time.  `roofline_dominant_kernel`
            # = the single event-timed kernel with the most time per step, `roofline_other` = the other kernels and the kernel families
            # (k_rs_scatter's denominator is not 8(d)'s -- which counts ordering as zero -- but what any sort must move: it says so itself);
            # bound = "valu": bound by the VALU issue rate, its HBM fraction is for reference.  Everything read from profiles/*.json
            # (counter passes cannot run inside a timed run) sits under the ONE key `from_committed_profiles`.
            "roofline": roofline,
            "roofline_dominant_kernel": dom,
            "roofline_other": [k_ for k_ in kernels if k_ is not dom] + fams,
            "from_committed_profiles": committed,
            "stage_ms_per_step": {**{"index_" + k: v / K for k, v in acc_tb.items() if v and k != "total"},
                                  **{k: v / K for k, v in acc_tm.items() if v}},
            "work_per_step": {k: (v if k in ("lpg_split", "index_parts") else v / K) for k, v in acc_cn.items()},
        }
        if world == 1 and not a.no_cpu_baseline and not a.inverse:   # (the CPU leg times the forward strategy)
            if gen == "cb" and Tn > 400000:
                cb = cpu_baseline_sampled(spec, Qn, Tn, a.cpu_seconds, preset)
                out["cpu_baseline"] = cb
                out["gpu_vs_cpu_port_SAMPLE"] = value / cb["value"]      # (the denominator is a pro-rated SAMPLE timed in this run: see cpu_baseline.in_run_sample)
                # The WHOLE job was timed once on the port (rounds 5 and 6, tools/c5_allcounts.py: all 100 000 forward counts equal the GPU's; ~6.5
                # minutes of CPU time do not fit a default bench run).  When that committed record is of this very configuration it is the
                # better figure -- the in-run sample pro-rates x40 and understates the port 2.4x -- so it becomes cpu_baseline.value, with
                # the in-run sample beside it and the record itself under from_committed_profiles
                try:
                    # (round 6: one record per preset, timed with that round's code; round 5's ava-pb record otherwise)
                    rec = "r06_c5_full_forward_allcounts_%s.json" % ("ava_pb" if preset else "ava_ont")
                    if not os.path.exists(os.path.join(ROOT, "profiles", rec)):
                        rec = "r05_c5_full_forward_allcounts.json"
                    with open(os.path.join(ROOT, "profiles", rec)) as f:
                        ac = json.load(f)
                    if a.config == ac.get("config") and a.scale == ac.get("scale") and ac.get("preset") == ("ava-pb" if preset else "ava-ont"):
                        m = ac["cpu_port_measured"]
                        committed["cpu_port_measured_full_job"] = {"reads_per_s": m["reads_per_s"], "job_seconds": m["job_seconds"], "index_seconds": m["index_seconds"],
                                                                    "map_seconds": m["map_seconds"], "threads": m["threads"], "counts_equal_gpu": ac["oracle_map"]["counts_equal"],
                                                                    "reads_checked": ac["oracle_map"]["reads_checked"], "note": m["note"],
                                                                    "file": "profiles/%s (a committed record, not timed in this run)" % rec}
                        cb["in_run_sample"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "sample": cb["sample"]}
                        cb["value"] = m["reads_per_s"]; cb["cores"] = m["threads"]
                        cb["value_source"] = ("from_committed_profiles.cpu_port_measured_full_job: the whole job timed once on the port (%.0f s on %d threads); "
                                              "in_run_sample is what THIS run timed" % (m["job_seconds"], m["threads"]))
                        cb["sample"] = "the whole job: all %d query reads against all %d target reads (committed record); in_run_sample: this run's bounded sample" % (Qn, Tn)
                        out["gpu_vs_cpu_port_MEASURED_FULL_JOB"] = value / m["reads_per_s"]
                except Exception:      # noqa: BLE001
                    pass
                out["parity_vs_oracle_sample"] = None
                if a.parity_sample > 0 and a.config == "c5_human_twoset" and a.scale == 1.0:
                    # FORWARD counts of a sample of queries against the oracle at full size: the oracle's index restricted to the keys
                    # of the sample (oracle/c5_sample.py; every target read goes through the oracle's mm_sketch), mid_occ / n_keys /
                    # n_minimizers from the oracle's committed KeyStats fixture
                    from oracle import c5_sample
                    from oracle import oracle as _O
                    O_threads = _O.default_threads()
                    fx = c5_sample.fixture_stats(spec, Qn, Tn, "ava-pb" if preset else "ava-ont")
                    idx = c5_sample.sample_indices(Qn, a.parity_sample)
                    if job.qs.dev.ptr:       # make room: the sample's target chunks are written by the device twin
                        job.qs.dev.free(); job.ts.dev.free()
                    r = c5_sample.forward_sample(spec, Qn, Tn, preset, idx, fx["mid_occ"], source="device", device=local_rank)
                    out["parity_vs_oracle_sample"] = {
                        "reads": int(len(idx)), "which": "query reads i * Q / %d, i = 0 .. %d (every anchor batch and index part)" % (len(idx), len(idx) - 1),
                        "counts_equal": bool(np.array_equal(r["counts"], counts[idx])),
                        "mid_occ_equal": bool(fx["mid_occ"] == st["mid_occ"]),
                        "n_minimizers_equal": bool(fx["n_minimizers"] == st["n_minimizers"] == r["n_minimizers_seen"]),
                        "n_keys_equal": bool(fx["n_keys"] == st["n_keys"]),
                        "oracle": "restricted index (lo_ridx_*): complete position lists of the sample's keys out of all %d target minimizers "
                                  "(%d kept); %.1f s on %d threads (reads %.1f, sketch + filter %.1f, map %.1f)"
                                  % (r["n_minimizers_seen"], r["n_kept"], r["seconds"], O_threads, r["seconds_reads"], r["seconds_sketch"], r["seconds_map"]),
                        "overlaps_in_sample": int(r["counts"].sum())}
            else:
                if gen == "cb":
                    q, t = spec.host_reads(first=0, n=Qn), spec.host_reads(first=Qn, n=Tn)
                cb, ccounts, cmid = cpu_baseline(q, t, a.cpu_seconds, preset)
                out["cpu_baseline"] = cb
                out["gpu_vs_cpu_port"] = value / cb["value"]
                n = len(ccounts)
                out["parity_vs_oracle_sample"] = {"reads": n, "counts_equal": bool(np.array_equal(ccounts, counts[:n])),
                                                  "mid_occ_equal": bool(cmid == st["mid_occ"])}
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    if rccl_thread is not None and rccl_thread.is_alive():     # a communicator bootstrap that never returned: leave without joining it
        sys.stdout.flush()
        os._exit(0)
    if use_dist:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
