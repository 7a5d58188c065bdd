// k_chain_lpg.h -- K6, lane-per-group form: SIXTY-FOUR groups per wavefront.
//
// k_chain_hw resolves the order-dependent state of mg_lchain_dp's predecessor loop with cross-lane
// scans; that costs ~195 VALU instructions per step of two anchors, of which comput_sc is ~35 -- the
// kernel is bound by VALU issue and per-step latency, not by memory.  Here every LANE owns
// one (query, target, strand) group and runs the plain sequential loop over its candidates, so the
// scans disappear: a candidate costs ~34 lane-operations and one wave instruction serves 64 groups.
//   * the last 32 anchors of each lane's group live in VGPR arrays (fully unrolled loops, static
//     indices), shifted by one slot per step;
//   * the t[] marks of the sequential loop become a 32-bit register mask (a mark is only ever read
//     in the step that wrote it);
//   * anchor i of every lane is prefetched one step ahead; (f, p) leave as one 8-byte store per lane;
//   * candidates beyond the 32-anchor window and long max_ii rescans continue per lane through HBM
//     with the same arithmetic (exact, rare);
//   * backtrack: every lane tracks its best chain end during the DP and walks that one chain; when the
//     first chain is accepted and no records are wanted that settles the group's flags, otherwise the
//     group falls back to the wave-wide backtrack_group().
// A step takes ~1400 instructions regardless of the group size (4.1 us: ~7 cycles per instruction of
// one wavefront, measured issue costs in tools/micro/valu_rate.hip), so the latency per anchor is ~8x
// that of k_chain_hw: groups above LPG_MAX_N anchors stay on k_chain_hw (the host splits the sorted
// list), everything else -- ~90 % of the anchors of the headline workload -- runs here.
#pragma once
#include "k_chain_hw.h"

#ifndef LPG_W
#define LPG_W 32
#endif
#ifndef LPG_OCC
#define LPG_OCC 2   // wavefronts per SIMD the register allocation aims at
#endif
#ifndef LPG_B
#define LPG_B 8    // candidates per evaluate / resolve block
#endif
#ifndef LPG_CH
#define LPG_CH 8   // anchors per input / output staging chunk
#endif
#ifndef LPG_NPM
#define LPG_NPM 2  // block boundaries at which the candidate scan tests its bound (pruned scan, below)
#endif
#ifndef LPG_WAVES
#define LPG_WAVES 1   // wavefronts per workgroup (they share the penalty table; every wavefront has its own ring)
#endif
#define LPG_RING_BYTES ((2 * (LPG_CH / 2) * 128 * 2 + LPG_CH * 64) * 8)      // per wavefront
#define LPG_MAX_AUTO 0xFFFFFFFEu   // split chosen per batch from the group-size census (host_overlap_batch.inl: OverlapRun::batch)

struct LpgChainArgs {
    const u64 *akey, *aval;
    const u32 *gstart;
    u32 n_groups; u64 n_anchors;
    const u32 *list;   // groups for this kernel, largest first
    u32 n_list;
    u64 *grec;         // [n_anchors]
    u32 *tmark;        // [n_anchors] zero-initialised
    u32 prio;          // raise the wavefronts' issue priority (they run beside k_chain_hw)
    // A lane's slow paths (candidates / rescans behind the 32-anchor window) walk HBM one element at a time: fine
    // when rare, hopeless on repeat-rich groups.  Measured: on clean input 0.6 % of the groups ever leave the window,
    // once or twice, ~100 iterations each; on a repeat-rich genome 4 % do, ~56 times each.  A lane gives its group up
    // when the group has left the window more than slow_entries times (or has cost more than slow_budget iterations):
    // it is appended to redo_list and chained afterwards by k_chain_hw_redo, whose slow paths scan 64 candidates per step.
    u32 *redo_list, *redo_count; u32 slow_budget, slow_entries;
    // ... both allowances GROW with the anchors the lane has chained so far (round 4): slow_rate iterations per anchor and one entry
    // per slow_entry_every anchors on top of the fixed ones.  A long HiFi group with a detour every few hundred anchors stays here
    // (giving ~10 000 of them up per H. sapiens-scale step cost 24 ms of k_chain_hw_redo), a repeat-rich group -- tens of slow
    // iterations per anchor -- still leaves at once.
    u32 slow_rate, slow_entry_every;
    u32 no_prune;      // option LPG_NO_PRUNE: never cut the candidate scan short (tests: the full loop and its slow paths)
};

// PENTAB: with chain_skip_scale == 0 (every preset lrge uses) comput_sc's penalty depends on dd alone --
// (i32)(pen_gap * dd + .5 * mg_log2(dd + 1)), and 0 for dd == 0 -- so it is tabulated once per wavefront in
// LDS ([0, bw] + one "out of band" entry) with the very same f32 operations, and a candidate needs no f32 math.
// FASTREACH (max_iter >= 64, i.e. always outside the tests): the number of window candidates in reach is not counted.
// x is sorted, so all 32 are in reach iff the oldest one is, which is all the "continue behind the window" test needs;
// and when the loop did not break, the exact end of the scanned range only feeds `mi < end_j`, which is false then:
// at that point mi is in x-reach (the rescan above just made it so), every anchor in x-reach sits in the window in front
// of end_j unless the window is exhausted, and an exhausted window with more candidates behind it takes the slow path,
// which computes end_j itself.  Saves a compare, a select and an add per candidate and 32 live compare masks.
template <bool PENTAB, bool FASTREACH>
__global__ __launch_bounds__(64 * LPG_WAVES) __attribute__((amdgpu_waves_per_eu(2, LPG_OCC))) void k_chain_lpg(LpgChainArgs R, ChainParams P, GroupOut out) {
    extern __shared__ i32 pen_tab[];   // [bw + 2] when PENTAB, then the anchor / record staging ring (LPG_RING_BYTES)
    // this kernel's longest wavefronts are the critical path of the chain stage; k_chain_hw's wavefronts on
    // the other stream share the SIMDs and should fill the gaps, not compete for issue slots
    if (R.prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (R.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (R.prio == 1) __builtin_amdgcn_s_setprio(1);
    const u32 li = blockIdx.x * (64 * LPG_WAVES) + threadIdx.x;
    const bool has = li < R.n_list;
    const u32 g = has ? R.list[li] : 0;
    const u32 s0 = has ? R.gstart[g] : 0;
    const u32 e0 = has ? ((g + 1 < R.n_groups) ? R.gstart[g + 1] : (u32)R.n_anchors) : 0;
    const i32 n = (i32)(e0 - s0);
    i32 n_max = n;
    n_max = wave_incl_max_i32(n_max, 0);
    n_max = __builtin_amdgcn_readlane(n_max, 63);
    const u64 rmask = (1ULL << P.kl.bits_rpos) - 1;
    const u64 *gk = R.akey + s0, *gv = R.aval + s0;
    u64 *grec = R.grec + s0;
    u32 *tmark = R.tmark + s0;

    const i32 maxdx = P.max_dist_x, bw = P.bw, max_skip = P.max_skip, max_iter = P.max_iter, min_sc = P.min_sc;
    const u32 dqlim = (u32)(P.max_dist_x < P.max_dist_y ? P.max_dist_x : P.max_dist_y);
    const float pen_gap = P.pen_gap, pen_skip = P.pen_skip;

    const u32 tabn = (u32)bw + 1;
    if (PENTAB) {
        for (u32 d = threadIdx.x; d < tabn; d += 64 * LPG_WAVES) {
            const float lin_pen = pen_gap * (float)(i32)d + pen_skip * 0.0f;
            float log_pen = mg_log2_dev((float)(i32)(d + 1));
            log_pen = d >= 1 ? log_pen : 0.0f;
            pen_tab[d] = (i32)(lin_pen + .5f * log_pen);
        }
        if (threadIdx.x == 0) pen_tab[tabn] = 1 << 30;            // dd > bw: pushes s below NEG_BIG
        __syncthreads();
    }
    // Anchor input / record output staging.  A lane walks its own group, so a plain per-step load touches 64
    // different cache lines per instruction and each line is re-fetched for every anchor it holds (measured:
    // 10x the algorithmic bytes).  Instead every lane streams its group in chunks of LPG_CH anchors: two 16-byte
    // direct-to-LDS loads per array (global_load_lds_dwordx4, no VGPRs) one chunk ahead, and (f, p) records leave
    // as one 32-byte run per lane and chunk.
    //   ring_k / ring_v: [buffer 2][pair LPG_CH/2][lane 64][2]   ring_o: [row LPG_CH][lane 64]
    u64 *ring_k = (u64 *)((char *)pen_tab + (PENTAB ? (((size_t)tabn + 1) * 4 + 15) / 16 * 16 : 0) + (size_t)(threadIdx.x >> 6) * LPG_RING_BYTES);
    u64 *ring_v = ring_k + 2 * (LPG_CH / 2) * 128;
    u64 *ring_o = ring_v + 2 * (LPG_CH / 2) * 128;
    const u32 lane = threadIdx.x & 63;
    auto issue_chunk = [&](i32 a0, i32 buf) {            // anchors [a0, a0 + LPG_CH) -> buffer buf (reads <= 1 anchor past n)
#pragma unroll
        for (int pr = 0; pr < LPG_CH / 2; ++pr) {
            if (a0 + 2 * pr < n) {
                __builtin_amdgcn_global_load_lds(gk + a0 + 2 * pr, ring_k + (buf * (LPG_CH / 2) + pr) * 128, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(gv + a0 + 2 * pr, ring_v + (buf * (LPG_CH / 2) + pr) * 128, 16, 0, 0);
            }
        }
    };
    issue_chunk(0, 0);

    // window: slot k <-> anchor i-1-k.  WO = one-hot of (j - p[j] - 1), 0 when there is no predecessor or it
    // lies >= 32 anchors back: shifted left by k+1 it is the mark that candidate k leaves on a later candidate.
    i32 WX[LPG_W], WY[LPG_W], WF[LPG_W], WS[LPG_W];
    u32 WO[LPG_W];
#pragma unroll
    // empty slots (i < 32) must never be candidates: y = INT32_MAX makes dq <= 0, f = 0 loses every max_ii rescan
    for (int k = 0; k < LPG_W; ++k) { WX[k] = 0; WY[k] = INT32_MAX; WF[k] = 0; WS[k] = 0; WO[k] = 0; }
    i32 mi = -1, mi_x = 0, mi_y = 0, mi_f = 0, mi_sp = 0;
    u64 bkey = 0;                                   // best chain end: f << 32 | i  (f >= min_sc)
    // PRUNED SCAN (round 4).  comput_sc(i, j) <= span(j) (it is min(span(j), dg) minus penalties), so candidate j can lift anchor
    // i's score to at most g(j) = f[j] + span(j).  PM[k] = the largest g over every anchor of the group that lies BEHIND the first
    // k + 1 blocks of the window (slots >= (k + 1) LPG_B, and everything that has left the window).  Once max_f has
    // reached that bound no later candidate can be STRICTLY better -- mg_lchain_dp's `sc > max_f` -- nor can the max_ii shortcut
    // (its candidate is either one of those, or already scanned), so (f, p) of the anchor are final and the scan stops: exact.  The
    // t[] marks and n_skip only live inside one anchor's scan.  Measured on the oracle's own loop: 24.8 candidates per anchor as
    // minimap2 scans them, 2.2 until the bound holds (ONT; all within 8), 26.6 -> 5.3 at HiFi (96.4 % within 8, all but 3e-6 within
    // 16): a wavefront leaves the candidate loop after the first block or two instead of the fourth.
    i32 PM[LPG_NPM];
#pragma unroll
    for (int k = 0; k < LPG_NPM; ++k) PM[k] = R.no_prune ? INT32_MAX : 0;
    // the same idea for the max_ii rescan ("best f in reach", below): PF = the largest f among the anchors that have LEFT the window.
    // The rescan wants the newest anchor of maximal f in x-reach; when the window already holds an f >= PF it is in the window (a
    // tie goes to the newer anchor) and the walk through HBM behind the window -- up to 5000 bp of anchors, the reason groups were
    // given up to k_chain_hw_redo: 24 ms per H. sapiens-scale step -- is not needed.  (PF also counts anchors out of reach: a
    // sufficient test, never a wrong one.)
    i32 PF = R.no_prune ? INT32_MAX : 0;

    u32 slow_iters = 0, slow_entries = 0;
    bool abandoned = false;
    for (i32 i = 0; i < n_max; ++i) {
        const bool alive = i < n && !abandoned;
        const i32 r4 = i & (LPG_CH - 1), cb = (i / LPG_CH) & 1;
        if (r4 == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this chunk has landed (issued LPG_CH steps ago)
            issue_chunk(i + LPG_CH, cb ^ 1);
        }
        const u32 ri = ((cb * (LPG_CH / 2) + (r4 >> 1)) * 64 + lane) * 2 + (r4 & 1);
        const u64 ck = ring_k[ri], cv = ring_v[ri];
        const i32 xi = (i32)(ck & rmask), yi = (i32)(u32)cv, spi = (i32)((cv >> 32) & 0xff);
        const i32 navail = i < max_iter ? i : max_iter;         // candidates j in [i - navail, i - 1]
        const i32 kcap = navail < LPG_W ? navail : LPG_W;
        const bool more = navail > LPG_W;
        const i32 lower = i - navail;

        // The 32 window candidates are handled in blocks of LPG_B, each in two phases.
        // Phase A -- comput_sc(i, j) + f[j].  Nothing here depends on the loop's running state, so the evaluations
        // of a block (and their LDS lookups, all issued before the first is consumed) overlap freely, also with
        // phase B of the block before.
        //   S[k] = a value below NEG_BIG for a candidate that is out of reach / fails comput_sc
        // Phase B -- the sequential predecessor loop over those candidates.  All per-lane state is integer VGPR
        // state and every predicate is one compare feeding a select (no SGPR mask logic on the critical path):
        //   lim = INT32_MAX while the loop is live, INT32_MIN once it has stopped (or for a lane past the end of
        //         its group): min(S[k], lim) then makes every later candidate invalid, so nothing changes any more
        i32 n_reach = 0;
        i32 max_f = spi, max_k = -1, end_k = -1, lim = alive ? INT32_MAX : INT32_MIN;
        u32 n_skip = 0, marks = 0;
#pragma unroll
        for (int kb = 0; kb < LPG_W; kb += LPG_B) {
            i32 S[LPG_B];
            if (PENTAB) {
                i32 DG[LPG_B], PEN[LPG_B];
#pragma unroll
                for (int b = 0; b < LPG_B; ++b) {
                    const int k = kb + b;
                    const i32 dr = xi - WX[k], dq = yi - WY[k];
                    const i32 dg = dr < dq ? dr : dq;
                    const i32 mx = dr < dq ? dq : dr;
                    const u32 dd = (u32)(mx - dg);
                    DG[b] = dg;
                    PEN[b] = (i32)(dd < tabn ? dd : tabn);
                }
#pragma unroll
                for (int b = 0; b < LPG_B; ++b) PEN[b] = pen_tab[PEN[b]];
#pragma unroll
                for (int b = 0; b < LPG_B; ++b) {
                    const int k = kb + b;
                    const i32 dr = xi - WX[k], dq = yi - WY[k], spj = WS[k];
                    i32 s = (spj < DG[b] ? spj : DG[b]) - PEN[b] + WF[k];
                    s = (u32)(dq - 1) < dqlim ? s : SC_NONE;
                    s = (u32)(dr - 1) < (u32)maxdx ? s : SC_NONE;      // 1 <= dr <= max_dist_x: in reach and dr != 0
                    if (!FASTREACH) n_reach += dr <= maxdx ? 1 : 0;
                    S[b] = s;
                }
            } else {
#pragma unroll
                for (int b = 0; b < LPG_B; ++b) {   // same operations in the same order as comput_sc_dev
                    const int k = kb + b;
                    const i32 dr = xi - WX[k], dq = yi - WY[k], spj = WS[k];
                    const i32 dg = dr < dq ? dr : dq;
                    const i32 mx = dr < dq ? dq : dr;
                    const u32 dd = (u32)(mx - dg);
                    const i32 sc0 = spj < dg ? spj : dg;
                    const float lin_pen = pen_gap * (float)(i32)dd + pen_skip * (float)dg;
                    float log_pen = mg_log2_dev((float)(i32)(dd + 1));
                    log_pen = dd >= 1 ? log_pen : 0.0f;
                    const i32 pen = (i32)(lin_pen + .5f * log_pen);
                    i32 pen_ap = dg > spj ? pen : 0;
                    pen_ap = dd != 0 ? pen : pen_ap;
                    i32 s = sc0 - pen_ap + WF[k];
                    s = dd <= (u32)bw ? s : SC_NONE;
                    s = (u32)(dq - 1) < dqlim ? s : SC_NONE;
                    s = (u32)(dr - 1) < (u32)maxdx ? s : SC_NONE;
                    if (!FASTREACH) n_reach += dr <= maxdx ? 1 : 0;
                    S[b] = s;
                }
            }
#pragma unroll
            for (int b = 0; b < LPG_B; ++b) {
                const int k = kb + b;
                const i32 s = S[b] < lim ? S[b] : lim;
                const bool improve = s > max_f;
                const bool valid = s > NEG_BIG;
                const u32 bv = valid ? (marks >> k) & 1u : 0u;
                const u32 dec = n_skip ? n_skip - 1 : 0u;
                n_skip = improve ? dec : n_skip + bv;
                max_f = improve ? s : max_f;
                max_k = improve ? k : max_k;
                const bool brk = n_skip > (u32)max_skip;
                end_k = brk ? k : end_k;
                lim = brk ? INT32_MIN : lim;
                n_skip = brk ? 0u : n_skip;
                marks |= (valid ? WO[k] : 0u) << (k + 1);
            }
            if (kb / LPG_B < LPG_NPM) {          // (compile-time) the bound behind this block: see PM[] above
                const bool prune = lim == INT32_MAX && max_f >= PM[kb / LPG_B < LPG_NPM ? kb / LPG_B : 0];
                end_k = prune ? -2 : end_k;          // (-2: stopped by the bound, not by a max_skip break)
                lim = prune ? INT32_MIN : lim;
            }
            // every live lane's loop has stopped (bound reached, or max_skip break): the remaining blocks would change
            // nothing (min(S, lim) is invalid for all of them)
            if (FASTREACH && kb + LPG_B < LPG_W && __ballot(lim == INT32_MAX) == 0) break;
        }
        i32 max_j = max_k < 0 ? -1 : i - 1 - max_k;
        const i32 end_b = end_k < 0 ? -1 : i - 1 - end_k;
        // end of the loop: a break, the window start (x out of reach / max_iter), or more candidates behind the window
        const bool broke = end_b >= 0, pruned = end_k == -2;
        n_reach = n_reach < kcap ? n_reach : kcap;               // empty slots may have counted as "in reach"
        i32 end_j = broke ? end_b : (FASTREACH ? -1 : i - 1 - n_reach);
        const bool cont = alive && !broke && !pruned && more && (FASTREACH ? xi - WX[LPG_W - 1] <= maxdx : n_reach == LPG_W);   // (more: i > 32, the window is full)
        if (__ballot(cont)) {
            // rare: a lane's loop runs past its 32-anchor window; continue that lane's loop through HBM
            slow_iters += cont ? LPG_W : 0;
            slow_entries += cont ? 1u : 0u;
            if (cont && (slow_iters > R.slow_budget + (u32)i * R.slow_rate || slow_entries > R.slow_entries + (R.slow_entry_every ? (u32)i / R.slow_entry_every : 0u))) abandoned = true;
            else if (cont) {
                const u32 stamp = (u32)i + 1;
                // marks the window candidates left on anchors behind the window (all were valid and reached)
                for (i32 k = 0; k < LPG_W; ++k) {
                    const i32 j = i - 1 - k;
                    const u64 v = gv[j];
                    if (comput_sc_dev(xi, yi, (i32)(gk[j] & rmask), (i32)(u32)v, (i32)((v >> 32) & 0xff), P) == SC_NONE) continue;
                    const i32 pj = grec_p(ld_u64_l2(grec + j));
                    if (pj >= 0 && pj < i - LPG_W) tmark[pj] = stamp;
                }
                drain_stores();
                i32 j = i - 1 - LPG_W;
                for (;; --j) {
                    if (j < lower) { end_j = j; break; }
                    if (++slow_iters > R.slow_budget + (u32)i * R.slow_rate) { abandoned = true; break; }
                    const i32 xj = (i32)(gk[j] & rmask);
                    if (xi - xj > maxdx) { end_j = j; break; }
                    const u64 v = gv[j], r = ld_u64_l2(grec + j);
                    const i32 sc = comput_sc_dev(xi, yi, xj, (i32)(u32)v, (i32)((v >> 32) & 0xff), P);
                    if (sc == SC_NONE) continue;
                    const i32 s = sc + grec_f(r);
                    if (s > max_f) { max_f = s; max_j = j; if (n_skip > 0) --n_skip; }
                    else if (ld_u32_l2(tmark + j) == stamp) { if (++n_skip > (u32)max_skip) { end_j = j; break; } }
                    const i32 pj = grec_p(r);
                    if (pj >= 0) { tmark[pj] = stamp; drain_stores(); }
                }
            }
        }
        // ---- max_ii bookkeeping (mm2:lchain.c, the "best f in reach" shortcut) ----
        const bool need_rescan = alive && (mi < 0 || xi - mi_x > maxdx);
        if (__ballot(need_rescan)) {
            i32 bf = 0, bj = -1;      // f > 0 always
#pragma unroll
            for (int k = 0; k < LPG_W; ++k) {
                const bool in = xi - WX[k] <= maxdx && WF[k] > bf;
                bf = in ? WF[k] : bf;
                bj = in ? i - 1 - k : bj;
            }
            if (need_rescan) {
                if (more && xi - WX[LPG_W - 1] <= maxdx && bf < PF) {
                    drain_stores();
                    for (i32 j = i - 1 - LPG_W; j >= lower; --j) {
                        if (++slow_iters > R.slow_budget + (u32)i * R.slow_rate) { abandoned = true; break; }
                        if (xi - (i32)(gk[j] & rmask) > maxdx) break;
                        const i32 f = grec_f(ld_u64_l2(grec + j));
                        if (f > bf) { bf = f; bj = j; }
                    }
                }
                if (bj < 0) mi = -1;
                else {
                    const u64 k = gk[bj], v = gv[bj];
                    mi = bj; mi_f = bf; mi_x = (i32)(k & rmask); mi_y = (i32)(u32)v; mi_sp = (i32)((v >> 32) & 0xff);
                }
            }
        }
        const bool shortcut = alive && !pruned && mi >= 0 && mi < end_j;
        if (__ballot(shortcut)) {
            const i32 tmp = comput_sc_dev(xi, yi, mi_x, mi_y, mi_sp, P);
            if (shortcut && tmp != SC_NONE && max_f < tmp + mi_f) { max_f = tmp + mi_f; max_j = mi; }
        }
        if (alive && (mi < 0 || (xi - mi_x <= maxdx && mi_f < max_f))) { mi = i; mi_x = xi; mi_y = yi; mi_f = max_f; mi_sp = spi; }
        if (alive) {
            ring_o[r4 * 64 + lane] = grec_make(max_f, max_j);
            const u64 key = (u64)(u32)max_f << 32 | (u32)i;
            bkey = (max_f >= min_sc && key > bkey) ? key : bkey;
        }
        if (r4 == LPG_CH - 1) {
            if (alive) {                                             // a full chunk: one 32-byte run per lane
#pragma unroll
                for (int t = 0; t < LPG_CH; t += 2) {
                    ulonglong2 o; o.x = ring_o[t * 64 + lane]; o.y = ring_o[(t + 1) * 64 + lane];
                    *(ulonglong2 *)(grec + i - (LPG_CH - 1) + t) = o;
                }
            }
        } else if (__ballot(alive && i == n - 1)) {                 // a group ends inside its chunk
            if (alive && i == n - 1)
                for (i32 t = 0; t <= r4; ++t) grec[i - r4 + t] = ring_o[t * 64 + lane];
        }
        // the bounds of the next step: the slots that leave the first / the first two blocks when the window moves on
        if (alive) {
#pragma unroll
            for (int k = 0; k < LPG_NPM; ++k) { const i32 g = WF[(k + 1) * LPG_B - 1] + WS[(k + 1) * LPG_B - 1]; PM[k] = g > PM[k] ? g : PM[k]; }
            PF = WF[LPG_W - 1] > PF ? WF[LPG_W - 1] : PF;
        }
        // shift the window, insert anchor i at slot 0
        const u32 reli = (u32)(i - 1 - max_j);                       // >= 32 (or "no predecessor"): no mark inside the window
        const u32 oh = (max_j >= 0 && reli < 32u) ? 1u << reli : 0u;
#pragma unroll
        for (int k = LPG_W - 1; k > 0; --k) { WX[k] = WX[k - 1]; WY[k] = WY[k - 1]; WF[k] = WF[k - 1]; WS[k] = WS[k - 1]; WO[k] = WO[k - 1]; }
        WX[0] = xi; WY[0] = yi; WF[0] = max_f; WS[0] = spi; WO[0] = oh;
    }
    drain_stores();

    // ---------------- backtrack ----------------
    // Per lane: the best chain end and its walk (mg_chain_bk_end).  Nothing has been claimed yet, so this
    // is exactly the first iteration of backtrack_group(); anything it cannot settle falls back to it.
    const bool need_records = out.chains != nullptr || P.remove_internal != 0;
    u32 flags = 0;
    bool fb = false;
    if (has && !abandoned && bkey != 0) {
        const i32 zx = (i32)(u32)(bkey >> 32), top = (i32)(u32)bkey;
        i32 i = top, max_i = top, max_s = 0, depth = 0, cnt = 0;
        // A dependent load per hop would be the critical path of a long group.  Chains mostly step 1..3 anchors
        // back, so each round trip fetches the 8 records ending at the current anchor and the lane then hops
        // inside them without touching memory; all lanes of the wave refill in the same iteration.
        bool done = false;
        i32 cur = top;                                   // anchor whose record is needed next
        while (!done) {
            u64 blk[8];
            const i32 lo = cur > 7 ? cur - 7 : 0;
#pragma unroll
            for (int t = 0; t < 8; ++t) blk[t] = lo + t <= cur ? ld_u64_l2(grec + lo + t) : 0;
            while (!done && cur >= lo) {
                const i32 d = cur - lo;
                u64 r = blk[0];
#pragma unroll
                for (int t = 1; t < 8; ++t) r = d == t ? blk[t] : r;
                if (cur != top) {                        // (the record of `top` itself only supplies p)
                    const i32 s = zx - grec_f(r);
                    if (s > max_s) { max_s = s; max_i = cur; cnt = depth; }
                    else if (max_s - s > P.max_drop) { done = true; break; }
                }
                i = grec_p(r);
                ++depth;
                if (i < 0) {                             // walked off the chain start: s = zx
                    if (zx > max_s) { max_s = zx; max_i = -1; cnt = depth; }
                    done = true;
                }
                cur = i;
            }
        }
        const i32 sc = max_i == top ? 0 : max_s;
        const bool accepted = sc >= P.min_sc && cnt > 0 && cnt >= P.min_cnt;
        if (accepted && !need_records) flags = 3u;
        else fb = true;
    }
    u64 fbm = __ballot(fb);
    while (fbm) {
        const i32 l = (i32)__ffsll((unsigned long long)fbm) - 1;
        fbm &= fbm - 1;
        const u32 s0l = __builtin_amdgcn_readlane(s0, l);
        const i32 nl = __builtin_amdgcn_readlane(n, l);
        const u64 k0 = R.akey[s0l];
        const u32 rev = RFL((u32)(k0 >> P.kl.sh_rev()) & 1);
        const u32 rid = RFL((u32)(k0 >> P.kl.sh_rid()) & ((1u << P.kl.bits_rid) - 1));
        const u32 qid = RFL(P.q0 + (u32)(k0 >> P.kl.sh_q()));
        const u32 f = backtrack_group(R.akey + s0l, R.aval + s0l, R.grec + s0l, nl, rmask, qid, rid, rev, P, out);
        if ((i32)lane_id() == l) flags = f;
    }
    if (has && !abandoned) out.flags[g] = flags;
    if (abandoned) R.redo_list[atomicAdd(R.redo_count, 1u)] = g;
}
