// k_chain_hw.h -- K6, half-wave form: TWO groups per wavefront.
//
// k_chain_reg (one group per wave) is bound by instruction issue: ~141 VALU + ~83 SALU per anchor, with
// one scalar ALU per CU.  The predecessor loop needs <= 32 candidates for ~100 % of anchors, so half
// a wavefront is enough for a group: lanes 0..31 chain group A, lanes 32..63 chain group B, in
// lockstep on the anchor index i.  What was wave-uniform scalar state (max_f, max_j, end_j, max_ii and
// its anchor, window start) becomes VGPR state that is uniform per HALF, and the break / improve mask
// logic runs on the VALU (v_ffbl / v_ffbh on each half's 32-bit mask word), so one instruction stream
// now serves two anchors.  DPP row_shr + row_bcast:15 give 32-lane scans for both halves at once.
// Groups are paired in size order (sorted list), so the two halves finish together.
// Everything rare (candidates beyond the 32-anchor window, long max_ii rescans) and the backtrack run
// per half through the same out-of-line / wave-wide code k_chain_reg uses.
#pragma once
#include "k_chain_common.h"

struct HwChainArgs {
    const u64 *akey, *aval;
    const u32 *gstart;
    u32 n_groups; u64 n_anchors;
    const u32 *list;   // chained groups, largest first
    u32 n_list;
    u64 *grec;         // [n_anchors]
    u32 *tmark;        // [n_anchors] zero-initialised
    u32 prio;          // wavefront issue priority (0..3) while it runs beside k_chain_lpg
};

// inclusive max-scan inside each 32-lane half
__device__ __forceinline__ i32 half_incl_max_i32(i32 v, i32 identity) {
    i32 o;
    o = __builtin_amdgcn_update_dpp(identity, v, DPP_ROW_SHR(1), 0xf, 0xf, false); v = v > o ? v : o;
    o = __builtin_amdgcn_update_dpp(identity, v, DPP_ROW_SHR(2), 0xf, 0xf, false); v = v > o ? v : o;
    o = __builtin_amdgcn_update_dpp(identity, v, DPP_ROW_SHR(4), 0xf, 0xf, false); v = v > o ? v : o;
    o = __builtin_amdgcn_update_dpp(identity, v, DPP_ROW_SHR(8), 0xf, 0xf, false); v = v > o ? v : o;
    o = __builtin_amdgcn_update_dpp(identity, v, DPP_BCAST15, 0xa, 0xf, false); v = v > o ? v : o;
    return v;
}
// shift by one lane inside each half; lanes 0 and 32 receive `fill`
__device__ __forceinline__ i32 half_shr1_i32(i32 v, i32 fill, i32 hl) {
    const i32 o = __builtin_amdgcn_update_dpp(fill, v, DPP_WAVE_SHR1, 0xf, 0xf, false);
    return hl == 0 ? fill : o;
}

// A 64-record block of grec held one record per lane, so that the (wave-uniform) pointer chase of the
// backtrack hops through registers: chains mostly step 1..3 anchors back, and a dependent HBM/L2 load per
// hop (~1 us) was the critical path of long groups.  get(i) reloads the block (coalesced, ending at i)
// only when i falls outside it.  The block is a snapshot: reset() after any store to grec.
struct RecBlock {
    u64 v; i32 lo;
    __device__ __forceinline__ void reset() { lo = INT32_MAX; }
    __device__ __forceinline__ u64 get(const u64 *grec, i32 n, i32 i) {
        if (i < lo || i >= lo + 64) {
            lo = i > 63 ? i - 63 : 0;
            const i32 idx = lo + (i32)lane_id();
            v = idx < n ? ld_u64_l2(grec + idx) : 0;
        }
        const i32 d = i - lo;
        return (u64)(u32)__builtin_amdgcn_readlane((i32)(u32)v, d) | (u64)(u32)__builtin_amdgcn_readlane((i32)(u32)(v >> 32), d) << 32;
    }
};

// mg_chain_backtrack for one group, wave-wide (identical to the second phase of k_chain_reg)
__device__ __noinline__ u32 backtrack_group(const u64 *gk, const u64 *gv, u64 *grec, i32 n, u64 rmask, u32 qid, u32 rid,
                                            u32 rev, ChainParams P, GroupOut out) {
    const i32 lane = (i32)lane_id();
    u8 *gstate = (u8 *)grec;
    u32 flags = 0;
    const i32 qlen = (i32)P.q_len[qid], tlen = (i32)P.t_len[rid];
    const bool need_records = out.chains != nullptr || P.remove_internal != 0;
    for (;;) {
        u64 best = 0;
        for (i32 i = n - 1 - lane; i >= 0; i -= 64) {
            const u64 r = ld_u64_l2(grec + i);
            const i32 fi = grec_f(r);
            if (fi >= P.min_sc && grec_state(r) == 0) { const u64 key = (u64)(u32)fi << 32 | (u32)i; best = key > best ? key : best; }
        }
        best = wave_max_u64(best);
        const i32 zx = (i32)RFL((u32)(best >> 32));
        if (zx == 0) break;
        const i32 top = (i32)RFL((u32)best);
        i32 i = top, max_i = top, max_s = 0, depth = 0, cnt = 0;
        RecBlock blk; blk.reset();
        u64 r = blk.get(grec, n, top);
        for (;;) {   // mg_chain_bk_end (its t[]=2 marks are dead: p[i] < i)
            i = RFL(grec_p(r));
            ++depth;
            i32 s;
            if (i < 0) s = zx;
            else { r = blk.get(grec, n, i); s = zx - RFL(grec_f(r)); }
            if (s > max_s) { max_s = s; max_i = i; cnt = depth; }
            else if (max_s - s > P.max_drop) break;
            if (i < 0 || (RFL(grec_state(r)) & 3) != 0) break;
        }
        const i32 sc = max_i == top ? 0 : max_s;
        const bool accepted = sc >= P.min_sc && cnt > 0 && cnt >= P.min_cnt;
        if (accepted && !need_records) { flags = 3u; break; }
        i32 first = top, mlen = 0, blen = 0;
        for (i = top; i != max_i;) {
            if (lane == 0) gstate[(u64)i * 8 + 7] = 1;
            first = i;
            const i32 pi = RFL(grec_p(blk.get(grec, n, i)));   // (p is never rewritten: the snapshot stays valid for it)
            if (accepted && pi != max_i) {
                const u64 ki = gk[i], vi = gv[i], kp = gk[pi], vp = gv[pi];
                const i32 span = RFL((i32)((vi >> 32) & 0xff));
                const i32 tl = RFL((i32)(ki & rmask)) - RFL((i32)(kp & rmask)), ql = RFL((i32)(u32)vi) - RFL((i32)(u32)vp);
                blen += tl > ql ? tl : ql;
                mlen += (tl > span && ql > span) ? span : (tl < ql ? tl : ql);
            }
            i = pi;
        }
        if (cnt == 0 && lane == 0) gstate[(u64)top * 8 + 7] = 4;
        drain_stores();
        if (accepted) {
            const u64 kf = gk[first], vf = gv[first], kt = gk[top], vt = gv[top];
            const i32 fx = RFL((i32)(kf & rmask)), fy = RFL((i32)(u32)vf), q_span = RFL((i32)((vf >> 32) & 0xff));
            const i32 tx = RFL((i32)(kt & rmask)), ty = RFL((i32)(u32)vt);
            const i32 rs = fx + 1 > q_span ? fx + 1 - q_span : 0;
            const i32 re = tx + 1;
            i32 qs, qe;
            if (!rev) { qs = fy + 1 - q_span; qe = ty + 1; }
            else { qs = qlen - (ty + 1); qe = qlen - (fy + 1 - q_span); }
            mlen += q_span; blen += q_span;
            bool keep = true;
            if (P.remove_internal) {
                i32 overhang = !rev ? min(qs, rs) + min(qlen - qe, tlen - re) : min(qs, tlen - re) + min(qlen - qe, rs);
                i32 maplen = max(qe - qs, re - rs);
                if (P.remove_internal == 1) {
                    float ratio = (float)overhang / (float)maplen;
                    if (ratio < P.max_overhang_ratio) keep = false;
                } else {
                    float prod = (float)maplen * P.max_overhang_ratio;
                    i32 lim = prod != prod ? 0 : (prod >= 2147483648.0f ? INT32_MAX : (prod <= -2147483648.0f ? INT32_MIN : (i32)prod));
                    if (overhang > lim) keep = false;
                }
            }
            flags |= 1u | (keep ? 2u : 0u);
            if (out.chains && lane == 0) {
                unsigned long long slot = atomicAdd(out.n_chains, 1ULL);
                if (slot < out.chain_cap) {
                    lrge_hip_chain c;
                    c.query = qid; c.target = rid + out.rid_base; c.rev = (i32)rev; c.score = sc; c.cnt = cnt;
                    c.qs = qs; c.qe = qe; c.rs = rs; c.re = re; c.mlen = mlen; c.blen = blen;
                    {   // entries of minimap2's mini_pos[] spanned by the chain (kept-seed ranks ride in the anchor values)
                        const i32 r0 = (i32)(vf >> AVAL_RANK_SHIFT), r1 = (i32)(vt >> AVAL_RANK_SHIFT);
                        c.n_seeds = (r1 > r0 ? r1 - r0 : r0 - r1) + 1;
                    }
                    out.chains[slot] = c;
                }
            }
            if (!P.want_all && (flags & 2u)) break;
        }
    }
    return flags;
}

// one wavefront, the pair of groups list[ia], list[ia + 1]
__device__ __forceinline__ void chain_hw_pair(const HwChainArgs &R, const ChainParams &P, const GroupOut &out, u32 ia, u32 n_list,
                                              bool clear_tmark) {
    const u32 ib = ia + 1;
    if (ia >= n_list) return;
    const i32 lane = (i32)lane_id();
    const i32 h = lane >> 5, hl = lane & 31, hbase = lane & 32;
    const bool hasB = ib < n_list;
    const u32 gA = RFL(R.list[ia]), gB = hasB ? RFL(R.list[ib]) : gA;
    const u32 s0A = RFL(R.gstart[gA]), s0B = RFL(R.gstart[gB]);
    const u32 e0A = (gA + 1 < R.n_groups) ? RFL(R.gstart[gA + 1]) : (u32)R.n_anchors;
    const u32 e0B = (gB + 1 < R.n_groups) ? RFL(R.gstart[gB + 1]) : (u32)R.n_anchors;
    const i32 nA = (i32)(e0A - s0A), nB = hasB ? (i32)(e0B - s0B) : 0;
    const i32 n_max = nA > nB ? nA : nB;
    if (clear_tmark) {   // groups another kernel gave up half way: its stamps must not be mistaken for ours
        for (i32 t = lane; t < nA; t += 64) R.tmark[s0A + t] = 0;
        for (i32 t = lane; t < nB; t += 64) R.tmark[s0B + t] = 0;
        drain_stores();
    }
    if (R.prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (R.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (R.prio == 1) __builtin_amdgcn_s_setprio(1);
    const u64 rmask = (1ULL << P.kl.bits_rpos) - 1;
    // per-lane view of "my" group
    const u32 s0 = h ? s0B : s0A;
    const i32 n = h ? nB : nA;
    const u64 *gk = R.akey + s0, *gv = R.aval + s0;
    u64 *grec = R.grec + s0;

    const i32 maxdx = P.max_dist_x, bw = P.bw, max_skip = P.max_skip, max_iter = P.max_iter;
    const u32 dqlim = (u32)(P.max_dist_x < P.max_dist_y ? P.max_dist_x : P.max_dist_y);
    const float pen_gap = P.pen_gap, pen_skip = P.pen_skip;
    i32 wx = 0, wy = 0, wf = 0, wp = -1, ws = 0;     // window: lane hl <-> anchor i-1-hl of my group
    i32 sx = 0, sy = 0, ss = 0;                      // staging in:  lane hl <-> anchor (i & ~31) + hl
    i32 of = 0, op = -1;                             // staging out
    i32 mi = -1, mi_x = 0, mi_y = 0, mi_f = 0, mi_sp = 0;   // max_ii and its anchor (uniform per half)

    for (i32 i = 0; i < n_max; ++i) {
        const i32 il = i & 31;
        if (il == 0) {
            const i32 idx = i + hl;
            if (idx < n) { const u64 k = gk[idx], v = gv[idx]; sx = (i32)(k & rmask); sy = (i32)(u32)v; ss = (i32)((v >> 32) & 0xff); }
        }
        const bool alive = i < n;
        const i32 xiA = __builtin_amdgcn_readlane(sx, il), xiB = __builtin_amdgcn_readlane(sx, 32 + il);
        const i32 yiA = __builtin_amdgcn_readlane(sy, il), yiB = __builtin_amdgcn_readlane(sy, 32 + il);
        const i32 spA = __builtin_amdgcn_readlane(ss, il), spB = __builtin_amdgcn_readlane(ss, 32 + il);
        const i32 xi = h ? xiB : xiA, yi = h ? yiB : yiA, spi = h ? spB : spA;
        const i32 lim = i - 1 < max_iter - 1 ? i - 1 : max_iter - 1;
        const i32 lower = i - max_iter > 0 ? i - max_iter : 0;
        // reach (a prefix of each half's lanes)
        i32 actv = wx + (maxdx - xi);
        actv = (hl <= lim && alive) ? actv : -1;
        const u64 act_mask = __ballot(actv >= 0);
        const i32 n_act = h ? (i32)__popc((u32)(act_mask >> 32)) : (i32)__popc((u32)act_mask);
        // comput_sc, branch-free
        const i32 dq = yi - wy, dr = xi - wx;
        const i32 df = dr - dq;
        const i32 dd = df < 0 ? -df : df;
        const i32 dg = dr < dq ? dr : dq;
        i32 sc = ws < dg ? ws : dg;
        const float lin_pen = pen_gap * (float)dd + pen_skip * (float)dg;
        float log_pen = mg_log2_dev((float)(dd + 1));
        log_pen = dd >= 1 ? log_pen : 0.0f;
        const i32 pen = (i32)(lin_pen + .5f * log_pen);
        sc = (dd != 0 || dg > ws) ? sc - pen : sc;
        i32 s = sc + wf;
        s = (u32)(dq - 1) < dqlim ? s : SC_NONE;
        s = dr != 0 ? s : SC_NONE;
        s = dd <= bw ? s : SC_NONE;
        s = actv >= 0 ? s : SC_NONE;
        // marks: push a flag to the lane of my half that holds my predecessor
        const i32 tl = (i - 1) - wp;
        const bool in_reg = s != SC_NONE && (u32)tl < 32u && wp >= 0;
        const i32 got = __builtin_amdgcn_ds_permute((hbase + (in_reg ? tl : 0)) << 2, in_reg ? 1 : 0);
        // running maximum before each lane, inside the half
        i32 exc = half_shr1_i32(half_incl_max_i32(s, SC_NONE), SC_NONE, hl);
        exc = exc > spi ? exc : spi;
        const bool improve = s > exc;
        const u64 im_all = __ballot(improve);
        const i32 bumpv = (s != SC_NONE && s <= exc && got != 0) ? 1 : 0;
        const u64 bm_all = __ballot(bumpv != 0);
        // n_skip after each lane (Lindley recursion, see k_chain_reg)
        const i32 cb = h ? (i32)__builtin_amdgcn_mbcnt_hi((u32)(bm_all >> 32), 0) : (i32)__builtin_amdgcn_mbcnt_lo((u32)bm_all, 0);
        i32 ns = cb + bumpv;
        if (im_all != 0) {
            const i32 ci = h ? (i32)__builtin_amdgcn_mbcnt_hi((u32)(im_all >> 32), 0) : (i32)__builtin_amdgcn_mbcnt_lo((u32)im_all, 0);
            const i32 S = ns - (ci + (improve ? 1 : 0));
            i32 mn = half_incl_max_i32(-S, SC_NONE);
            mn = mn > 0 ? mn : 0;
            ns = S + mn;
        }
        const u64 brk = __ballot(bumpv != 0 && ns > max_skip);
        // per-half resolution on the VALU: each half looks at its own 32-bit mask word
        const u32 brk_w = h ? (u32)(brk >> 32) : (u32)brk;
        const u32 im_w = h ? (u32)(im_all >> 32) : (u32)im_all;
        const bool has_brk = brk_w != 0;
        const i32 bl = has_brk ? (i32)__ffs((int)brk_w) - 1 : 0;
        const u32 im_c = has_brk ? (im_w & ((2u << bl) - 1u)) : im_w;
        const bool has_im = im_c != 0;
        const i32 L = has_im ? 31 - (i32)__clz((int)im_c) : 0;
        const i32 sL = __builtin_amdgcn_ds_bpermute((hbase + L) << 2, s);
        i32 max_f = has_im ? sL : spi;
        i32 max_j = has_im ? i - 1 - L : -1;
        i32 end_j = has_brk ? i - 1 - bl : i - n_act - 1;
        i32 st = i - n_act;                                  // -1 below = not determined
        const bool beyond = alive && n_act == 32 && i - 33 >= lower;
        if (beyond) st = -1;
        const u64 need_tail = __ballot(beyond && !has_brk);
        if (need_tail) {
            // rare: a half's loop runs past its 32-anchor window.  Serve each such half wave-wide.
#pragma unroll 1
            for (i32 hh = 0; hh < 2; ++hh) {
                if (!((need_tail >> (32 * hh)) & 1)) continue;
                GroupView V; V.gk = R.akey + (hh ? s0B : s0A); V.gv = R.aval + (hh ? s0B : s0A);
                V.grec = R.grec + (hh ? s0B : s0A); V.tmark = R.tmark + (hh ? s0B : s0A); V.rmask = rmask;
                const bool far_push = h == hh && s != SC_NONE && wp >= 0 && tl >= 32;
                const SlowTail r = chain_slow_tail(V, P, i, hh ? xiB : xiA, hh ? yiB : yiA, lower, far_push, wp,
                                                   __builtin_amdgcn_readlane(max_f, 32 * hh), __builtin_amdgcn_readlane(max_j, 32 * hh),
                                                   __builtin_amdgcn_readlane(ns, 32 * hh + 31), i - 33);
                if (h == hh) { st = r.st; max_f = r.max_f; max_j = r.max_j; end_j = r.end_j; }
            }
        }
        // ---- max_ii bookkeeping ----
        const bool need_rescan = alive && (mi < 0 || xi - mi_x > maxdx);
        const u64 rescan_m = __ballot(need_rescan);
        if (rescan_m) {
            // best f among the in-reach lanes of the half; ties keep the larger j = the smaller lane
            const i32 fm = actv >= 0 ? wf : 0;                 // f > 0 always
            const i32 fsc = half_incl_max_i32(fm, 0);
            const i32 fmax = h ? __builtin_amdgcn_readlane(fsc, 63) : __builtin_amdgcn_readlane(fsc, 31);
            const u64 eq = __ballot(fm == fmax && fmax > 0);
            const u32 eq_w = h ? (u32)(eq >> 32) : (u32)eq;
            i32 best_f = fmax, best_j = eq_w ? i - 1 - ((i32)__ffs((int)eq_w) - 1) : -1;
            i32 bx = 0, by = 0, bsp = 0;
            {
                const i32 d = best_j >= 0 ? i - 1 - best_j : 0;
                bx = __builtin_amdgcn_ds_bpermute((hbase + d) << 2, wx);
                by = __builtin_amdgcn_ds_bpermute((hbase + d) << 2, wy);
                bsp = __builtin_amdgcn_ds_bpermute((hbase + d) << 2, ws);
            }
            const u64 far_m = __ballot(need_rescan && beyond);
            if (far_m) {
#pragma unroll 1
                for (i32 hh = 0; hh < 2; ++hh) {
                    if (!((far_m >> (32 * hh)) & 1)) continue;
                    GroupView V; V.gk = R.akey + (hh ? s0B : s0A); V.gv = R.aval + (hh ? s0B : s0A);
                    V.grec = R.grec + (hh ? s0B : s0A); V.tmark = R.tmark + (hh ? s0B : s0A); V.rmask = rmask;
                    u64 key = (h == hh && actv >= 0) ? ((u64)(u32)wf << 32 | (u32)(i - 1 - hl)) : 0;
                    key = chain_slow_rescan(V, P, i, hh ? xiB : xiA, lower, __builtin_amdgcn_readlane(st, 32 * hh), key, i - 33);
                    key = wave_max_u64(key);
                    const i32 kf = (i32)RFL((u32)(key >> 32)), kj = (i32)RFL((u32)key);
                    i32 kx = 0, ky = 0, ksp = 0;
                    if (kf > 0 && i - 1 - kj >= 32) {           // the winner lives behind the window
                        const u64 k = V.gk[kj], v = V.gv[kj];
                        kx = RFL((i32)(k & rmask)); ky = RFL((i32)(u32)v); ksp = RFL((i32)((v >> 32) & 0xff));
                        if (h == hh) { best_f = kf; best_j = kj; bx = kx; by = ky; bsp = ksp; }
                    } else if (h == hh) { best_f = kf; best_j = kf > 0 ? kj : -1; }   // inside the window: bx/by/bsp already hold it
                }
            }
            if (need_rescan) {
                if (best_j < 0) mi = -1;
                else { mi = best_j; mi_f = best_f; mi_x = bx; mi_y = by; mi_sp = bsp; }
            }
        }
        const bool shortcut = alive && mi >= 0 && mi < end_j;
        if (__ballot(shortcut)) {
            const i32 tmp = comput_sc_dev(xi, yi, mi_x, mi_y, mi_sp, P);
            if (shortcut && tmp != SC_NONE && max_f < tmp + mi_f) { max_f = tmp + mi_f; max_j = mi; }
        }
        if (alive && (mi < 0 || (xi - mi_x <= maxdx && mi_f < max_f))) { mi = i; mi_x = xi; mi_y = yi; mi_f = max_f; mi_sp = spi; }
        // results -> staging lane; coalesced flush every 32 anchors and at the end of the group
        of = hl == il ? max_f : of;
        op = hl == il ? max_j : op;
        if (alive && (il == 31 || i == n - 1) && hl <= il) grec[(i & ~31) + hl] = grec_make(of, op);
        // shift the window inside each half, insert anchor i at the half's lane 0
        wx = __builtin_amdgcn_update_dpp(xi, wx, DPP_WAVE_SHR1, 0xf, 0xf, false); wx = hl == 0 ? xi : wx;
        wy = __builtin_amdgcn_update_dpp(yi, wy, DPP_WAVE_SHR1, 0xf, 0xf, false); wy = hl == 0 ? yi : wy;
        wf = __builtin_amdgcn_update_dpp(max_f, wf, DPP_WAVE_SHR1, 0xf, 0xf, false); wf = hl == 0 ? max_f : wf;
        wp = __builtin_amdgcn_update_dpp(max_j, wp, DPP_WAVE_SHR1, 0xf, 0xf, false); wp = hl == 0 ? max_j : wp;
        ws = __builtin_amdgcn_update_dpp(spi, ws, DPP_WAVE_SHR1, 0xf, 0xf, false); ws = hl == 0 ? spi : ws;
    }
    drain_stores();

    // ---------------- backtrack: one group at a time, wave-wide ----------------
#pragma unroll 1
    for (i32 hh = 0; hh < 2; ++hh) {
        if (hh == 1 && !hasB) break;
        const u32 g = hh ? gB : gA, s0h = hh ? s0B : s0A;
        const i32 nh = hh ? nB : nA;
        const u64 k0 = R.akey[s0h];
        const u32 rev = RFL((u32)(k0 >> P.kl.sh_rev()) & 1);
        const u32 rid = RFL((u32)(k0 >> P.kl.sh_rid()) & ((1u << P.kl.bits_rid) - 1));
        const u32 qid = RFL(P.q0 + (u32)(k0 >> P.kl.sh_q()));
        const u32 flags = backtrack_group(R.akey + s0h, R.aval + s0h, R.grec + s0h, nh, rmask, qid, rid, rev, P, out);
        if (lane == 0) out.flags[g] = flags;
    }
}

__global__ __launch_bounds__(64) void k_chain_hw(HwChainArgs R, ChainParams P, GroupOut out) {
    chain_hw_pair(R, P, out, 2 * blockIdx.x, R.n_list, false);
}

// the groups k_chain_lpg gave up (see LpgChainArgs): their number only exists on the device, so a fixed grid strides
// over the list
__global__ __launch_bounds__(64) void k_chain_hw_redo(HwChainArgs R, ChainParams P, GroupOut out, const u32 *__restrict__ d_n_list) {
    const u32 n_list = RFL(*d_n_list);
    for (u32 ia = 2 * blockIdx.x; ia < n_list; ia += 2 * gridDim.x) chain_hw_pair(R, P, out, ia, n_list, true);
}

// ------------------------------------------------------------------------------------------
// list of chained groups with their sizes (for the size sort that pairs groups for k_chain_hw)
// ------------------------------------------------------------------------------------------
// Size census of the groups worth chaining (n >= min_n): n_out[0] = how many, anchors_out[0] = their anchors, and a
// histogram over size classes of GSZ_W anchors -- class c holds (c*GSZ_W, (c+1)*GSZ_W], the last class everything
// above -- as counts hist_n[] and anchor sums hist_a[].  The host picks the k_chain_hw / k_chain_lpg split from it.
// (n_out[1] is k_group_fill's cursor.)
#define GSZ_W 64
#define GSZ_BINS 257
__device__ __forceinline__ u32 gsz_class(u32 n) { const u32 c = (n - 1) / GSZ_W; return c < GSZ_BINS - 1 ? c : GSZ_BINS - 1; }

// (the number of groups is read from device memory: the census is launched before the host knows it, grid-stride)
__global__ void k_group_count(const u32 *__restrict__ gstart, const u32 *__restrict__ d_n_groups, u64 n_anchors, u32 min_n,
                              u32 *__restrict__ n_out, unsigned long long *__restrict__ anchors_out,
                              u32 *__restrict__ hist_n, unsigned long long *__restrict__ hist_a) {
    __shared__ u32 lc; __shared__ unsigned long long la;
    __shared__ u32 hn[GSZ_BINS]; __shared__ unsigned long long ha[GSZ_BINS];
    const u32 n_groups = *d_n_groups;
    if (threadIdx.x == 0) { lc = 0; la = 0; }
    for (u32 i = threadIdx.x; i < GSZ_BINS; i += blockDim.x) { hn[i] = 0; ha[i] = 0; }
    __syncthreads();
    u32 c = 0; unsigned long long a = 0;
    for (u64 gg = (u64)blockIdx.x * blockDim.x + threadIdx.x; gg < n_groups; gg += (u64)gridDim.x * blockDim.x) {
        const u32 g = (u32)gg;
        const u64 e = (g + 1 < n_groups) ? gstart[g + 1] : n_anchors;
        const u32 n = (u32)(e - gstart[g]);
        if (n >= min_n && n > 0) { ++c; a += n; const u32 k = gsz_class(n); atomicAdd(&hn[k], 1u); atomicAdd(&ha[k], (unsigned long long)n); }
    }
    if (c) { atomicAdd(&lc, c); atomicAdd(&la, a); }
    __syncthreads();
    if (threadIdx.x == 0 && lc) { atomicAdd(n_out, lc); atomicAdd(anchors_out, la); }
    for (u32 i = threadIdx.x; i < GSZ_BINS; i += blockDim.x)
        if (hn[i]) { atomicAdd(&hist_n[i], hn[i]); atomicAdd(&hist_a[i], ha[i]); }
}

// keys = 65535 - min(n, 65535), so that an ascending 16-bit sort puts the largest groups first; vals = group id
__global__ void k_group_fill(const u32 *__restrict__ gstart, u32 n_groups, u64 n_anchors, u32 min_n,
                             u32 *__restrict__ cursor, u64 *__restrict__ keys, u64 *__restrict__ vals) {
    __shared__ u32 lc, lbase, lfill;
    if (threadIdx.x == 0) { lc = 0; lfill = 0; }
    __syncthreads();
    const u64 c0 = (u64)blockIdx.x * GB_CHUNK;
    for (int pass = 0; pass < 2; ++pass) {
        for (u64 gg = c0 + threadIdx.x; gg < c0 + GB_CHUNK && gg < n_groups; gg += blockDim.x) {
            const u32 g = (u32)gg;
            const u64 e = (g + 1 < n_groups) ? gstart[g + 1] : n_anchors;
            const u32 n = (u32)(e - gstart[g]);
            if (n < min_n) continue;
            if (pass == 0) atomicAdd(&lc, 1u);
            else { const u32 o = lbase + atomicAdd(&lfill, 1u); keys[o] = 65535u - (n < 65535u ? n : 65535u); vals[o] = g; }
        }
        __syncthreads();
        if (pass == 0 && threadIdx.x == 0 && lc) lbase = atomicAdd(cursor, lc);
        __syncthreads();
    }
}

__global__ void k_vals_to_u32(const u64 *__restrict__ vals, u32 n, u32 *__restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (u32)vals[i];
}
