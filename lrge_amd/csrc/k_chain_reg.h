// k_chain_reg.h -- K6, register-window form (the hot kernel).
//
// Same arithmetic as chain_group<> in k_chain.h (mg_lchain_dp + mg_chain_backtrack), organised around
// what the data show: the predecessor loop of mg_lchain_dp ends after <= 32 candidates for ~all
// anchors (the max_skip break), so the candidates of anchor i are almost always the 64 anchors just
// before it.  One wavefront owns one (query, target, strand) group and keeps those 64 anchors in
// VGPRs -- lane L holds anchor i-1-L as {x, y, f, p|span} -- shifting them by one lane per step with
// a DPP wave_shr.  A step therefore needs no LDS or HBM read for its candidates:
//   * anchor i itself comes from a 64-anchor staging register filled by one coalesced HBM load every
//     64 steps (v_readlane with a scalar lane index);
//   * comput_sc runs on all lanes; the scalar loop's order-dependent state is resolved with DPP
//     scans exactly as in k_chain.h;
//   * the t[] marks ("candidate j is the predecessor of a candidate already visited") are one
//     ds_permute_b32: every valid candidate pushes a 1 to the lane that holds its predecessor;
//   * (f, p) go to HBM once (8 bytes per anchor) for the backtrack.
// No LDS is allocated, so occupancy is bounded by registers only and every group size runs in the
// same kernel.  When the 64-anchor window is exhausted without a break (or the max_ii shortcut has
// to rescan a longer window) an exact slow path continues through HBM.
#pragma once
#include "k_chain.h"

#define GREC_NONE 0xFFFFFFu   // "no predecessor" in the 24-bit p field

// per-anchor record written by the DP: f (bits 0..31) | p (bits 32..55) | backtrack state (bits 56..63)
__device__ __forceinline__ u64 grec_make(i32 f, i32 p) { return (u64)(u32)f | (u64)((u32)p & GREC_NONE) << 32; }
__device__ __forceinline__ i32 grec_f(u64 r) { return (i32)(u32)r; }
__device__ __forceinline__ i32 grec_p(u64 r) { u32 p = (u32)(r >> 32) & GREC_NONE; return p == GREC_NONE ? -1 : (i32)p; }
__device__ __forceinline__ u32 grec_state(u64 r) { return (u32)(r >> 56); }

// loads that must observe this wave's own earlier stores: bypass the per-CU L1 (served by L2)
__device__ __forceinline__ u64 ld_u64_l2(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 ld_u32_l2(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { __builtin_amdgcn_s_waitcnt(0x0070 | 0x0f00); }  // vmcnt(0)

struct RegChainArgs {
    const u64 *akey, *aval;
    const u32 *gstart;
    u32 n_groups; u64 n_anchors;
    const u32 *list;        // [N_BINS][n_groups] group ids per size bin
    u32 bin_first[N_BINS];  // block index at which each bin starts (largest bin first)
    u32 bin_of[N_BINS];     // which bin that is
    u32 n_blocks;
    u64 *grec;              // [n_anchors]
    u32 *tmark;             // [n_anchors] zero-initialised; slow-path t[] stamps
};

// resolve one 64-candidate chunk of the scalar predecessor loop.  Inputs are per lane (lane order =
// visiting order); scalar state is passed by reference.  Returns true when the loop broke.
__device__ __forceinline__ bool resolve_chunk(i32 s, bool valid, bool marked, i32 base, i32 max_skip, i32 &max_f,
                                              i32 &max_j, i32 &n_skip, i32 &end_j) {
    i32 exc = wave_shr1_i32(wave_incl_max_i32(s, SC_NONE), SC_NONE);
    exc = exc > max_f ? exc : max_f;
    const bool improve = valid && s > exc;
    const bool bump = valid && !improve && marked;
    const u64 im_all = __ballot(improve);
    const u64 bm_all = __ballot(bump);
    u64 brk;
    i32 ns_after;
    if (im_all == 0) {
        ns_after = n_skip + (i32)__builtin_amdgcn_mbcnt_hi((u32)(bm_all >> 32), __builtin_amdgcn_mbcnt_lo((u32)bm_all, 0)) + (bump ? 1 : 0);
        brk = __ballot(bump && ns_after > max_skip);
    } else {
        i32 a = improve ? -1 : (bump ? 1 : 0);
        i32 b = improve ? 0 : NEG_BIG;
        wave_incl_clampadd(a, b, NEG_BIG);
        ns_after = n_skip + a; ns_after = ns_after > b ? ns_after : b;
        brk = __ballot(bump && ns_after > max_skip);
    }
    u64 consider = ~0ULL;
    i32 bl = 64;
    if (brk) { bl = (i32)__ffsll((unsigned long long)brk) - 1; consider = (bl == 63) ? ~0ULL : ((1ULL << (bl + 1)) - 1); }
    const u64 im = im_all & consider;
    if (im) {
        const i32 L = 63 - (i32)__clzll((long long)im);
        max_f = __builtin_amdgcn_readlane(s, L);
        max_j = base - L;
    }
    if (brk) { end_j = base - bl; return true; }
    n_skip = __builtin_amdgcn_readlane(ns_after, 63);
    return false;
}

__global__ __launch_bounds__(64) void k_chain_reg(RegChainArgs R, ChainParams P, GroupOut out) {
    if (blockIdx.x >= R.n_blocks) return;
    const i32 lane = (i32)lane_id();
    // block -> (bin, slot): bins are laid out largest first so that long groups start early
    u32 bsel = 0;
#pragma unroll
    for (int b = 1; b < N_BINS; ++b) if (blockIdx.x >= R.bin_first[b]) bsel = b;
    const u32 g = RFL(R.list[(u64)R.bin_of[bsel] * R.n_groups + (blockIdx.x - R.bin_first[bsel])]);
    const u32 s0 = RFL(R.gstart[g]);
    const u32 e0 = (g + 1 < R.n_groups) ? RFL(R.gstart[g + 1]) : (u32)R.n_anchors;
    const i32 n = (i32)(e0 - s0);
    const u64 *gk = R.akey + s0, *gv = R.aval + s0;
    u64 *grec = R.grec + s0;
    u32 *tmark = R.tmark + s0;
    const u64 rmask = (1ULL << P.kl.bits_rpos) - 1;
    const u64 k0 = gk[0];
    const u32 rev = RFL((u32)(k0 >> P.kl.sh_rev()) & 1);
    const u32 rid = RFL((u32)(k0 >> P.kl.sh_rid()) & ((1u << P.kl.bits_rid) - 1));
    const u32 qid = RFL(P.q0 + (u32)(k0 >> P.kl.sh_q()));

    // ---------------- DP ----------------
    i32 wx = 0, wy = 0, wf = 0, wp = -1, ws = 0;     // window: lane L <-> anchor i-1-L
    i32 sx = 0, sy = 0, ss = 0;                      // staging: lane L <-> anchor (i & ~63) + L
    i32 mi = -1, mi_x = 0, mi_y = 0, mi_f = 0, mi_sp = 0;
    for (i32 i = 0; i < n; ++i) {
        if ((i & 63) == 0) {
            const i32 idx = i + lane;
            if (idx < n) { const u64 k = gk[idx], v = gv[idx]; sx = (i32)(k & rmask); sy = (i32)(u32)v; ss = (i32)((v >> 32) & 0xff); }
        }
        const i32 xi = __builtin_amdgcn_readlane(sx, i & 63), yi = __builtin_amdgcn_readlane(sy, i & 63),
                  spi = __builtin_amdgcn_readlane(ss, i & 63);
        const i32 lower = i - P.max_iter > 0 ? i - P.max_iter : 0;   // st >= lower (the max_iter clamp)
        const i32 j = i - 1 - lane;
        const bool act = j >= lower && wx + P.max_dist_x >= xi;      // x ascending: the active lanes are a prefix
        const i32 n_act = (i32)__popcll(__ballot(act));
        const i32 sc = comput_sc_dev(xi, yi, wx, wy, ws, P);
        const bool valid = act && sc != SC_NONE;
        const i32 s = valid ? sc + wf : SC_NONE;
        // t[p[j]] = i for every visited candidate: push a flag to the lane that holds p[j]
        const bool push = valid && wp >= 0;
        const i32 tl = lane + (j - wp);
        const bool in_reg = push && tl < 64;
        const i32 got = __builtin_amdgcn_ds_permute((in_reg ? tl : 0) << 2, 1);   // lane 0 is never a real target
        const bool marked = valid && lane > 0 && got != 0;
        i32 max_f = spi, max_j = -1, n_skip = 0, end_j = i - n_act - 1;
        bool brk = resolve_chunk(s, valid, marked, i - 1, P.max_skip, max_f, max_j, n_skip, end_j);
        i32 st = i - n_act;                       // exact unless the window reaches past the registers
        bool st_known = !(n_act == 64 && i - 64 > lower);
        if (!brk && !st_known) {
            // ---- slow path: the scalar loop runs past the 64 anchors held in registers ----
            const u32 stamp = (u32)i + 1;
            if (push && !in_reg) tmark[wp] = stamp;          // marks that fell outside the window
            drain_stores();
            end_j = lower - 1; st = lower; st_known = true;
            for (i32 base = i - 65; base >= lower; base -= 64) {
                const i32 jj = base - lane;
                const bool inb = jj >= lower;
                i32 xj = 0, yj = 0, sj = 0, fj = 0, pj = -1;
                if (inb) {
                    const u64 k = gk[jj], v = gv[jj], r = ld_u64_l2(grec + jj);
                    xj = (i32)(k & rmask); yj = (i32)(u32)v; sj = (i32)((v >> 32) & 0xff); fj = grec_f(r); pj = grec_p(r);
                }
                const bool reach = inb && xj + P.max_dist_x >= xi;
                const i32 n_reach = (i32)__popcll(__ballot(reach));
                if (n_reach == 0) { st = base + 1; end_j = base; break; }
                const i32 sc2 = comput_sc_dev(xi, yi, xj, yj, sj, P);
                const bool valid2 = reach && sc2 != SC_NONE;
                const i32 s2 = valid2 ? sc2 + fj : SC_NONE;
                if (valid2 && pj >= 0) tmark[pj] = stamp;
                drain_stores();
                const bool marked2 = valid2 && ld_u32_l2(tmark + jj) == stamp;
                brk = resolve_chunk(s2, valid2, marked2, base, P.max_skip, max_f, max_j, n_skip, end_j);
                if (brk) { st_known = false; break; }
                if (n_reach < 64) { st = base - n_reach + 1; end_j = st - 1; break; }
            }
        }
        // ---- max_ii bookkeeping (the "best f in the window" shortcut) ----
        if (mi < 0 || xi - mi_x > P.max_dist_x) {
            u64 best = act ? ((u64)(u32)wf << 32 | (u32)j) : 0;   // f > 0 always; ties keep the larger j
            if (n_act == 64 && i - 64 > lower) {
                if (!st_known) {   // window start behind the registers and not yet determined
                    st = lower;
                    for (i32 base = i - 65; base >= lower; base -= 64) {
                        const i32 jj = base - lane;
                        const bool reach = jj >= lower && (i32)(gk[jj >= 0 ? jj : 0] & rmask) + P.max_dist_x >= xi;
                        const i32 n_reach = (i32)__popcll(__ballot(reach));
                        if (n_reach < 64) { st = base - n_reach + 1; break; }
                    }
                }
                drain_stores();
                for (i32 jj = i - 65 - lane; jj >= st; jj -= 64) {
                    const u64 key = (u64)(u32)grec_f(ld_u64_l2(grec + jj)) << 32 | (u32)jj;
                    best = key > best ? key : best;
                }
            }
            best = wave_max_u64(best);
            const u32 bhi = RFL((u32)(best >> 32)), blo = RFL((u32)best);
            if (bhi == 0) mi = -1;
            else {
                mi = (i32)blo; mi_f = (i32)bhi;
                const i32 d = i - 1 - mi;      // where the anchor lives
                if (d < 64) { mi_x = __builtin_amdgcn_readlane(wx, d); mi_y = __builtin_amdgcn_readlane(wy, d); mi_sp = __builtin_amdgcn_readlane(ws, d); }
                else { const u64 k = gk[mi], v = gv[mi]; mi_x = RFL((i32)(k & rmask)); mi_y = RFL((i32)(u32)v); mi_sp = RFL((i32)((v >> 32) & 0xff)); }
            }
        }
        if (mi >= 0 && mi < end_j) {
            const i32 tmp = RFL(comput_sc_dev(xi, yi, mi_x, mi_y, mi_sp, P));
            if (tmp != SC_NONE && max_f < tmp + mi_f) { max_f = tmp + mi_f; max_j = mi; }
        }
        if (lane == 0) grec[i] = grec_make(max_f, max_j);
        if (mi < 0 || (xi - mi_x <= P.max_dist_x && mi_f < max_f)) { mi = i; mi_x = xi; mi_y = yi; mi_f = max_f; mi_sp = spi; }
        // shift the window by one lane and insert anchor i at lane 0
        wx = __builtin_amdgcn_update_dpp(xi, wx, DPP_WAVE_SHR1, 0xf, 0xf, false);
        wy = __builtin_amdgcn_update_dpp(yi, wy, DPP_WAVE_SHR1, 0xf, 0xf, false);
        wf = __builtin_amdgcn_update_dpp(max_f, wf, DPP_WAVE_SHR1, 0xf, 0xf, false);
        wp = __builtin_amdgcn_update_dpp(max_j, wp, DPP_WAVE_SHR1, 0xf, 0xf, false);
        ws = __builtin_amdgcn_update_dpp(spi, ws, DPP_WAVE_SHR1, 0xf, 0xf, false);
    }
    drain_stores();

    // ---------------- backtrack (mg_chain_backtrack; commentary in k_chain.h) ----------------
    // state lives in the top byte of grec: low 2 bits = t[] (0 free, 1 claimed, 2 tentative), bit 2 =
    // "end visited while unclaimed".  Byte stores hit grec's byte 7.
    u8 *gstate = (u8 *)grec;
    u32 flags = 0;
    const i32 qlen = (i32)P.q_len[qid], tlen = (i32)P.t_len[rid];
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
        i32 i = top, end_i = -1, max_i = top, max_s = 0;
        u64 r = ld_u64_l2(grec + i);
        for (;;) {   // mg_chain_bk_end
            if (lane == 0) gstate[(u64)i * 8 + 7] = (u8)((grec_state(r) & 4) | 2);
            i = RFL(grec_p(r));
            end_i = i;
            i32 s;
            if (i < 0) s = zx;
            else { drain_stores(); r = ld_u64_l2(grec + i); s = zx - RFL(grec_f(r)); }
            if (s > max_s) { max_s = s; max_i = i; }
            else if (max_s - s > P.max_drop) break;
            if (i < 0 || (RFL(grec_state(r)) & 3) != 0) break;
        }
        drain_stores();
        for (i = top; i >= 0 && i != end_i;) {
            const u64 rr = ld_u64_l2(grec + i);
            if (lane == 0) gstate[(u64)i * 8 + 7] = (u8)(grec_state(rr) & 4);
            i = RFL(grec_p(rr));
        }
        drain_stores();
        i32 cnt = 0, first = top, mlen = 0, blen = 0;
        for (i = top; i != max_i;) {
            if (lane == 0) gstate[(u64)i * 8 + 7] = 1;
            ++cnt; first = i;
            const i32 pi = RFL(grec_p(ld_u64_l2(grec + i)));
            if (pi != max_i) {
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
        const i32 sc = i < 0 ? zx : zx - RFL(grec_f(ld_u64_l2(grec + i)));
        if (sc >= P.min_sc && cnt > 0 && cnt >= P.min_cnt) {
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
                    c.query = qid; c.target = rid; c.rev = (i32)rev; c.score = sc; c.cnt = cnt;
                    c.qs = qs; c.qe = qe; c.rs = rs; c.re = re; c.mlen = mlen; c.blen = blen; c.reserved = 0;
                    out.chains[slot] = c;
                }
            }
            if (!P.want_all && (flags & 2u)) break;
        }
    }
    if (lane == 0) out.flags[g] = flags;
}
