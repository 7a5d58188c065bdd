// k_chain_common.h -- pieces shared by the two chain kernels (k_chain_hw.h, k_chain_lpg.h): the per-anchor DP record
// (f | p | backtrack state), loads of data the wave stored itself, and the exact out-of-line continuations of the
// predecessor loop and of the max_ii rescan beyond the anchors a kernel keeps in registers (wave-wide, 64 candidates
// per step, through HBM; `tmark[]` holds the t[] stamps there).
#pragma once
#include "k_chain.h"

#define GREC_NONE 0xFFFFFFu   // "no predecessor" in the 24-bit p field

// per-anchor record written by the DP: f (bits 0..31) | p (bits 32..55) | backtrack state (bits 56..63)
__device__ __forceinline__ u64 grec_make(i32 f, i32 p) { return (u64)(u32)f | (u64)((u32)p & GREC_NONE) << 32; }
__device__ __forceinline__ i32 grec_f(u64 r) { return (i32)(u32)r; }
__device__ __forceinline__ i32 grec_p(u64 r) { u32 p = (u32)(r >> 32) & GREC_NONE; return p == GREC_NONE ? -1 : (i32)p; }
__device__ __forceinline__ u32 grec_state(u64 r) { return (u32)(r >> 56); }

// Loads of data this wave stored earlier (grec, tmark).  A group is private to one wavefront, and a
// CU's L1 is coherent with that CU's own stores once they have drained (vmcnt), so ordinary cached
// loads are correct; the loads only have to stop the compiler from reusing a value across our stores.  Relaxed atomics of WAVEFRONT
// scope say exactly that and compile to plain global loads; `volatile` (rounds 2-5) compiled to FLAT loads with system-scope bits --
// past the L1, and every one of them waiting on both memory counters.
__device__ __forceinline__ u64 ld_u64_l2(const u64 *p) { return __hip_atomic_load(const_cast<u64 *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ u32 ld_u32_l2(const u32 *p) { return __hip_atomic_load(const_cast<u32 *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ void drain_stores() { __builtin_amdgcn_s_waitcnt(0x0070 | 0x0f00); }  // vmcnt(0)

// resolve one 64-candidate chunk of the scalar predecessor loop (used by the out-of-line slow path).
// Inputs are per lane (lane order = visiting order); scalar state is passed by reference.  Returns
// true when the loop broke.
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

struct GroupView {   // what the out-of-line paths need to reach a group's data in HBM
    const u64 *gk, *gv; u64 *grec; u32 *tmark; u64 rmask;
};

// Slow path 1: the scalar predecessor loop of anchor i runs past the 64 anchors held in registers
// (no break inside the window and older anchors still in reach).  Continues exactly, through HBM.
// far_push/far_p: lanes whose mark target fell outside the window.  Returns st (window start) or -1
// when the loop broke before the start was determined.
struct SlowTail { i32 st, max_f, max_j, end_j; };   // returned by value: nothing in the hot loop may be address-taken
__device__ __noinline__ SlowTail chain_slow_tail(GroupView V, ChainParams P, i32 i, i32 xi, i32 yi, i32 lower, bool far_push,
                                                 i32 far_p, i32 max_f, i32 max_j, i32 n_skip, i32 first_base) {
    const i32 lane = (i32)lane_id();
    const u32 stamp = (u32)i + 1;
    i32 end_j = lower - 1, st = lower;
    if (far_push) V.tmark[far_p] = stamp;
    drain_stores();
    for (i32 base = first_base; base >= lower; base -= 64) {
        const i32 jj = base - lane;
        const bool inb = jj >= lower;
        i32 xj = 0, yj = 0, sj = 0, fj = 0, pj = -1;
        if (inb) {
            const u64 k = V.gk[jj], v = V.gv[jj], r = ld_u64_l2(V.grec + jj);
            xj = (i32)(k & V.rmask); yj = (i32)(u32)v; sj = (i32)((v >> 32) & 0xff); fj = grec_f(r); pj = grec_p(r);
        }
        const bool reach = inb && xj + P.max_dist_x >= xi;
        const i32 n_reach = (i32)__popcll(__ballot(reach));
        if (n_reach == 0) { st = base + 1; end_j = base; break; }
        const i32 sc2 = comput_sc_dev(xi, yi, xj, yj, sj, P);
        const bool valid2 = reach && sc2 != SC_NONE;
        const i32 s2 = valid2 ? sc2 + fj : SC_NONE;
        if (valid2 && pj >= 0) V.tmark[pj] = stamp;
        drain_stores();
        const bool marked2 = valid2 && ld_u32_l2(V.tmark + jj) == stamp;
        if (resolve_chunk(s2, valid2, marked2, base, P.max_skip, max_f, max_j, n_skip, end_j)) { st = -1; break; }
        if (n_reach < 64) { st = base - n_reach + 1; end_j = st - 1; break; }
    }
    SlowTail r; r.st = st; r.max_f = max_f; r.max_j = max_j; r.end_j = end_j;
    return r;
}

// Slow path 2: max_ii must be re-derived over a window longer than the registers.  `best` holds the
// per-lane candidates from the register window; returns the wave-wide best key (f << 32 | j).
__device__ __noinline__ u64 chain_slow_rescan(GroupView V, ChainParams P, i32 i, i32 xi, i32 lower, i32 st, u64 best,
                                              i32 first_base) {
    const i32 lane = (i32)lane_id();
    if (st < 0) {   // window start not determined yet
        st = lower;
        for (i32 base = first_base; base >= lower; base -= 64) {
            const i32 jj = base - lane;
            const bool reach = jj >= lower && (i32)(V.gk[jj >= 0 ? jj : 0] & V.rmask) + P.max_dist_x >= xi;
            const i32 n_reach = (i32)__popcll(__ballot(reach));
            if (n_reach < 64) { st = base - n_reach + 1; break; }
        }
    }
    drain_stores();
    for (i32 jj = first_base - lane; jj >= st; jj -= 64) {
        const u64 key = (u64)(u32)grec_f(ld_u64_l2(V.grec + jj)) << 32 | (u32)jj;
        best = key > best ? key : best;
    }
    return best;
}
