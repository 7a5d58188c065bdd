// k_prims.h -- device-wide exclusive scan and stable LSD radix sort (64-bit key, 64-bit value),
// written for gfx950: 64-lane wavefronts, ballot-based digit matching, LDS counters.
#pragma once
#include "internal.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ u64 lanemask_lt() { return (1ULL << lane_id()) - 1ULL; }

// Lanes of the wavefront that hold the same digit as this lane ("match any"; gfx9 has no instruction for it).  Per digit bit:
// the sign-extended bit (v_bfe_i32), its ballot (one v_cmp), and acc |= ballot ^ bit on each half of the mask (one v_bitop3
// each) -- 4 instructions; the plain `m &= bit ? bal : ~bal` compiled to 10.  acc collects the lanes that DIFFER in some bit;
// the class is what is left of the lanes under consideration.  NB: the number of bits is a compile-time or wavefront-uniform
// value.
__device__ __forceinline__ void wave_match_bit(u32 d, int b, u32 &acc_lo, u32 &acc_hi) {
    const u32 nb = (u32)__builtin_amdgcn_sbfe((i32)d, (u32)b, 1u);      // 0 or ~0
    const u64 bal = __ballot(nb != 0);
    acc_lo = __builtin_amdgcn_bitop3_b32(acc_lo, (u32)bal, nb, 0xF6);        // a | (b ^ c)
    acc_hi = __builtin_amdgcn_bitop3_b32(acc_hi, (u32)(bal >> 32), nb, 0xF6);
}
// before = same-digit lanes below this one, total = same-digit lanes: the stable rank ingredients of one row of 64 items
__device__ __forceinline__ u32 wave_match_before(u32 m_lo, u32 m_hi) { return __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u)); }
__device__ __forceinline__ u32 wave_match_total(u32 m_lo, u32 m_hi) { return (u32)__popc(m_lo) + (u32)__popc(m_hi); }

__device__ __forceinline__ u32 wave_incl_scan_u32(u32 v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 o = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v += o;
    }
    return v;
}

// DPP wave scans (gfx9 family: row_shr within 16-lane rows, then row_bcast:15 / row_bcast:31 to
// stitch the four rows).  These stay in the VALU pipeline; __shfl_up would go through ds_bpermute.
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_BCAST15 0x142
#define DPP_BCAST31 0x143
#define DPP_WAVE_SHR1 0x138

// inclusive max-scan over the 64 lanes
__device__ __forceinline__ i32 wave_incl_max_i32(i32 v, i32 identity) {
    i32 o;
    o = __builtin_amdgcn_update_dpp(identity, v, DPP_ROW_SHR(1), 0xf, 0xf, false); v = v > o ? v : o;
    o = __builtin_amdgcn_update_dpp(identity, v, DPP_ROW_SHR(2), 0xf, 0xf, false); v = v > o ? v : o;
    o = __builtin_amdgcn_update_dpp(identity, v, DPP_ROW_SHR(4), 0xf, 0xf, false); v = v > o ? v : o;
    o = __builtin_amdgcn_update_dpp(identity, v, DPP_ROW_SHR(8), 0xf, 0xf, false); v = v > o ? v : o;
    o = __builtin_amdgcn_update_dpp(identity, v, DPP_BCAST15, 0xa, 0xf, false); v = v > o ? v : o;
    o = __builtin_amdgcn_update_dpp(identity, v, DPP_BCAST31, 0xc, 0xf, false); v = v > o ? v : o;
    return v;
}
// value of the previous lane (lane 0 gets `identity`)
__device__ __forceinline__ i32 wave_shr1_i32(i32 v, i32 identity) {
    return __builtin_amdgcn_update_dpp(identity, v, DPP_WAVE_SHR1, 0xf, 0xf, false);
}
// inclusive scan of the monoid of maps x -> max(x + a, b) under composition (earlier lanes first)
__device__ __forceinline__ void wave_incl_clampadd(i32 &a, i32 &b, i32 neg_big) {
#define CLAMPADD_STEP(ctrl, rmask)                                                        \
    {                                                                                     \
        i32 ao = __builtin_amdgcn_update_dpp(0, a, ctrl, rmask, 0xf, false);              \
        i32 bo = __builtin_amdgcn_update_dpp(neg_big, b, ctrl, rmask, 0xf, false);        \
        i32 nb = bo + a;                                                                  \
        b = nb > b ? nb : b;                                                              \
        a = ao + a;                                                                       \
    }
    CLAMPADD_STEP(DPP_ROW_SHR(1), 0xf)
    CLAMPADD_STEP(DPP_ROW_SHR(2), 0xf)
    CLAMPADD_STEP(DPP_ROW_SHR(4), 0xf)
    CLAMPADD_STEP(DPP_ROW_SHR(8), 0xf)
    CLAMPADD_STEP(DPP_BCAST15, 0xa)
    CLAMPADD_STEP(DPP_BCAST31, 0xc)
#undef CLAMPADD_STEP
}
// wave-wide maximum of a packed (score << 32 | index) key; every lane receives the result
__device__ __forceinline__ u64 wave_max_u64(u64 v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        u64 o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

// ------------------------------------------------------------------------------------------
// exclusive scan of u32 (n < 2^32, totals < 2^32).  SCAN_TILE items per 256-thread block.
// ------------------------------------------------------------------------------------------
#define SCAN_THREADS 256
#define SCAN_ITEMS 16
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const u32 *in, u64 n, u32 *block_sums) {
    __shared__ u32 wsum[SCAN_THREADS / 64];
    u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
    u32 s = 0;
    if (base + SCAN_ITEMS <= n && ((uintptr_t)in & 15) == 0) {   // 4 x 16-byte loads instead of 16 dword loads
        const uint4 *p = (const uint4 *)(in + base);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS / 4; ++i) { const uint4 q = p[i]; s += q.x + q.y + q.z + q.w; }
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) s += in[base + i];
    }
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 t = 0;
        for (int w = 0; w < SCAN_THREADS / 64; ++w) t += wsum[w];
        block_sums[blockIdx.x] = t;
    }
}

// single block: in-place exclusive scan of up to any n, writes total to *total.  Every thread owns SS_ITEMS consecutive items,
// so the <= 8192 block sums of a 32 M-entry scan take ONE trip through memory and the block's barriers.  256 threads, not
// 1024: this kernel runs ~16 times per step between kernels of the same stream while OTHER streams fill the chip (the
// query sketch beside the index sort, the chain kernels beside each other); a 16-wavefront workgroup then waited for a CU
// with four free slots on every SIMD -- 94 us on average, 1.2 ms at worst, for 10 us of work -- where four wavefronts fit anywhere.
#define SS_THREADS 256
#define SS_ITEMS 32
__global__ __launch_bounds__(SS_THREADS) void k_scan_small(u32 *data, u32 n, u32 *total) {
    __shared__ u32 wsum[SS_THREADS / 64];
    __shared__ u32 carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (u32 base = 0; base < n; base += SS_THREADS * SS_ITEMS) {
        const u32 i0 = base + threadIdx.x * SS_ITEMS;
        u32 v[SS_ITEMS];
        if (i0 + SS_ITEMS <= n) {
            const uint4 *p = (const uint4 *)(data + i0);      // (i0 is a multiple of 32: 16-byte aligned with the block)
#pragma unroll
            for (int t = 0; t < SS_ITEMS / 4; ++t) { const uint4 q = p[t]; v[4 * t] = q.x; v[4 * t + 1] = q.y; v[4 * t + 2] = q.z; v[4 * t + 3] = q.w; }
        } else {
#pragma unroll
            for (int t = 0; t < SS_ITEMS; ++t) v[t] = i0 + t < n ? data[i0 + t] : 0;
        }
        u32 s = 0;
#pragma unroll
        for (int t = 0; t < SS_ITEMS; ++t) s += v[t];
        const u32 inc = wave_incl_scan_u32(s);
        if (lane_id() == 63) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        u32 off = carry_s + inc - s;
        for (u32 w = 0; w < (threadIdx.x >> 6); ++w) off += wsum[w];
        if (i0 + SS_ITEMS <= n) {
            uint4 *p = (uint4 *)(data + i0);
#pragma unroll
            for (int t = 0; t < SS_ITEMS / 4; ++t) {
                uint4 q;
                q.x = off; off += v[4 * t]; q.y = off; off += v[4 * t + 1]; q.z = off; off += v[4 * t + 2]; q.w = off; off += v[4 * t + 3];
                p[t] = q;
            }
        } else {
#pragma unroll
            for (int t = 0; t < SS_ITEMS; ++t) { if (i0 + t < n) data[i0 + t] = off; off += v[t]; }
        }
        __syncthreads();
        if (threadIdx.x == SS_THREADS - 1) carry_s = off;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = carry_s;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const u32 *in, u32 *out, u64 n, const u32 *block_offs,
                                                            u32 *total) {
    __shared__ u32 wsum[SCAN_THREADS / 64];
    u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
    u32 v[SCAN_ITEMS];
    u32 s = 0;
    const bool vec = base + SCAN_ITEMS <= n && (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    if (vec) {
        const uint4 *p = (const uint4 *)(in + base);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS / 4; ++i) { const uint4 q = p[i]; v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w; }
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) s += v[i];
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = base + i < n ? in[base + i] : 0; s += v[i]; }
    }
    u32 inc = wave_incl_scan_u32(s);
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    u32 off = block_offs[blockIdx.x];
    for (u32 w = 0; w < (threadIdx.x >> 6); ++w) off += wsum[w];
    off += inc - s;
    if (vec) {
        uint4 *po = (uint4 *)(out + base);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS / 4; ++i) {
            uint4 q;
            q.x = off; off += v[4 * i]; q.y = off; off += v[4 * i + 1]; q.z = off; off += v[4 * i + 2]; q.w = off; off += v[4 * i + 3];
            po[i] = q;
        }
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) { if (base + i < n) out[base + i] = off; off += v[i]; }
    }
    if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1) *total = off;
}

// out may alias in.  d_total (device u32) receives the grand total (may be null).  `st`: stream (default ctx->stream);
// `keep`: the block-sum scratch stays with `sc` instead of going back to the pool at once -- required when `st` is not
// ctx->stream, because the pool recycles memory in the order of ctx->stream only.
// bs_pre: the caller's own block-sum array (n / SCAN_TILE + 2 words; single level only: n <= 8192 * SCAN_TILE) -- nothing is allocated
static int scan_exclusive_u32(lrge_hip_ctx *ctx, Scratch &sc, const u32 *in, u32 *out, u64 n, u32 *d_total, hipStream_t st = nullptr,
                              bool keep = false, u32 *bs_pre = nullptr) {
    if (!st) st = ctx->stream;
    if (n == 0) {
        if (d_total) HIPCHK(ctx, hipMemsetAsync(d_total, 0, 4, st));
        return LRGE_OK;
    }
    u64 nb = div_up(n, SCAN_TILE);
    if (bs_pre && nb > 8192) return LRGE_ERR_INVALID;
    u32 *bs = bs_pre ? bs_pre : sc.get<u32>(nb + 1);
    if (!bs) return LRGE_ERR_DEVICE;
    hipLaunchKernelGGL(k_scan_reduce, dim3((u32)nb), dim3(SCAN_THREADS), 0, st, in, n, bs);
    KCHK(ctx);
    if (nb <= 8192) {
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(SS_THREADS), 0, st, bs, (u32)nb, (u32 *)nullptr);
        KCHK(ctx);
    } else {
        int rc = scan_exclusive_u32(ctx, sc, bs, bs, nb, nullptr, st, keep);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_scan_apply, dim3((u32)nb), dim3(SCAN_THREADS), 0, st, in, out, n, bs, d_total);
    KCHK(ctx);
    if (!keep && !bs_pre) sc.drop(bs);
    return LRGE_OK;
}

// ------------------------------------------------------------------------------------------
// stable LSD radix sort, 8 bits per pass.  Tile order inside a block is wave-major:
//   item(tile, w, r, lane) = tile*RS_TILE + w*RS_ITEMS*64 + r*64 + lane
// so that per-wave running digit counters give a stable rank.
// ------------------------------------------------------------------------------------------
#ifndef RS_THREADS
#define RS_THREADS 256
#endif
#define RS_WAVES (RS_THREADS / 64)
#ifndef RS_ITEMS
#define RS_ITEMS 16
#endif
#define RS_TILE (RS_THREADS * RS_ITEMS)

// SEG: segmented sort.  The array is a sequence of independent segments (e.g. one per query) that must
// each be sorted in place; every tile lies inside one segment and the histogram is laid out
// [segment][digit][tile of the segment], so that ONE exclusive scan over it yields, per (tile, digit), the
// global destination of that digit's run -- segments never mix and the segment id costs no sort pass.
// hist index of digit d: hbase + d * hstride; seg = segment id; delta = items in front of the segment that are NOT covered by
// tiles of this sort (segments sorted elsewhere): the scanned histogram only counts tiled items
struct SegTile { u32 start, len, hbase, hstride, seg, delta; };

// Packed anchors (count-only runs): one u64 = [self 1 | span 8 | qpos bits_qy | sort bits sb], sorted KEYS-ONLY on the
// low sb bits; the last scatter pass unpacks every record into the (key, value) pair the chain kernels read
// (key = segment << sh_q | sort bits, value = self << 43 | span << 32 | qpos) -- see k_seed.h for the layouts.
struct UnpackParams { u32 sb, bits_qy, sh_q, dmask; };   // dmask: digit mask of the pass (the last digit may be narrower than 8 bits)
#define RS_MODE_PAIRS 0
#define RS_MODE_KEYS 1
#define RS_MODE_UNPACK 2
#define RS_MODE_PACK 3      // (hash, y) pairs in, ONE packed u64 out: (hash >> 8) << up.sb | rid << up.bits_qy | (pos << 1 | strand); `shift` addresses the packed value

// Blocks are observed to be dealt round-robin over the 8 XCDs (block b -> XCD b % 8), each with its own L2.  Tile t and
// tile t + 1 of a pass write adjacent runs in every digit's region, so they should meet in ONE L2: XCD x takes the
// x-th contiguous eighth of the tiles.  A bijection of [0, nb); placement is a speed matter only.
__device__ __forceinline__ u32 xcd_tile(u32 b, u32 nb) {
#ifdef RS_NO_XCD_MAP
    (void)nb; return b;
#else
    const u32 per = nb >> 3, rem = nb & 7, x = b & 7;
    return x * per + (x < rem ? x : rem) + (b >> 3);
#endif
}

// A sort whose FIRST pass reads the sketch's per-chunk slots in place of a dense array (k_sketch.h: k_sketch_direct writes chunk c's
// entries to slots[c * cap ...), offs[] = exclusive scan of the per-chunk counts): dense index i lives in chunk c = the last one
// with offs[c] <= i, at slots[c * cap + i - offs[c]].  What that saves is k_sketch_compact: one read and one write of every entry.
struct SlotSrc { const u64 *slots; const u32 *offs; const u32 *tile_chunk; u32 n_chunks, cap, total; };
#define SLOT_LDS 384        // chunk offsets a tile keeps in LDS (4096 entries span ~95 chunks of ~43; more: read from memory)

// tile_chunk[t] = the chunk that holds dense index t * RS_TILE (the last c with offs[c] <= it)
__global__ __launch_bounds__(256) void k_tile_chunks(const u32 *__restrict__ offs, u32 n_chunks, u32 n_tiles, u32 *__restrict__ tile_chunk) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const u64 target = (u64)t * RS_TILE;
    u32 lo = 0, hi = n_chunks - 1;
    while (lo < hi) { const u32 mid = lo + (hi - lo + 1) / 2; if ((u64)offs[mid] <= target) lo = mid; else hi = mid - 1; }
    tile_chunk[t] = lo;
}

// the keys of tile `bid` in the wave-major item layout of the sort kernels (row (w, r) = 64 consecutive dense indices), from the slots.
// The chunk offsets the tile needs sit in LDS behind a sentinel (the total); the chunk of a row's first entry moves on from the row
// before (wave-uniform), the lanes of a row pick theirs among the one to three chunks the row spans.  A tile of more than SLOT_LDS - 2
// chunks (tiny reads: chunks of a few entries) takes the general path: offsets from memory, one search per entry.
__device__ __forceinline__ void rs_load_from_slots(const SlotSrc &S, u32 bid, u64 tile0, u32 n_tile, u32 *s_offs, u64 (&k)[RS_ITEMS], u64 fill) {
    const u32 c_lo = S.tile_chunk[bid];
    const u32 left = S.n_chunks - c_lo;                                  // offsets c_lo .. n_chunks - 1 exist; offs[n_chunks] = total
    const u32 n_l = left + 1 < SLOT_LDS ? left + 1 : SLOT_LDS;           // cached: s_offs[j] = off(c_lo + j), j < n_l
    for (u32 j = threadIdx.x; j < n_l; j += RS_THREADS) s_offs[j] = j < left ? S.offs[c_lo + j] : S.total;
    __syncthreads();
    const u32 w = threadIdx.x >> 6, lane = lane_id();
    const u32 last = (u32)tile0 + n_tile - 1;                            // the tile's last dense index
    const bool cached = s_offs[n_l - 1] > last;                          // the cached offsets reach past the tile
    if (cached) {
        u32 jr = 0;                                                      // (relative) chunk of the row's first entry
        {
            const u32 i0 = (u32)tile0 + w * (RS_ITEMS * 64);
            u32 lo = 0, hi = n_l - 1;                                    // last j with s_offs[j] <= i0 (s_offs[0] <= tile0 <= i0)
            while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if (s_offs[mid] <= i0) lo = mid; else hi = mid - 1; }
            jr = lo;
        }
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) {
            const u32 i0 = (u32)tile0 + w * (RS_ITEMS * 64) + (u32)r * 64, i = i0 + lane;
            k[r] = fill;
            if (i0 > last) continue;                                     // (wave-uniform)
            while (s_offs[jr + 1] <= i0) ++jr;                           // wave-uniform: the sentinel stops it
            u32 jm = jr, jc = jr;
            while (s_offs[jc + 1] < i0 + 64 && jc + 1 < n_l - 1) { ++jc; if (i >= s_offs[jc]) jm = jc; }   // chunks that begin inside the row
            if (i <= last) k[r] = S.slots[(u64)(c_lo + jm) * S.cap + (i - s_offs[jm])];
        }
        return;
    }
    auto off_of = [&](u32 c) -> u32 { return c >= S.n_chunks ? S.total : S.offs[c]; };
    u32 c = c_lo;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const u32 il = w * (RS_ITEMS * 64) + (u32)r * 64 + lane;
        k[r] = fill;
        if (il < n_tile) {
            const u32 i = (u32)tile0 + il;
            u32 lo = c, hi = S.n_chunks - 1;                             // last chunk with off <= i
            while (lo < hi) { const u32 mid = lo + (hi - lo + 1) / 2; if (off_of(mid) <= i) lo = mid; else hi = mid - 1; }
            c = lo;
            k[r] = S.slots[(u64)c * S.cap + (i - off_of(c))];
        }
    }
}

// DB: digit bits (8, or 10: three passes instead of four over a 30-bit hash; the runs a tile writes per digit shrink from 16 to 4
// keys.  MEASURED: one 10-bit pass 0.58 + 0.15 + 1.33 ms against 0.43 + 0.05 + 0.88, the whole sort 5.8 against 5.0 ms -- the
// index build stays with 8)
template <bool SEG, int DB = 8, bool SLOTS = false>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const u64 *__restrict__ keys, u64 n, int shift, u32 nb,
                                                        u32 *__restrict__ hist, const SegTile *__restrict__ tiles, u32 dmask = 255, SlotSrc src = SlotSrc()) {
    constexpr u32 ND = 1u << DB;
    static_assert(!SLOTS || !SEG, "slots feed whole (unsegmented) sorts only");
    __shared__ u32 h[ND];
    __shared__ u32 s_offs[SLOTS ? SLOT_LDS : 1];
    for (u32 i = threadIdx.x; i < ND; i += RS_THREADS) h[i] = 0;
    __syncthreads();
    const u32 bid = xcd_tile(blockIdx.x, gridDim.x);
    const u64 tile0 = SEG ? (u64)tiles[bid].start : (u64)bid * RS_TILE;
    const u32 n_tile = SEG ? tiles[bid].len : (u32)((n - tile0) < (u64)RS_TILE ? (n - tile0) : (u64)RS_TILE);
    // the whole tile in flight before the first count (the 4-deep unrolled load -> atomic loop ran at 2.7 TB/s)
    const u32 l0 = (threadIdx.x >> 6) * (RS_ITEMS * 64) + lane_id();
    u64 kk[RS_ITEMS];
    if (SLOTS) rs_load_from_slots(src, bid, tile0, n_tile, s_offs, kk, 0ULL);
    else {
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) kk[r] = l0 + (u32)r * 64 < n_tile ? keys[tile0 + l0 + (u32)r * 64] : 0;
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r)
        if (l0 + (u32)r * 64 < n_tile) atomicAdd(&h[(u32)(kk[r] >> shift) & dmask], 1u);
    __syncthreads();
    for (u32 d = threadIdx.x; d < ND; d += RS_THREADS) {
        const u64 hi = SEG ? (u64)tiles[bid].hbase + (u64)d * tiles[bid].hstride : (u64)d * nb + bid;
        hist[hi] = h[d];
    }
}

template <bool SEG, int MODE, int DB = 8, bool SLOTS = false>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const u64 *__restrict__ keys_in, const u64 *__restrict__ vals_in,
                                                           u64 *__restrict__ keys_out, u64 *__restrict__ vals_out, u64 n,
                                                           int shift, u32 nb, const u32 *__restrict__ hist_scanned,
                                                           const SegTile *__restrict__ tiles, UnpackParams up, SlotSrc src = SlotSrc()) {
    static_assert(!SLOTS || (!SEG && MODE == RS_MODE_KEYS), "slots feed whole keys-only sorts only");
    __shared__ u32 s_offs[SLOTS ? SLOT_LDS : 1];
    // 1. per-wave stable ranks (ballot digit matching + per-wave LDS counters)
    // 2. block-local destinations: the tile is first reordered through LDS so that each digit's
    //    items are contiguous, then written out as coalesced runs (one run per digit per tile)
    constexpr u32 ND = 1u << DB;         // digits
    constexpr int DPT = ND / RS_THREADS; // digits per thread in the digit scan (consecutive ones)
    static_assert(ND % RS_THREADS == 0, "digit count must be a multiple of the block size");
    __shared__ u32 cnt[RS_WAVES][ND];    // per-wave digit counts -> block-local start of (wave, digit)
    __shared__ u32 gbase[ND];            // global destination of the tile's digit run minus its local start
    __shared__ u32 wtot[RS_WAVES];
    __shared__ u64 stage[RS_TILE];       // 32 KB: keys, then values
    const u32 w = threadIdx.x >> 6, lane = lane_id();
    const u32 bid = xcd_tile(blockIdx.x, gridDim.x);
    const u32 dmask = MODE == RS_MODE_PAIRS ? 255u : up.dmask;
    for (u32 i = threadIdx.x; i < RS_WAVES * ND; i += RS_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const u64 tile0 = SEG ? (u64)tiles[bid].start : (u64)bid * RS_TILE;
    const u32 n_tile = SEG ? tiles[bid].len : (u32)((n - tile0) < (u64)RS_TILE ? (n - tile0) : (u64)RS_TILE);
    const u32 l0 = w * (RS_ITEMS * 64) + lane;           // tile-local index of my first item
    const u64 base = tile0 + l0;
    u64 k[RS_ITEMS], v[RS_ITEMS];
    u32 rank[RS_ITEMS];
    // all loads of the tile are issued up front: the values arrive while the keys are being ranked
    if (SLOTS) rs_load_from_slots(src, bid, tile0, n_tile, s_offs, k, ~0ULL);
    else {
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) k[r] = l0 + (u32)r * 64 < n_tile ? keys_in[base + (u64)r * 64] : ~0ULL;
    }
    if (MODE == RS_MODE_PAIRS || MODE == RS_MODE_PACK) {
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) v[r] = l0 + (u32)r * 64 < n_tile ? vals_in[base + (u64)r * 64] : 0;
    }
    if (MODE == RS_MODE_PACK) {
        // the low hash byte is implied by the segment the tile lies in (index_sort_segpacked): what is left of the hash and
        // the position fit one word, and every later pass moves 8 bytes per entry instead of 16
        const u64 pmask = (1ULL << up.bits_qy) - 1;
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r)
            if (l0 + (u32)r * 64 < n_tile) k[r] = (k[r] >> 8) << up.sb | (v[r] >> 32) << up.bits_qy | (v[r] & pmask);
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const bool valid = l0 + (u32)r * 64 < n_tile;
        const u32 d = (u32)(k[r] >> shift) & dmask;
        const u64 mv = __ballot(valid);
        u32 a_lo = 0, a_hi = 0;
#pragma unroll
        for (int b = 0; b < DB; ++b) wave_match_bit(d, b, a_lo, a_hi);
        const u32 m_lo = (u32)mv & ~a_lo, m_hi = (u32)(mv >> 32) & ~a_hi;
        const u32 before = wave_match_before(m_lo, m_hi);
        // every lane of a digit class reads the class counter (one broadcast LDS read), the lowest lane of the class -- the one
        // with nobody before it -- moves it on: no leader search, no cross-lane shuffle.  Same wavefront, in-order LDS: row
        // r + 1 reads what row r wrote.
        const u32 old = cnt[w][d];
        if (valid && before == 0) cnt[w][d] = old + wave_match_total(m_lo, m_hi);
        rank[r] = old + before;
    }
    __syncthreads();
    {   // thread t: digits [t * DPT, (t + 1) * DPT): totals -> exclusive scan over digits -> local starts per (wave, digit)
        u32 c[DPT][RS_WAVES], tot[DPT], ttot = 0;
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            tot[j] = 0;
#pragma unroll
            for (int ww = 0; ww < RS_WAVES; ++ww) { c[j][ww] = cnt[ww][threadIdx.x * DPT + j]; tot[j] += c[j][ww]; }
            ttot += tot[j];
        }
        u32 inc = wave_incl_scan_u32(ttot);
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        u32 dstart = inc - ttot;
        for (u32 ww = 0; ww < w; ++ww) dstart += wtot[ww];
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            const u32 d = threadIdx.x * DPT + j;
            const u64 hi = SEG ? (u64)tiles[bid].hbase + (u64)d * tiles[bid].hstride : (u64)d * nb + bid;
            gbase[d] = hist_scanned[hi] + (SEG ? tiles[bid].delta : 0u) - dstart;
            u32 run = dstart;
#pragma unroll
            for (int ww = 0; ww < RS_WAVES; ++ww) { cnt[ww][d] = run; run += c[j][ww]; }
            dstart += tot[j];
        }
    }
    __syncthreads();
    u32 lpos[RS_ITEMS];
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        u32 d = (u32)(k[r] >> shift) & dmask;
        lpos[r] = cnt[w][d] + rank[r];
        if (l0 + (u32)r * 64 < n_tile) stage[lpos[r]] = k[r];
    }
    __syncthreads();
    u64 ko[RS_ITEMS];
    if (MODE == RS_MODE_UNPACK) {
        const u64 seg = SEG ? (u64)tiles[bid].seg << up.sh_q : 0;
        const u64 smask = (1ULL << up.sb) - 1, qmask = (1ULL << up.bits_qy) - 1;
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) {
            u32 p = (u32)r * RS_THREADS + threadIdx.x;
            if (p < n_tile) {
                const u64 pk = stage[p];
                const u64 o = (u64)gbase[(u32)(pk >> shift) & dmask] + p;
                keys_out[o] = seg | (pk & smask);
                vals_out[o] = ((pk >> (up.sb + up.bits_qy + 8)) & 1) << 43 | ((pk >> (up.sb + up.bits_qy)) & 0xff) << 32 | ((pk >> up.sb) & qmask);
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        u32 p = (u32)r * RS_THREADS + threadIdx.x;
        if (p < n_tile) { ko[r] = stage[p]; keys_out[gbase[(u32)(ko[r] >> shift) & dmask] + p] = ko[r]; }
    }
    if (MODE == RS_MODE_KEYS || MODE == RS_MODE_PACK) return;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        if (l0 + (u32)r * 64 < n_tile) stage[lpos[r]] = v[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        u32 p = (u32)r * RS_THREADS + threadIdx.x;
        if (p < n_tile) vals_out[gbase[(u32)(ko[r] >> shift) & dmask] + p] = stage[p];
    }
}

// Sorts (keys, vals) by bits [begin_bit, begin_bit + nbits) of the key (rounded up to whole bytes).  Ping-pongs between (k0,v0) and (k1,v1);
// *res_k / *res_v point at the buffers holding the result.
// reverse_digits: the LOWEST byte becomes the most significant digit (result ascending in the byte-reversed key).
// d_tiles / n_tiles: segmented sort (see SegTile); every segment is sorted by the given bits, in place.
static int radix_sort_pairs(lrge_hip_ctx *ctx, Scratch &sc, u64 *k0, u64 *v0, u64 *k1, u64 *v1, u64 n, int begin_bit,
                            int nbits, u64 **res_k, u64 **res_v, bool reverse_digits = false,
                            const SegTile *d_tiles = nullptr, u32 n_tiles = 0, int pass_begin = 0, int pass_end = -1) {
    // pass_begin / pass_end: only the LSD passes [pass_begin, pass_end) of the sort (a stable sort by digit each, so a
    // caller may filter the stream between two of them: k_restrict.h)
    *res_k = k0; *res_v = v0;
    if (n <= 1 || nbits <= 0) return LRGE_OK;
    if (n >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "radix sort limited to < 2^32 items (got %llu)", (unsigned long long)n); return LRGE_ERR_INVALID; }
    u32 nb = d_tiles ? n_tiles : (u32)div_up(n, RS_TILE);
    if (nb == 0) return LRGE_OK;
    ALLOC_OR_FAIL(hist, sc, u32, (u64)256 * nb);
    int passes = (nbits + 7) / 8;
    u64 *ki = k0, *vi = v0, *ko = k1, *vo = v1;
    for (int p = pass_begin; p < (pass_end < 0 ? passes : std::min(pass_end, passes)); ++p) {
        int shift = begin_bit + (reverse_digits ? passes - 1 - p : p) * 8;
        if (d_tiles) hipLaunchKernelGGL(k_rs_hist<true>, dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, n, shift, nb, hist, d_tiles);
        else hipLaunchKernelGGL(k_rs_hist<false>, dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, n, shift, nb, hist, d_tiles);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, hist, hist, (u64)256 * nb, nullptr);
        if (rc) return rc;
        {
            StageTimer ts(ctx, LRGE_T_RS_SCATTER);
            if (d_tiles) hipLaunchKernelGGL((k_rs_scatter<true, RS_MODE_PAIRS>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, vi, ko, vo, n, shift, nb, hist, d_tiles, UnpackParams{0, 0, 0, 255});
            else hipLaunchKernelGGL((k_rs_scatter<false, RS_MODE_PAIRS>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, vi, ko, vo, n, shift, nb, hist, d_tiles, UnpackParams{0, 0, 0, 255});
            KCHK(ctx);
            ts.stop();
            ctx->counters[LRGE_C_RS_SCATTER_LAUNCHES] += 1;
            ctx->counters[LRGE_C_RS_SCATTER_ITEMS] += n;
            ctx->counters[LRGE_C_RS_SCATTER_BYTES] += 32 * n;
        }
        u64 *t = ki; ki = ko; ko = t;
        t = vi; vi = vo; vo = t;
    }
    sc.drop(hist);
    *res_k = ki; *res_v = vi;
    return LRGE_OK;
}

// Segmented keys-only sort of packed anchors on their low `nbits` bits (pk0/pk1 ping-pong); the last pass unpacks
// into (out_k, out_v).  See UnpackParams.
// n_items = items covered by the tiles (for the byte counters).
static int radix_sort_packed_seg(lrge_hip_ctx *ctx, Scratch &sc, u64 *pk0, u64 *pk1, u64 *out_k, u64 *out_v, u64 n, int nbits,
                                 const SegTile *d_tiles, u32 n_tiles, UnpackParams up, u64 n_items) {
    if (n == 0 || n_tiles == 0) return LRGE_OK;
    if (n >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "radix sort limited to < 2^32 items (got %llu)", (unsigned long long)n); return LRGE_ERR_INVALID; }
    const u32 nb = n_tiles;
    ALLOC_OR_FAIL(hist, sc, u32, (u64)256 * nb);
    const int passes = nbits > 0 ? (nbits + 7) / 8 : 1;
    u64 *ki = pk0, *ko = pk1;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * 8;
        up.dmask = nbits - shift >= 8 ? 255u : (1u << (nbits - shift)) - 1u;   // bits above nbits are payload, not key
        hipLaunchKernelGGL(k_rs_hist<true>, dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, n, shift, nb, hist, d_tiles, up.dmask);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, hist, hist, (u64)256 * nb, nullptr);
        if (rc) return rc;
        {
            StageTimer ts(ctx, LRGE_T_RS_SCATTER);
            if (p + 1 < passes)
                hipLaunchKernelGGL((k_rs_scatter<true, RS_MODE_KEYS>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, (const u64 *)nullptr, ko, (u64 *)nullptr, n, shift, nb, hist, d_tiles, up);
            else
                hipLaunchKernelGGL((k_rs_scatter<true, RS_MODE_UNPACK>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, (const u64 *)nullptr, out_k, out_v, n, shift, nb, hist, d_tiles, up);
            KCHK(ctx);
            ts.stop();
            ctx->counters[LRGE_C_RS_SCATTER_LAUNCHES] += 1;
            ctx->counters[LRGE_C_RS_SCATTER_ITEMS] += n_items;
            ctx->counters[LRGE_C_RS_SCATTER_BYTES] += (p + 1 < passes ? 16 : 24) * n_items;
        }
        u64 *t = ki; ki = ko; ko = t;
    }
    sc.drop(hist);
    return LRGE_OK;
}

// Pass 0 of radix_sort_keys (8-bit digits) with the sketch's slots as its input: the keys land in `out` (n entries), ordered by the
// pass's digit; the caller goes on with radix_sort_keys(out, other, ..., pass_begin = 1).
static int radix_sort_keys_first_pass_from_slots(lrge_hip_ctx *ctx, Scratch &sc, const u64 *slots, const u32 *offs, u32 n_chunks, u32 cap, u64 *out, u64 n,
                                                 int begin_bit, int nbits, bool reverse_digits) {
    if (n == 0 || nbits <= 0) return LRGE_OK;
    if (n >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "radix sort limited to < 2^32 items (got %llu)", (unsigned long long)n); return LRGE_ERR_INVALID; }
    const u32 nb = (u32)div_up(n, RS_TILE);
    const int passes = (nbits + 7) / 8, d = reverse_digits ? passes - 1 : 0, shift = begin_bit + d * 8;
    UnpackParams up{0, 0, 0, nbits - d * 8 >= 8 ? 255u : (1u << (nbits - d * 8)) - 1u};
    ALLOC_OR_FAIL(hist, sc, u32, (u64)256 * nb);
    ALLOC_OR_FAIL(tile_chunk, sc, u32, (size_t)nb + 1);
    hipLaunchKernelGGL(k_tile_chunks, dim3((u32)div_up(nb, 256)), dim3(256), 0, ctx->stream, offs, n_chunks, nb, tile_chunk);
    KCHK(ctx);
    SlotSrc src{slots, offs, tile_chunk, n_chunks, cap, (u32)n};
    hipLaunchKernelGGL((k_rs_hist<false, 8, true>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, (const u64 *)nullptr, n, shift, nb, hist, (const SegTile *)nullptr, up.dmask, src);
    KCHK(ctx);
    int rc = scan_exclusive_u32(ctx, sc, hist, hist, (u64)256 * nb, nullptr);
    if (rc) return rc;
    {
        StageTimer ts(ctx, LRGE_T_RS_SCATTER);
        hipLaunchKernelGGL((k_rs_scatter<false, RS_MODE_KEYS, 8, true>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, (const u64 *)nullptr, (const u64 *)nullptr, out, (u64 *)nullptr, n,
                           shift, nb, hist, (const SegTile *)nullptr, up, src);
        KCHK(ctx);
        ts.stop();
        ctx->counters[LRGE_C_RS_SCATTER_LAUNCHES] += 1;
        ctx->counters[LRGE_C_RS_SCATTER_ITEMS] += n;
        ctx->counters[LRGE_C_RS_SCATTER_BYTES] += 16 * n;
    }
    sc.drop(hist); sc.drop(tile_chunk);
    return LRGE_OK;
}

// Stable keys-only LSD sort on bits [begin_bit, begin_bit + nbits) (k0 / k1 ping-pong, *res = buffer holding the result).
// digit_bits: 8, or 10 (whole sorts only: no pass range)
static int radix_sort_keys(lrge_hip_ctx *ctx, Scratch &sc, u64 *k0, u64 *k1, u64 n, int begin_bit, int nbits, u64 **res,
                           bool reverse_digits, int pass_begin = 0, int pass_end = -1, int digit_bits = 8) {
    *res = k0;
    if (n <= 1 || nbits <= 0) return LRGE_OK;
    if (n >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "radix sort limited to < 2^32 items (got %llu)", (unsigned long long)n); return LRGE_ERR_INVALID; }
    if (digit_bits != 8 && digit_bits != 10) return LRGE_ERR_INVALID;
    const int DB = digit_bits;
    const u32 nb = (u32)div_up(n, RS_TILE);
    ALLOC_OR_FAIL(hist, sc, u32, ((u64)1 << DB) * nb);
    const int passes = (nbits + DB - 1) / DB;
    u64 *ki = k0, *ko = k1;
    for (int p = pass_begin; p < (pass_end < 0 ? passes : std::min(pass_end, passes)); ++p) {
        const int d = reverse_digits ? passes - 1 - p : p;
        const int shift = begin_bit + d * DB;
        UnpackParams up{0, 0, 0, nbits - d * DB >= DB ? (1u << DB) - 1u : (1u << (nbits - d * DB)) - 1u};
        if (DB == 8) hipLaunchKernelGGL((k_rs_hist<false, 8>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, n, shift, nb, hist, (const SegTile *)nullptr, up.dmask);
        else hipLaunchKernelGGL((k_rs_hist<false, 10>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, n, shift, nb, hist, (const SegTile *)nullptr, up.dmask);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, hist, hist, ((u64)1 << DB) * nb, nullptr);
        if (rc) return rc;
        {
            StageTimer ts(ctx, LRGE_T_RS_SCATTER);
            if (DB == 8)
                hipLaunchKernelGGL((k_rs_scatter<false, RS_MODE_KEYS, 8>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, (const u64 *)nullptr, ko, (u64 *)nullptr, n, shift, nb,
                                   hist, (const SegTile *)nullptr, up);
            else
                hipLaunchKernelGGL((k_rs_scatter<false, RS_MODE_KEYS, 10>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, (const u64 *)nullptr, ko, (u64 *)nullptr, n, shift, nb,
                                   hist, (const SegTile *)nullptr, up);
            KCHK(ctx);
            ts.stop();
            ctx->counters[LRGE_C_RS_SCATTER_LAUNCHES] += 1;
            ctx->counters[LRGE_C_RS_SCATTER_ITEMS] += n;
            ctx->counters[LRGE_C_RS_SCATTER_BYTES] += 16 * n;
        }
        u64 *t = ki; ki = ko; ko = t;
    }
    sc.drop(hist);
    *res = ki;
    return LRGE_OK;
}

// ------------------------------------------------------------------------------------------
// One-sweep form of the keys-only LSD sort (index entries: radix_sort_keys_onesweep).  The three-kernel pass above reads every
// key twice (k_rs_hist, then k_rs_scatter) to know, per tile and digit, where the tile's run starts.  Here the digit totals of ALL
// passes come from one read of the input (k_rs_hist_all: a digit's total does not depend on the order of the keys), and a tile
// learns what the tiles in front of it hold from their published counts (decoupled look-back, Merrill & Garland; Adinets &
// Merrill's Onesweep): (1 + 2 P) n words of traffic instead of 3 P n.
//   state[tile][digit] = flag << 30 | count:  flag 1 = the tile's own count of the digit, 2 = the count of tiles 0 .. tile.
// One 32-bit word carries flag and value, written and polled with relaxed agent-scope atomics (sc1: served by memory, not by
// the writer's or the reader's own L1 / XCD L2), so no ordering between separate words is needed.  Tiles are handed out by a
// ticket counter: a tile only ever waits for tiles with smaller tickets, which are resident -- no deadlock whatever the
// dispatch order.  A poll that runs into OS_SPIN_LIMIT raises *err and gives up (the caller reports it; nothing hangs).
// MEASURED (tools/micro/sort_bench.hip, 242 M keys, 30-bit hash): exact, and SLOWER than the three-kernel passes -- 6.0-6.3 ms
// against 4.9-5.2.  With the tiles' positions handed to it ready-made the pass runs at the scatter's own 0.88 ms (so the XCD
// grouping of the tickets keeps the write locality: without it 1.22 ms); what it loses is the wait: a tile is ready to write
// ~8 us after it starts and then sits on its LDS and registers for ~10 us more until the counts of the ~100 tiles in front
// have been published and walked (8 tiles per trip to memory), 1.45 ms per pass against 0.43 + 0.05 + 0.88.  NOT used by the
// index build; kept with its bench as the record of the attempt.
// ------------------------------------------------------------------------------------------
#define OS_FLAG_SHIFT 30
#define OS_FLAG_AGG (1u << OS_FLAG_SHIFT)
#define OS_FLAG_INCL (2u << OS_FLAG_SHIFT)
#define OS_VAL_MASK ((1u << OS_FLAG_SHIFT) - 1u)
#define OS_SPIN_LIMIT (1u << 22)
#define OS_MAX_PASSES 8
#define OS_LOOK 8

struct OsPasses { int n; int shift[OS_MAX_PASSES]; u32 dmask[OS_MAX_PASSES]; };

// ghist[p * 256 + d] += keys whose digit of pass p is d
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist_all(const u64 *__restrict__ keys, u64 n, OsPasses P, u32 *__restrict__ ghist) {
    __shared__ u32 h[OS_MAX_PASSES * 256];
    for (u32 i = threadIdx.x; i < (u32)P.n * 256; i += RS_THREADS) h[i] = 0;
    __syncthreads();
    const u64 n_tiles = (n + RS_TILE - 1) / RS_TILE;
    for (u64 t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const u64 tile0 = t * RS_TILE;
        const u32 n_tile = (u32)((n - tile0) < (u64)RS_TILE ? (n - tile0) : (u64)RS_TILE);
        const u32 l0 = (threadIdx.x >> 6) * (RS_ITEMS * 64) + lane_id();
        u64 kk[RS_ITEMS];
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) kk[r] = l0 + (u32)r * 64 < n_tile ? keys[tile0 + l0 + (u32)r * 64] : 0;
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r)
            if (l0 + (u32)r * 64 < n_tile)
                for (int p = 0; p < P.n; ++p) atomicAdd(&h[p * 256 + ((u32)(kk[r] >> P.shift[p]) & P.dmask[p])], 1u);
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < (u32)P.n * 256; i += RS_THREADS) if (h[i]) atomicAdd(&ghist[i], h[i]);
}

// one block of 256 threads: ghist[p][.] -> its exclusive prefix sums, in place
__global__ __launch_bounds__(256) void k_rs_gscan(u32 *__restrict__ ghist, int passes) {
    __shared__ u32 wt[4];
    for (int p = 0; p < passes; ++p) {
        const u32 v = ghist[p * 256 + threadIdx.x];
        const u32 inc = wave_incl_scan_u32(v);
        if (lane_id() == 63) wt[threadIdx.x >> 6] = inc;
        __syncthreads();
        u32 before = 0;
        for (u32 w = 0; w < (threadIdx.x >> 6); ++w) before += wt[w];
        ghist[p * 256 + threadIdx.x] = before + inc - v;
        __syncthreads();
    }
}

typedef unsigned int os_v4u __attribute__((ext_vector_type(4)));
#define OS_AUX_SC1 16       // aux bits of the raw buffer intrinsics: sc1

__global__ __launch_bounds__(RS_THREADS) void k_rs_onesweep(const u64 *__restrict__ keys_in, u64 *__restrict__ keys_out, u64 n, int shift, u32 dmask,
                                                            const u32 *__restrict__ gstart /* [256]: first output index of every digit */,
                                                            u32 *__restrict__ state, u32 nb, u32 lg_group, u32 *__restrict__ ticket, u32 *__restrict__ err) {
    __shared__ u32 cnt[RS_WAVES][256];
    __shared__ u32 gbase[256];
    __shared__ u32 dtot[256];            // the tile's digit counts, then (wave 0) the counts of the tiles in front of it
    __shared__ u32 wtot[RS_WAVES];
    __shared__ u64 stage[RS_TILE];
    __shared__ u32 s_tile;
    const u32 w = threadIdx.x >> 6, lane = lane_id();
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket + (blockIdx.x & 7u), 1u);
    for (u32 i = threadIdx.x; i < RS_WAVES * 256; i += RS_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();
    // ticket -> tile.  Blocks are dealt round-robin over the 8 XCDs (block b -> XCD b % 8: see xcd_tile; a speed matter only), and
    // every residue class of b has its own ticket counter.  Tiles are taken in groups of G = 2^lg_group consecutive ones, group g
    // by class g % 8 in ticket order -- the runs a group writes per digit meet in ONE L2.  A tile then waits for tiles of the
    // other classes at most G tickets ahead of its own; blocks are dispatched in index order, so the classes' counters stay within
    // one of each other, and the launcher keeps 8 G well below the number of blocks the device holds at once: the tiles a
    // resident block waits for are drawn whatever else happens.  Past the last full round of 8 groups: tile = 8 ticket + class.
    const u32 G = 1u << lg_group, n_full = nb & ~(8u * G - 1u);
    const u32 xc = blockIdx.x & 7u, tq = s_tile;
    const u32 bid = tq < (n_full >> 3) ? ((((tq >> lg_group) << 3) | xc) << lg_group) | (tq & (G - 1u)) : n_full + ((tq - (n_full >> 3)) << 3) + xc;
    const u64 tile0 = (u64)bid * RS_TILE;
    const u32 n_tile = (u32)((n - tile0) < (u64)RS_TILE ? (n - tile0) : (u64)RS_TILE);
    const u32 l0 = w * (RS_ITEMS * 64) + lane;
    const u64 base = tile0 + l0;
    u64 k[RS_ITEMS];
    u32 rank[RS_ITEMS];
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) k[r] = l0 + (u32)r * 64 < n_tile ? keys_in[base + (u64)r * 64] : ~0ULL;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {      // per-wave stable ranks: see k_rs_scatter
        const bool valid = l0 + (u32)r * 64 < n_tile;
        const u32 d = (u32)(k[r] >> shift) & dmask;
        const u64 mv = __ballot(valid);
        u32 a_lo = 0, a_hi = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) wave_match_bit(d, b, a_lo, a_hi);
        const u32 m_lo = (u32)mv & ~a_lo, m_hi = (u32)(mv >> 32) & ~a_hi;
        const u32 before = wave_match_before(m_lo, m_hi);
        const u32 old = cnt[w][d];
        if (valid && before == 0) cnt[w][d] = old + wave_match_total(m_lo, m_hi);
        rank[r] = old + before;
    }
    __syncthreads();
    u32 dstart, tot = 0;
    {   // thread d: digit totals -> exclusive scan over digits -> local starts per (wave, digit)
        const u32 d = threadIdx.x;
        u32 c[RS_WAVES];
#pragma unroll
        for (int ww = 0; ww < RS_WAVES; ++ww) { c[ww] = cnt[ww][d]; tot += c[ww]; }
        dtot[d] = tot;
        u32 inc = wave_incl_scan_u32(tot);
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        dstart = inc - tot;
        for (u32 ww = 0; ww < w; ++ww) dstart += wtot[ww];
        u32 run = dstart;
#pragma unroll
        for (int ww = 0; ww < RS_WAVES; ++ww) { cnt[ww][d] = run; run += c[ww]; }
    }
    if (w == 0) {
        // Wave 0 publishes and looks back for the whole tile: lane l owns digits 4 l .. 4 l + 3, ONE 16-byte word of the tile's
        // state row, stored and polled write-through / L2-bypassing (sc1) -- a quarter of the transactions of a word per
        // digit, and a 16-byte sc1 access is never torn, so the four flags of a word always agree.
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(state, 0, (int)(nb * 1024u), 0x00020000);
        const os_v4u mine = *(const os_v4u *)&dtot[4 * lane];
        const u32 flag0 = bid == 0 ? OS_FLAG_INCL : OS_FLAG_AGG;
        __builtin_amdgcn_raw_buffer_store_b128(mine | flag0, rsrc, (int)(bid * 1024u + lane * 16u), 0, OS_AUX_SC1);
        os_v4u excl = {0u, 0u, 0u, 0u};
        if (bid > 0) {
            bool failed = false, closed = false;
            u32 p = bid;                                   // tiles [p, bid) are summed
            u32 spins = 0;
            // OS_LOOK tiles are polled at once (their loads are in flight together: one trip to memory per OS_LOOK tiles -- at
            // the start of a launch a thousand resident tiles wait for their predecessors' counts, and a tile-by-tile walk
            // would cost a microsecond per tile)
            while (p > 0 && !closed && !failed) {
                os_v4u v[OS_LOOK];
#pragma unroll
                for (int j = 0; j < OS_LOOK; ++j) {
                    const u32 q = p > (u32)j ? p - 1 - (u32)j : 0u;      // (in front of tile 0: tile 0 again, never used)
                    v[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(q * 1024u + lane * 16u), 0, OS_AUX_SC1);
                }
                asm volatile("" ::: "memory");             // (every round reads memory again)
                bool stalled = false;
                u32 took = 0;
#pragma unroll
                for (int j = 0; j < OS_LOOK; ++j) {
                    if (closed || stalled || (u32)j >= p) continue;
                    const u32 f = v[j].x >> OS_FLAG_SHIFT;
                    if (f == 0) { stalled = true; continue; }      // not published yet: poll again from this tile
                    excl += v[j] & OS_VAL_MASK;
                    ++took;
                    if (f == 2) closed = true;
                }
                p -= took;
                if (stalled) {
                    if (++spins > OS_SPIN_LIMIT) { failed = true; atomicExch(err, 1u); }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __builtin_amdgcn_raw_buffer_store_b128(((excl + mine) & OS_VAL_MASK) | OS_FLAG_INCL, rsrc, (int)(bid * 1024u + lane * 16u), 0, OS_AUX_SC1);
        }
        *(os_v4u *)&dtot[4 * lane] = excl;
    }
    __syncthreads();
    gbase[threadIdx.x] = gstart[threadIdx.x] + dtot[threadIdx.x] - dstart;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const u32 d = (u32)(k[r] >> shift) & dmask;
        if (l0 + (u32)r * 64 < n_tile) stage[cnt[w][d] + rank[r]] = k[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const u32 p = (u32)r * RS_THREADS + threadIdx.x;
        if (p < n_tile) { const u64 ko = stage[p]; keys_out[gbase[(u32)(ko >> shift) & dmask] + p] = ko; }
    }
}

// radix_sort_keys in the one-sweep form: whole sorts of 2^20 <= n < 2^30 keys (*done = false: not taken, nothing was queued).
// d_err: one word the caller zeroed and reads back at its next synchronisation (non-zero: a look-back gave up, the output is void).
static int radix_sort_keys_onesweep(lrge_hip_ctx *ctx, Scratch &sc, u64 *k0, u64 *k1, u64 n, int begin_bit, int nbits, u64 **res,
                                    bool reverse_digits, u32 *d_err, bool *done) {
    *done = false; *res = k0;
    const int passes = (nbits + 7) / 8;
    if (n < (1ULL << 20) || n >= (1ULL << OS_FLAG_SHIFT) || nbits <= 0 || passes > OS_MAX_PASSES) return LRGE_OK;
    const u32 nb = (u32)div_up(n, RS_TILE);
    OsPasses P; P.n = passes;
    for (int p = 0; p < passes; ++p) {
        const int d = reverse_digits ? passes - 1 - p : p;
        P.shift[p] = begin_bit + d * 8;
        P.dmask[p] = nbits - d * 8 >= 8 ? 255u : (1u << (nbits - d * 8)) - 1u;
    }
    // groups of 2^lg_group tiles per XCD (k_rs_onesweep): 8 G tickets must fit the device at once, with room to spare
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_rs_onesweep, RS_THREADS, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); return LRGE_OK; }
    u32 lg_group = (u32)ctx->opt_u64("ONESWEEP_LG_GROUP", 4);
    while (lg_group > 0 && (8u << lg_group) * 2u > (u32)ctx->n_cu * (u32)per_cu) --lg_group;
    ALLOC_OR_FAIL(ghist, sc, u32, (size_t)passes * 256 + 8 * OS_MAX_PASSES);     // + eight ticket counters per pass
    ALLOC_OR_FAIL(state, sc, u32, (u64)256 * nb);
    u32 *tickets = ghist + (size_t)passes * 256;
    HIPCHK(ctx, hipMemsetAsync(ghist, 0, ((size_t)passes * 256 + 8 * OS_MAX_PASSES) * 4, ctx->stream));
    hipLaunchKernelGGL(k_rs_hist_all, dim3(std::min<u32>(nb, (u32)ctx->n_cu * 8)), dim3(RS_THREADS), 0, ctx->stream, k0, n, P, ghist);
    KCHK(ctx);
    hipLaunchKernelGGL(k_rs_gscan, dim3(1), dim3(256), 0, ctx->stream, ghist, passes);
    KCHK(ctx);
    u64 *ki = k0, *ko = k1;
    for (int p = 0; p < passes; ++p) {
        HIPCHK(ctx, hipMemsetAsync(state, 0, (u64)256 * nb * 4, ctx->stream));
        {
            StageTimer ts(ctx, LRGE_T_RS_SCATTER);
            hipLaunchKernelGGL(k_rs_onesweep, dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, ko, n, P.shift[p], P.dmask[p], ghist + (size_t)p * 256, state,
                               nb, lg_group, tickets + 8 * p, d_err);
            KCHK(ctx);
            ts.stop();
            ctx->counters[LRGE_C_RS_SCATTER_LAUNCHES] += 1;
            ctx->counters[LRGE_C_RS_SCATTER_ITEMS] += n;
            ctx->counters[LRGE_C_RS_SCATTER_BYTES] += 16 * n;
        }
        u64 *t = ki; ki = ko; ko = t;
    }
    sc.drop(ghist); sc.drop(state);
    *res = ki; *done = true;
    return LRGE_OK;
}

// ------------------------------------------------------------------------------------------
// Run heads of a sorted key stream, compacted: starts[r] = index of the first element of run r, where a run is a
// maximal stretch of equal (key >> shift).  Two passes over the keys (count per tile, then fill) and nothing
// else -- no flag / rank arrays of the stream's length.  Every thread owns HC_ITEMS consecutive keys (one full
// 128-byte line), so its heads are consecutive in the output.
// ------------------------------------------------------------------------------------------
#define HC_THREADS 256
#define HC_ITEMS 16
#define HC_TILE (HC_THREADS * HC_ITEMS)

// seg_start (may be null; n_seg + 1 ascending entries): positions that start a run whatever the keys say -- the entries of a
// segment-packed index (index_sort_segpacked) carry their hash without its low byte, so two neighbours on either side of a
// segment boundary may look alike
__device__ __forceinline__ u32 hc_boundary_flags(const u32 *__restrict__ seg_start, u32 n_seg, u64 n, u64 base) {
    u32 lo = 0, hi = n_seg + 1;                      // first boundary >= base
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if ((u64)seg_start[mid] < base) lo = mid + 1; else hi = mid; }
    u32 f = 0;
    for (; lo <= n_seg; ++lo) {
        const u64 b = seg_start[lo];
        if (b >= base + 16 || b >= n) break;
        f |= 1u << (u32)(b - base);
    }
    return f;
}

__device__ __forceinline__ u32 hc_load_flags(const u64 *__restrict__ keys, u64 n, u32 shift, u64 base) {
    // bit t set: element base + t starts a run
    u32 f = 0;
    if (base >= n) return 0;
    u64 prev = base ? keys[base - 1] >> shift : 0;
    const bool first = base == 0;
    if (base + HC_ITEMS <= n) {
        const ulonglong2 *p = (const ulonglong2 *)(keys + base);
#pragma unroll
        for (int t = 0; t < HC_ITEMS / 2; ++t) {
            const ulonglong2 q = p[t];
            const u64 a = q.x >> shift, b = q.y >> shift;
            if (a != prev || (first && t == 0)) f |= 1u << (2 * t);
            if (b != a) f |= 1u << (2 * t + 1);
            prev = b;
        }
    } else {
        for (int t = 0; t < HC_ITEMS && base + t < n; ++t) {
            const u64 a = keys[base + t] >> shift;
            if (a != prev || (first && t == 0)) f |= 1u << t;
            prev = a;
        }
    }
    return f;
}

// flags: the 16 head bits of every thread's line, kept for k_heads_fill (2 bytes instead of re-reading 128 bytes of keys)
__global__ __launch_bounds__(HC_THREADS) void k_heads_count(const u64 *__restrict__ keys, u64 n, u32 shift, u32 *__restrict__ bcount,
                                                            u16 *__restrict__ flags, const u32 *__restrict__ seg_start = nullptr, u32 n_seg = 0) {
    __shared__ u32 ws[HC_THREADS / 64];
    const u64 base = (u64)blockIdx.x * HC_TILE + (u64)threadIdx.x * HC_ITEMS;
    u32 f = hc_load_flags(keys, n, shift, base);
    if (seg_start && base < n) f |= hc_boundary_flags(seg_start, n_seg, n, base);
    flags[(u64)blockIdx.x * HC_THREADS + threadIdx.x] = (u16)f;
    u32 c = (u32)__popc(f);
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if (lane_id() == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { u32 t = 0; for (int w = 0; w < HC_THREADS / 64; ++w) t += ws[w]; bcount[blockIdx.x] = t; }
}

__global__ __launch_bounds__(HC_THREADS) void k_heads_fill(const u16 *__restrict__ flags, const u32 *__restrict__ boff,
                                                           u32 *__restrict__ starts) {
    __shared__ u32 ws[HC_THREADS / 64];
    __shared__ u32 tile[HC_TILE];            // the block's heads, in order: they leave as one coalesced run
    const u64 base = (u64)blockIdx.x * HC_TILE + (u64)threadIdx.x * HC_ITEMS;
    u32 f = flags[(u64)blockIdx.x * HC_THREADS + threadIdx.x];
    const u32 c = (u32)__popc(f);
    const u32 inc = wave_incl_scan_u32(c);
    if (lane_id() == 63) ws[threadIdx.x >> 6] = inc;
    __syncthreads();
    u32 o = inc - c, total = 0;
    for (u32 w = 0; w < HC_THREADS / 64; ++w) { if (w < (threadIdx.x >> 6)) o += ws[w]; total += ws[w]; }
    // (a thread's heads are consecutive, but written one by one per thread every store instruction of the wavefront
    // touched 64 different lines)
    while (f) { const u32 t = (u32)__ffs((int)f) - 1; f &= f - 1; tile[o++] = (u32)(base + t); }
    __syncthreads();
    const u32 g0 = boff[blockIdx.x];
    for (u32 i = threadIdx.x; i < total; i += HC_THREADS) starts[g0 + i] = tile[i];
}

// Same without the host round trip: starts must hold n + 1 entries (upper bound), *d_count (device) receives the
// number of runs.  bc_out: scratch block the caller drops once the stream has passed.
static int compact_heads_async(lrge_hip_ctx *ctx, Scratch &sc, const u64 *keys, u64 n, u32 shift, u32 *starts, u32 *d_count) {
    if (n == 0) { HIPCHK(ctx, hipMemsetAsync(d_count, 0, 4, ctx->stream)); return LRGE_OK; }
    const u32 nb = (u32)div_up(n, HC_TILE);
    ALLOC_OR_FAIL(bc, sc, u32, (size_t)nb + 1);
    ALLOC_OR_FAIL(fl, sc, u16, (size_t)nb * HC_THREADS);
    hipLaunchKernelGGL(k_heads_count, dim3(nb), dim3(HC_THREADS), 0, ctx->stream, keys, n, shift, bc, fl);
    KCHK(ctx);
    int rc = scan_exclusive_u32(ctx, sc, bc, bc, nb, d_count);
    if (rc) return rc;
    hipLaunchKernelGGL(k_heads_fill, dim3(nb), dim3(HC_THREADS), 0, ctx->stream, fl, bc, starts);
    KCHK(ctx);
    sc.drop(bc); sc.drop(fl);   // (recycled in stream order)
    return LRGE_OK;
}

// d_starts receives a pool block of n_heads + 1 entries (the extra one is not written); n < 2^32
static int compact_heads(lrge_hip_ctx *ctx, Scratch &sc, const u64 *keys, u64 n, u32 shift, u32 **d_starts, u32 *n_heads,
                         const u32 *seg_start = nullptr, u32 n_seg = 0) {
    *d_starts = nullptr; *n_heads = 0;
    if (n == 0) return LRGE_OK;
    const u32 nb = (u32)div_up(n, HC_TILE);
    ALLOC_OR_FAIL(bc, sc, u32, (size_t)nb + 1);
    ALLOC_OR_FAIL(d_tot, sc, u32, 1);
    ALLOC_OR_FAIL(fl, sc, u16, (size_t)nb * HC_THREADS);
    hipLaunchKernelGGL(k_heads_count, dim3(nb), dim3(HC_THREADS), 0, ctx->stream, keys, n, shift, bc, fl, seg_start, n_seg);
    KCHK(ctx);
    int rc = scan_exclusive_u32(ctx, sc, bc, bc, nb, d_tot);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(n_heads, d_tot, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    u32 *st = sc.get<u32>((size_t)*n_heads + 1);
    if (!st) return LRGE_ERR_DEVICE;
    hipLaunchKernelGGL(k_heads_fill, dim3(nb), dim3(HC_THREADS), 0, ctx->stream, fl, bc, st);
    KCHK(ctx);
    sc.drop(bc); sc.drop(d_tot); sc.drop(fl);
    *d_starts = st;
    return LRGE_OK;
}

// ------------------------------------------------------------------------------------------
// Segment-local sort of packed anchors: ONE workgroup sorts one whole segment (a query's anchors) inside LDS, all
// radix passes, and writes the unpacked (key, value) pairs -- the data cross HBM once in (8 B) and once out (16 B)
// instead of once per pass plus a histogram read per pass.  Same stable LSD passes, same ranking (ballot digit
// matching + per-wave counters) as k_rs_scatter, so the resulting order is identical to the tiled global sort.
// Segments above the variant's capacity stay on the global segmented sort.
// ------------------------------------------------------------------------------------------
struct SegDesc { u32 start, len, seg, pad; };
// DB = digit bits of a pass.  9-bit digits (the 512- and 1024-thread variants: one digit per thread in the scan step) sort the
// 34 key bits of the headline workload in 4 passes instead of 5; their counters are 16-bit (a count is at most CAP <= 16384)
// so that the LDS footprint, hence the residency, stays that of the 8-bit form.
#ifndef LSORT_DB
#define LSORT_DB 9      // digit bits of the 512- and 1024-thread variants
#endif
#define LSORT_BYTES(THREADS, ITEMS, DB) ((size_t)(THREADS) * (ITEMS) * 8 + (size_t)((THREADS) / 64) * (1 << (DB)) * ((DB) > 8 ? 2 : 4) + 64)

template <int THREADS, int ITEMS, int DB>
__global__ __launch_bounds__(THREADS) void k_seg_sort_local(const u64 *__restrict__ pk_in, u64 *__restrict__ out_k, u64 *__restrict__ out_v,
                                                            const SegDesc *__restrict__ segs, UnpackParams up, int nbits) {
    constexpr int WAVES = THREADS / 64, CAP = THREADS * ITEMS, NDIG = 1 << DB;
    static_assert(THREADS >= NDIG, "one thread per digit in the scan step");
    typedef typename std::conditional<(DB > 8), u16, u32>::type CT;
    extern __shared__ u64 lsort_mem[];
    u64 *stage = lsort_mem;                                   // [CAP]
    u32 *wtot = (u32 *)(lsort_mem + CAP);                     // [NDIG / 64]: totals of the 64-digit groups
    CT *cnt = (CT *)(wtot + 16);                              // [WAVES][NDIG]
    const SegDesc sd = segs[blockIdx.x];
    const u32 n = sd.len;
    const u64 *src = pk_in + sd.start;
    const u32 w = threadIdx.x >> 6, lane = lane_id();
    const u32 l0 = w * (ITEMS * 64) + lane;
    u64 k[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) k[r] = l0 + (u32)r * 64 < n ? src[l0 + (u32)r * 64] : ~0ULL;
    // the key bits are dealt evenly over the passes (34 bits in 8-bit digits: 7,7,7,7,6 instead of 8,8,8,8,2): a digit bit costs
    // one ballot per row whether the pass needs it or not, and any stable LSD split gives the same order
    const int passes = nbits > 0 ? (nbits + DB - 1) / DB : 1;
    const int bpp = nbits > 0 ? (nbits + passes - 1) / passes : 1;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * bpp;
        const int nbp = nbits - shift >= bpp ? bpp : (nbits > shift ? nbits - shift : 1);
        const u32 dmask = (1u << nbp) - 1u;
        for (u32 i = threadIdx.x; i < (u32)WAVES * NDIG; i += THREADS) cnt[i] = 0;
        __syncthreads();
        u32 rank[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const u32 d = (u32)(k[r] >> shift) & dmask;
            u32 a_lo = 0, a_hi = 0;
#pragma unroll
            for (int b = 0; b < DB; ++b)
                if (b < nbp) wave_match_bit(d, b, a_lo, a_hi);   // (wavefront-uniform)
            const u32 m_lo = ~a_lo, m_hi = ~a_hi;
            const u32 before = wave_match_before(m_lo, m_hi);
            const u32 old = cnt[w * NDIG + d];                  // see k_rs_scatter
            if (before == 0) cnt[w * NDIG + d] = (CT)(old + wave_match_total(m_lo, m_hi));
            rank[r] = old + before;
        }
        __syncthreads();
        // digit totals -> exclusive scan over digits -> start of every (wave, digit) run (thread t < NDIG = digit t)
        u32 tot = 0, inc = 0;
        if (threadIdx.x < NDIG) {
            for (int ww = 0; ww < WAVES; ++ww) tot += cnt[ww * NDIG + threadIdx.x];
            inc = wave_incl_scan_u32(tot);
            if (lane == 63) wtot[threadIdx.x >> 6] = inc;
        }
        __syncthreads();
        if (threadIdx.x < NDIG) {
            u32 run = inc - tot;
            for (u32 g = 0; g < (threadIdx.x >> 6); ++g) run += wtot[g];
            for (int ww = 0; ww < WAVES; ++ww) { const u32 c = cnt[ww * NDIG + threadIdx.x]; cnt[ww * NDIG + threadIdx.x] = (CT)run; run += c; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const u32 d = (u32)(k[r] >> shift) & dmask;
            stage[cnt[w * NDIG + d] + rank[r]] = k[r];
        }
        __syncthreads();
        if (p + 1 < passes) {
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) k[r] = stage[l0 + (u32)r * 64];
            __syncthreads();
        }
    }
    // unpack and write, coalesced
    const u64 seg = (u64)sd.seg << up.sh_q;
    const u64 smask = (1ULL << up.sb) - 1, qmask = (1ULL << up.bits_qy) - 1;
    for (u32 pp = threadIdx.x; pp < n; pp += THREADS) {
        const u64 pk = stage[pp];
        out_k[sd.start + pp] = seg | (pk & smask);
        out_v[sd.start + pp] = ((pk >> (up.sb + up.bits_qy + 8)) & 1) << 43 | ((pk >> (up.sb + up.bits_qy)) & 0xff) << 32 | ((pk >> up.sb) & qmask);
    }
}

// ------------------------------------------------------------------------------------------
// Index sort, hybrid form (round 3).  The LSD sort of the index entries moved every 8-byte entry four times (k = 15:
// 30 hash bits in 8-bit digits), each pass reading the keys twice (histogram, scatter): 96 bytes per entry, 7 ms of the
// 32 ms headline step for an ordering SURVEY 8(d) counts as zero algorithmic bytes.  Here the two MOST significant digits
// go first -- one plain global pass, one pass segmented by the first digit's 256 buckets (the machinery of the anchor
// sort: SegTile) -- which leaves 65 536 sub-buckets of a few thousand entries, each contiguous and each small enough for
// a workgroup's LDS: the remaining digits are sorted there in one kernel, 8 bytes in, 8 bytes out (k_seg_sort_keys, the
// keys-only sibling of k_seg_sort_local).  64 bytes per entry instead of 96.  Every pass is a stable counting sort by
// one digit, so the result is the order the LSD passes produce: by (d0, d1, ..., d_last), equal keys in arrival order.
// Sub-buckets above the LDS capacity (repeat-rich data: one hash a hundred thousand times) take segmented global passes.
// ------------------------------------------------------------------------------------------
struct LocalPasses { int n; int shift[4]; int bits[4]; };      // LSD order: pass 0 = least significant of the remaining digits

template <int THREADS, int ITEMS, int DB>
__global__ __launch_bounds__(THREADS) void k_seg_sort_keys(const u64 *__restrict__ in, u64 *__restrict__ out, const SegDesc *__restrict__ segs,
                                                           LocalPasses lp) {
    constexpr int WAVES = THREADS / 64, CAP = THREADS * ITEMS, NDIG = 1 << DB;
    static_assert(THREADS >= NDIG, "one thread per digit in the scan step");
    typedef typename std::conditional<(DB > 8), u16, u32>::type CT;
    extern __shared__ u64 lsort_mem[];
    u64 *stage = lsort_mem;                                   // [CAP]
    u32 *wtot = (u32 *)(lsort_mem + CAP);                     // [NDIG / 64]
    CT *cnt = (CT *)(wtot + 16);                              // [WAVES][NDIG]
    const SegDesc sd = segs[blockIdx.x];
    const u32 n = sd.len;
    const u64 *src = in + sd.start;
    const u32 w = threadIdx.x >> 6, lane = lane_id();
    const u32 l0 = w * (ITEMS * 64) + lane;
    u64 k[ITEMS];
    // rows of 64 items; a row that lies wholly behind the segment's end is skipped in every phase (wavefront-uniform test):
    // a block's work follows the segment's length, not the class capacity.  The padding of the last partial row has every
    // digit all ones, sorts last, and lands behind the n real entries.
    const u32 w_base = w * (ITEMS * 64);
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) k[r] = l0 + (u32)r * 64 < n ? src[l0 + (u32)r * 64] : ~0ULL;
    for (int p = 0; p < lp.n; ++p) {
        const int shift = lp.shift[p], nbp = lp.bits[p];
        const u32 dmask = (1u << nbp) - 1u;
        for (u32 i = threadIdx.x; i < (u32)WAVES * NDIG; i += THREADS) cnt[i] = 0;
        __syncthreads();
        u32 rank[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            if (w_base + (u32)r * 64 >= n) { rank[r] = 0; continue; }
            const u32 d = (u32)(k[r] >> shift) & dmask;
            u32 a_lo = 0, a_hi = 0;
#pragma unroll
            for (int b = 0; b < DB; ++b)
                if (b < nbp) wave_match_bit(d, b, a_lo, a_hi);   // (wavefront-uniform)
            const u32 m_lo = ~a_lo, m_hi = ~a_hi;
            const u32 before = wave_match_before(m_lo, m_hi);
            const u32 old = cnt[w * NDIG + d];                  // see k_rs_scatter
            if (before == 0) cnt[w * NDIG + d] = (CT)(old + wave_match_total(m_lo, m_hi));
            rank[r] = old + before;
        }
        __syncthreads();
        u32 tot = 0, inc = 0;
        if (threadIdx.x < NDIG) {
            for (int ww = 0; ww < WAVES; ++ww) tot += cnt[ww * NDIG + threadIdx.x];
            inc = wave_incl_scan_u32(tot);
            if (lane == 63) wtot[threadIdx.x >> 6] = inc;
        }
        __syncthreads();
        if (threadIdx.x < NDIG) {
            u32 run = inc - tot;
            for (u32 g = 0; g < (threadIdx.x >> 6); ++g) run += wtot[g];
            for (int ww = 0; ww < WAVES; ++ww) { const u32 c = cnt[ww * NDIG + threadIdx.x]; cnt[ww * NDIG + threadIdx.x] = (CT)run; run += c; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            if (w_base + (u32)r * 64 >= n) continue;
            const u32 d = (u32)(k[r] >> shift) & dmask;
            stage[cnt[w * NDIG + d] + rank[r]] = k[r];
        }
        __syncthreads();
        if (p + 1 < lp.n) {
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) if (w_base + (u32)r * 64 < n) k[r] = stage[l0 + (u32)r * 64];
            __syncthreads();
        }
    }
    for (u32 pp = threadIdx.x; pp < n; pp += THREADS) out[sd.start + pp] = stage[pp];
}

// start of every (segment, digit) sub-bucket after a segmented pass: the scanned histogram entry of the segment's first tile
__global__ void k_subbucket_starts(const u32 *__restrict__ hist_scanned, const u32 *__restrict__ seg_tile_base, const u32 *__restrict__ seg_n_tiles,
                                   u32 n_segs, u32 *__restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;          // seg * 256 + digit
    if (i >= n_segs * 256) return;
    const u32 s = i >> 8, d = i & 255;
    // (an empty segment has no tile: its sub-buckets start where the next non-empty segment does; the host fills those in)
    out[i] = seg_n_tiles[s] ? hist_scanned[256u * seg_tile_base[s] + d * seg_n_tiles[s]] : 0xFFFFFFFFu;
}
__global__ void k_gather_strided_u32(const u32 *__restrict__ src, u64 stride, u32 n, u32 *__restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[(u64)i * stride];
}
__global__ void k_copy_segments(const u64 *__restrict__ in, u64 *__restrict__ out, const SegDesc *__restrict__ segs) {
    const SegDesc sd = segs[blockIdx.x];
    for (u32 i = threadIdx.x; i < sd.len; i += blockDim.x) out[sd.start + i] = in[sd.start + i];
}

// Sorts the packed index entries k0[0, n) by the hash bits [begin_bit, begin_bit + nbits) in BYTE-REVERSED digit order (what
// radix_sort_keys(..., reverse_digits = true) produces).  k1: a second buffer of n + 1 entries; *res = buffer holding the result.
// *done = false: the input does not suit the hybrid form (too few entries per sub-bucket to be worth it, or too many for LDS
// on average) and nothing has been touched -- the caller runs the LSD passes.
static int index_sort_hybrid(lrge_hip_ctx *ctx, Scratch &sc, u64 *k0, u64 *k1, u64 n, int begin_bit, int nbits, u64 **res, bool *done) {
    *done = false; *res = k0;
    const int passes = (nbits + 7) / 8;
    if (passes < 3 || passes > 6 || n < ctx->opt_u64("HYBRID_SORT_MIN", 1ULL << 22) || n >= (1ULL << 32) || n / 65536 > 6000 || ctx->opt("NO_HYBRID_SORT")) return LRGE_OK;
    // MEASURED (tools/micro/sort_bench.hip, 242 M packed entries, 30 hash bits): 4.77 ms against the LSD form's 5.00 ms alone, and
    // inside the C4 step 7.87 against 7.61 ms (its two host round trips -- bucket and sub-bucket boundaries -- leave the GPU idle
    // while the query sketch is not there to fill the gap).  The bytes fall from 96 to 64 per entry as planned, but the in-LDS
    // passes are bound by their ranking arithmetic (~45 wave instructions per row of 64 keys and pass), not by memory, and a global
    // pass already runs at ~4 TB/s.  So the form is exact, tested (tests/test_gpu_parity.py::test_hybrid_index_sort_is_exact) and
    // OFF unless option HYBRID_SORT asks for it.
    if (!ctx->opt("HYBRID_SORT") && !ctx->opt("HYBRID_SORT_MIN")) return LRGE_OK;
    if (!ctx->lsort_ok[1] || !ctx->lsort_ok[2]) return LRGE_OK;
    // LDS classes: 256 x 8, 256 x 16 (8-bit digits), 512 x 16, 1024 x 16 (9-bit digits)
    const u32 cap_lim[4] = {(u32)std::min<u64>(2048, ctx->opt_u64("DEBUG_LSORT_CAP0", 2048)), (u32)std::min<u64>(4096, ctx->opt_u64("DEBUG_LSORT_CAP0", 4096)),
                            (u32)std::min<u64>(8192, ctx->opt_u64("DEBUG_LSORT_CAP1", 8192)), (u32)std::min<u64>(16384, ctx->opt_u64("DEBUG_LSORT_CAP2", 16384))};
    // digit d of the reversed order = hash bits [8 d, 8 d + 8) (the last one narrower); d = 0 is the most significant
    auto dshift = [&](int d) { return begin_bit + 8 * d; };
    auto dbits = [&](int d) { return nbits - 8 * d >= 8 ? 8 : nbits - 8 * d; };
    const u32 nb = (u32)div_up(n, RS_TILE);
    ALLOC_OR_FAIL(hist, sc, u32, (u64)256 * (nb + 256 + 1));
    // ---- pass A: the most significant digit, over everything ----
    {
        UnpackParams up{0, 0, 0, (1u << dbits(0)) - 1u};
        hipLaunchKernelGGL(k_rs_hist<false>, dim3(nb), dim3(RS_THREADS), 0, ctx->stream, k0, n, dshift(0), nb, hist, (const SegTile *)nullptr, up.dmask);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, hist, hist, (u64)256 * nb, nullptr); if (rc) return rc;
        StageTimer ts(ctx, LRGE_T_RS_SCATTER);
        hipLaunchKernelGGL((k_rs_scatter<false, RS_MODE_KEYS>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, k0, (const u64 *)nullptr, k1, (u64 *)nullptr, n, dshift(0), nb,
                           hist, (const SegTile *)nullptr, up);
        KCHK(ctx);
        ts.stop();
        ctx->counters[LRGE_C_RS_SCATTER_LAUNCHES] += 1; ctx->counters[LRGE_C_RS_SCATTER_ITEMS] += n; ctx->counters[LRGE_C_RS_SCATTER_BYTES] += 16 * n;
    }
    std::vector<u32> bstart(257);
    {
        ALLOC_OR_FAIL(d_b, sc, u32, 256);
        hipLaunchKernelGGL(k_gather_strided_u32, dim3(1), dim3(256), 0, ctx->stream, hist, (u64)nb, 256u, d_b);
        KCHK(ctx);
        HIPCHK(ctx, ctx->d2h(bstart.data(), d_b, 256 * 4, ctx->stream));
        HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
        bstart[256] = (u32)n;
        sc.drop(d_b);
    }
    // ---- pass B: the second digit inside every bucket of the first (segmented: a tile never straddles two buckets) ----
    std::vector<SegTile> tiles; std::vector<u32> seg_tb(256), seg_nt(256);
    u32 tb = 0;
    for (u32 s = 0; s < 256; ++s) {
        const u32 c = bstart[s + 1] - bstart[s], nt = (u32)div_up((u64)c, RS_TILE);
        seg_tb[s] = tb; seg_nt[s] = nt;
        for (u32 lt = 0; lt < nt; ++lt) tiles.push_back(SegTile{bstart[s] + lt * RS_TILE, std::min<u32>(RS_TILE, c - lt * RS_TILE), 256u * tb + lt, nt, s, 0u});
        tb += nt;
    }
    const u32 n_tiles = (u32)tiles.size();
    ALLOC_OR_FAIL(d_tiles, sc, u32, (size_t)n_tiles * (sizeof(SegTile) / 4) + 4);
    ALLOC_OR_FAIL(d_segmeta, sc, u32, 512);
    ALLOC_OR_FAIL(d_sub, sc, u32, 65536);
    std::vector<u32> segmeta(512);
    for (u32 s = 0; s < 256; ++s) { segmeta[s] = seg_tb[s]; segmeta[256 + s] = seg_nt[s]; }
    HIPCHK(ctx, hipMemcpyAsync(d_tiles, tiles.data(), (size_t)n_tiles * sizeof(SegTile), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_segmeta, segmeta.data(), 512 * 4, hipMemcpyHostToDevice, ctx->stream));
    std::vector<u32> sub(65536);
    {
        UnpackParams up{0, 0, 0, (1u << dbits(1)) - 1u};
        hipLaunchKernelGGL(k_rs_hist<true>, dim3(n_tiles), dim3(RS_THREADS), 0, ctx->stream, k1, n, dshift(1), n_tiles, hist, (const SegTile *)d_tiles, up.dmask);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, hist, hist, (u64)256 * n_tiles, nullptr); if (rc) return rc;
        hipLaunchKernelGGL(k_subbucket_starts, dim3(256), dim3(256), 0, ctx->stream, hist, d_segmeta, d_segmeta + 256, 256u, d_sub);
        KCHK(ctx);
        HIPCHK(ctx, ctx->d2h(sub.data(), d_sub, 65536 * 4, ctx->stream));
        StageTimer ts(ctx, LRGE_T_RS_SCATTER);
        hipLaunchKernelGGL((k_rs_scatter<true, RS_MODE_KEYS>), dim3(n_tiles), dim3(RS_THREADS), 0, ctx->stream, k1, (const u64 *)nullptr, k0, (u64 *)nullptr, n, dshift(1), n_tiles,
                           hist, (const SegTile *)d_tiles, up);
        KCHK(ctx);
        ts.stop();
        ctx->counters[LRGE_C_RS_SCATTER_LAUNCHES] += 1; ctx->counters[LRGE_C_RS_SCATTER_ITEMS] += n; ctx->counters[LRGE_C_RS_SCATTER_BYTES] += 16 * n;
        HIPCHK(ctx, ctx->d2h_sync(ctx->stream));          // (also: `tiles` / `segmeta` have travelled)
    }
    // ---- the remaining digits inside every sub-bucket ----
    // sub[s * 256 + d] = start of sub-bucket (s, d) (0xFFFFFFFF for an empty first-level bucket): sizes by differences
    std::vector<SegDesc> cls[4], big;
    {
        u32 next = (u32)n;
        for (int i = 65535; i >= 0; --i) {
            const u32 s = (u32)i >> 8;
            u32 st = sub[(size_t)i];
            if (st == 0xFFFFFFFFu) st = bstart[s];            // (empty bucket: zero-length sub-buckets)
            const u32 len = next - st;
            next = st;
            if (!len) continue;
            const int c = len <= cap_lim[0] ? 0 : len <= cap_lim[1] ? 1 : len <= cap_lim[2] ? 2 : len <= cap_lim[3] ? 3 : 4;
            (c < 4 ? cls[c] : big).push_back(SegDesc{st, len, 0, 0});
        }
    }
    if (ctx->opt("VERBOSE"))
        fprintf(stderr, "[lrge_hip] hybrid index sort: %llu entries, sub-buckets in LDS classes %zu / %zu / %zu / %zu, %zu on global passes\n",
                (unsigned long long)n, cls[0].size(), cls[1].size(), cls[2].size(), cls[3].size(), big.size());
    LocalPasses lp; lp.n = 0;
    for (int d = passes - 1; d >= 2; --d) { lp.shift[lp.n] = dshift(d); lp.bits[lp.n] = dbits(d); ++lp.n; }    // least significant first
    SegDesc *d_seg[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int c = 0; c < 5; ++c) {
        std::vector<SegDesc> &v = c < 4 ? cls[c] : big;
        if (v.empty()) continue;
        d_seg[c] = (SegDesc *)sc.get<u32>(v.size() * 4);
        if (!d_seg[c]) return LRGE_ERR_DEVICE;
        HIPCHK(ctx, hipMemcpyAsync(d_seg[c], v.data(), v.size() * sizeof(SegDesc), hipMemcpyHostToDevice, ctx->stream));
    }
    // (input k0, output k1)
    if (d_seg[3]) { hipLaunchKernelGGL((k_seg_sort_keys<1024, 16, LSORT_DB>), dim3((u32)cls[3].size()), dim3(1024), LSORT_BYTES(1024, 16, LSORT_DB), ctx->stream, k0, k1, d_seg[3], lp); KCHK(ctx); }
    if (d_seg[2]) { hipLaunchKernelGGL((k_seg_sort_keys<512, 16, LSORT_DB>), dim3((u32)cls[2].size()), dim3(512), LSORT_BYTES(512, 16, LSORT_DB), ctx->stream, k0, k1, d_seg[2], lp); KCHK(ctx); }
    if (d_seg[1]) { hipLaunchKernelGGL((k_seg_sort_keys<256, 16, 8>), dim3((u32)cls[1].size()), dim3(256), LSORT_BYTES(256, 16, 8), ctx->stream, k0, k1, d_seg[1], lp); KCHK(ctx); }
    if (d_seg[0]) { hipLaunchKernelGGL((k_seg_sort_keys<256, 8, 8>), dim3((u32)cls[0].size()), dim3(256), LSORT_BYTES(256, 8, 8), ctx->stream, k0, k1, d_seg[0], lp); KCHK(ctx); }
    std::vector<SegTile> btiles;
    if (d_seg[4]) {
        // sub-buckets above a workgroup's LDS: the remaining digits as segmented global passes (k0 <-> k1), then into k1
        u32 tb2 = 0;
        for (size_t s = 0; s < big.size(); ++s) {
            const u32 nt = (u32)div_up((u64)big[s].len, RS_TILE);
            for (u32 lt = 0; lt < nt; ++lt) btiles.push_back(SegTile{big[s].start + lt * RS_TILE, std::min<u32>(RS_TILE, big[s].len - lt * RS_TILE), 256u * tb2 + lt, nt, (u32)s, 0u});
            tb2 += nt;
        }
        const u32 nbt = (u32)btiles.size();
        ALLOC_OR_FAIL(d_bt, sc, u32, (size_t)nbt * (sizeof(SegTile) / 4) + 4);
        ALLOC_OR_FAIL(bh, sc, u32, (u64)256 * nbt);
        HIPCHK(ctx, hipMemcpyAsync(d_bt, btiles.data(), (size_t)nbt * sizeof(SegTile), hipMemcpyHostToDevice, ctx->stream));
        // the histogram of a segmented pass counts tiled items only, and the tiles of these few sub-buckets are scattered over
        // the stream: every tile's destination is its own sub-bucket's start + what the scan says (delta = start of the
        // sub-bucket minus the tiled items in front of it)
        {
            u32 acc = 0;
            size_t ti = 0;
            for (size_t s = 0; s < big.size(); ++s) {
                const u32 nt = (u32)div_up((u64)big[s].len, RS_TILE);
                for (u32 lt = 0; lt < nt; ++lt) btiles[ti++].delta = big[s].start - acc;
                acc += big[s].len;
            }
            HIPCHK(ctx, hipMemcpyAsync(d_bt, btiles.data(), (size_t)nbt * sizeof(SegTile), hipMemcpyHostToDevice, ctx->stream));
        }
        u64 *ki = k0, *ko = k1;
        for (int p = 0; p < lp.n; ++p) {
            UnpackParams up{0, 0, 0, (1u << lp.bits[p]) - 1u};
            hipLaunchKernelGGL(k_rs_hist<true>, dim3(nbt), dim3(RS_THREADS), 0, ctx->stream, ki, n, lp.shift[p], nbt, bh, (const SegTile *)d_bt, up.dmask);
            KCHK(ctx);
            int rc = scan_exclusive_u32(ctx, sc, bh, bh, (u64)256 * nbt, nullptr); if (rc) return rc;
            hipLaunchKernelGGL((k_rs_scatter<true, RS_MODE_KEYS>), dim3(nbt), dim3(RS_THREADS), 0, ctx->stream, ki, (const u64 *)nullptr, ko, (u64 *)nullptr, n, lp.shift[p], nbt,
                               bh, (const SegTile *)d_bt, up);
            KCHK(ctx);
            u64 *t = ki; ki = ko; ko = t;
        }
        if (ki != k1) { hipLaunchKernelGGL(k_copy_segments, dim3((u32)big.size()), dim3(256), 0, ctx->stream, ki, k1, d_seg[4]); KCHK(ctx); }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));        // (`btiles` is a local)
        sc.drop(d_bt); sc.drop(bh);
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));            // (the descriptor vectors are locals)
    for (int c = 0; c < 5; ++c) if (d_seg[c]) sc.drop((u32 *)d_seg[c]);
    sc.drop(hist); sc.drop((u32 *)d_tiles); sc.drop(d_segmeta); sc.drop(d_sub);
    *res = k1; *done = true;
    return LRGE_OK;
}

// ------------------------------------------------------------------------------------------
// Index sort of the (hash, y) PAIR layout, segment-packed form (round 3).  When 2k + bits(rid) + bits(pos) + 1 exceeds 64 --
// the HiFi preset on any sizeable read set -- the index entries are 16-byte pairs and the LSD sort moves 32 bytes per entry and
// pass: five passes for k = 19, 100 GB of traffic per step at C5/10, the largest kernel family of that regime.  Here the MOST
// significant digit of the (byte-reversed) order -- the low hash byte -- goes first, as one pair pass; inside each of its 256
// segments that byte is implied, and what is left of the hash (2k - 8 bits) and the position (rid, pos, strand) fit ONE word
// whenever 2k - 8 + ybits <= 64.  The first of the remaining LSD passes reads the pairs and writes that word (RS_MODE_PACK), the
// others are keys-only passes segmented by the first digit: 40 + 32 + 24 (passes - 2) bytes per entry instead of 40 passes
// (k = 19: 144 instead of 200), and the resident index is 8 bytes per entry.  Stable passes, most significant digit first then
// LSD inside the segments: the order is the pair sort's.  seg_start[257] (host copy returned, device copy allocated from `sc`
// and handed to the caller) says where every segment begins: run detection and the table build need the low byte back.
// ------------------------------------------------------------------------------------------
static int index_sort_segpacked(lrge_hip_ctx *ctx, Scratch &sc, u64 *kx, u64 *ky, u64 *k1, u64 *v1, u64 n, int nbits, u32 ybits, u32 pos1,
                                u64 **res, u32 **d_seg_start, std::vector<u32> *h_seg_start) {
    const int passes = (nbits + 7) / 8;
    auto dbits = [&](int d) { return nbits - 8 * d >= 8 ? 8 : nbits - 8 * d; };
    const u32 nb = (u32)div_up(n, RS_TILE);
    ALLOC_OR_FAIL(hist, sc, u32, (u64)256 * (nb + 256 + 1));
    // ---- pass A: pairs by the low hash byte ----
    {
        hipLaunchKernelGGL(k_rs_hist<false>, dim3(nb), dim3(RS_THREADS), 0, ctx->stream, kx, n, 0, nb, hist, (const SegTile *)nullptr, 255u);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, hist, hist, (u64)256 * nb, nullptr); if (rc) return rc;
        StageTimer ts(ctx, LRGE_T_RS_SCATTER);
        hipLaunchKernelGGL((k_rs_scatter<false, RS_MODE_PAIRS>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, kx, ky, k1, v1, n, 0, nb, hist, (const SegTile *)nullptr,
                           UnpackParams{0, 0, 0, 255});
        KCHK(ctx);
        ts.stop();
        ctx->counters[LRGE_C_RS_SCATTER_LAUNCHES] += 1; ctx->counters[LRGE_C_RS_SCATTER_ITEMS] += n; ctx->counters[LRGE_C_RS_SCATTER_BYTES] += 32 * n;
    }
    u32 *d_b = sc.get<u32>(257);
    if (!d_b) return LRGE_ERR_DEVICE;
    std::vector<u32> &bstart = *h_seg_start;
    bstart.assign(257, 0);
    hipLaunchKernelGGL(k_gather_strided_u32, dim3(1), dim3(256), 0, ctx->stream, hist, (u64)nb, 256u, d_b);
    KCHK(ctx);
    const u32 n32 = (u32)n;
    HIPCHK(ctx, hipMemcpyAsync(d_b + 256, &n32, 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, ctx->d2h(bstart.data(), d_b, 256 * 4, ctx->stream));
    HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
    bstart[256] = n32;
    // ---- tiles of the segmented passes ----
    std::vector<SegTile> tiles;
    u32 tb = 0;
    for (u32 s = 0; s < 256; ++s) {
        const u32 c = bstart[s + 1] - bstart[s], nt = (u32)div_up((u64)c, RS_TILE);
        for (u32 lt = 0; lt < nt; ++lt) tiles.push_back(SegTile{bstart[s] + lt * RS_TILE, std::min<u32>(RS_TILE, c - lt * RS_TILE), 256u * tb + lt, nt, s, 0u});
        tb += nt;
    }
    const u32 n_tiles = (u32)tiles.size();
    ALLOC_OR_FAIL(d_tiles, sc, u32, (size_t)n_tiles * (sizeof(SegTile) / 4) + 4);
    HIPCHK(ctx, hipMemcpyAsync(d_tiles, tiles.data(), (size_t)n_tiles * sizeof(SegTile), hipMemcpyHostToDevice, ctx->stream));
    // ---- the remaining digits, least significant first: d = passes - 1 (pairs in, packed out), then passes - 2 .. 1 ----
    u64 *pi = kx, *po = ky;          // (the sketch's pair buffers are free once pass A has read them: they carry the packed words)
    for (int d = passes - 1; d >= 1; --d) {
        const bool first = d == passes - 1;
        const u32 dm = (1u << dbits(d)) - 1u;
        const int pshift = (int)ybits + 8 * (d - 1);             // where digit d sits in the packed word
        if (first) hipLaunchKernelGGL(k_rs_hist<true>, dim3(n_tiles), dim3(RS_THREADS), 0, ctx->stream, k1, n, 8 * d, n_tiles, hist, (const SegTile *)d_tiles, dm);
        else hipLaunchKernelGGL(k_rs_hist<true>, dim3(n_tiles), dim3(RS_THREADS), 0, ctx->stream, pi, n, pshift, n_tiles, hist, (const SegTile *)d_tiles, dm);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, hist, hist, (u64)256 * n_tiles, nullptr); if (rc) return rc;
        StageTimer ts(ctx, LRGE_T_RS_SCATTER);
        if (first) hipLaunchKernelGGL((k_rs_scatter<true, RS_MODE_PACK>), dim3(n_tiles), dim3(RS_THREADS), 0, ctx->stream, k1, v1, pi, (u64 *)nullptr, n, pshift, n_tiles, hist,
                                      (const SegTile *)d_tiles, UnpackParams{ybits, pos1, 0, dm});
        else hipLaunchKernelGGL((k_rs_scatter<true, RS_MODE_KEYS>), dim3(n_tiles), dim3(RS_THREADS), 0, ctx->stream, pi, (const u64 *)nullptr, po, (u64 *)nullptr, n, pshift, n_tiles,
                                hist, (const SegTile *)d_tiles, UnpackParams{0, 0, 0, dm});
        KCHK(ctx);
        ts.stop();
        ctx->counters[LRGE_C_RS_SCATTER_LAUNCHES] += 1; ctx->counters[LRGE_C_RS_SCATTER_ITEMS] += n; ctx->counters[LRGE_C_RS_SCATTER_BYTES] += (first ? 24 : 16) * n;
        if (!first) { u64 *t = pi; pi = po; po = t; }
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      // (`tiles` is a local)
    sc.drop(hist); sc.drop((u32 *)d_tiles);
    *res = pi; *d_seg_start = d_b;
    return LRGE_OK;
}
