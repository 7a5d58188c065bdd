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
// src: where the tile's items lie in the input of the FIRST pass when that differs from `start` (use_src: the expansion leaves a
// query's kept anchors at the front of a sparse slot and the first pass gathers them into the dense layout every later pass --
// and every later stage -- works in; OverlapRun::batch)
struct SegTile { u32 start, len, hbase, hstride, seg, delta, src, pad; };

// Packed anchors (count-only runs): one u64 = [self 1 | span 8 | qpos bits_qy | sort bits sb], sorted KEYS-ONLY on the
// low sb bits; the last scatter pass unpacks every record into the (key, value) pair the chain kernels read
// (key = segment << sh_q | sort bits, value = self << 43 | span << 32 | qpos) -- see k_seed.h for the layouts.
// The SIGNIFICANCE STRING of an nbits-bit hash: the index stream is ordered by the byte-reversed hash (k_index.h), i.e. by
//     S = b0 . b1 . b2 ... b(m-1)          (b0 = low byte, most significant; the top byte b(m-1) holds tb = nbits - 8 (m - 1) bits)
// read as one number -- a byte swap with the partial top byte squeezed.  Segment-packed index entries (index_sort_segpacked) keep the
// low bits of S; its top bits are the number of their segment.
__host__ __device__ __forceinline__ u64 hash_to_sig(u64 h, u32 nbits) {
    const u32 m = (nbits + 7) >> 3, tb = nbits - 8 * (m - 1);
    const u64 X = __builtin_bswap64(h) >> (64 - 8 * m);                          // b0 | b1 | ... | b(m-1), whole bytes
    return (X >> 8) << tb | (X & 0xFFu);                                          // (the top byte's upper bits are zero)
}
__host__ __device__ __forceinline__ u64 sig_to_hash(u64 S, u32 nbits) {
    const u32 m = (nbits + 7) >> 3, tb = nbits - 8 * (m - 1);
    const u64 X = (S >> tb) << 8 | (S & ((1ULL << tb) - 1));
    return __builtin_bswap64(X << (64 - 8 * m));
}
struct UnpackParams {
    u32 sb, bits_qy, sh_q, dmask, nbits;
    // nd_out != null (round 6): the scatter also leaves, beside every key it writes, the key's digit of the NEXT pass -- (key >> nd_shift) &
    // nd_mask, one byte -- so that the next pass's histogram reads 1 byte per entry instead of the 8-byte key (k_rs_hist, key32 == 2).
    // The bytes of a (tile, digit) run are consecutive like its keys, and neighbouring tiles meet in one L2 (xcd_tile): whole lines go out.
    u8 *nd_out; u32 nd_shift, nd_mask;
    u32 dig16;      // the DIG member of SEGW entries (RS_MODE_DW*: keys_in, and RS_MODE_DW's keys_out) is a u16 array instead of a u32 one (the wave-dense sketch's, round 6)
};   // dmask: digit mask of the pass (the last digit may be narrower than 8 bits); nbits: hash bits (PACK / PACKQ)
#define RS_MODE_PAIRS 0
#define RS_MODE_KEYS 1
#define RS_MODE_UNPACK 2
#define RS_MODE_PACK 3      // (hash, y) pairs in, ONE packed u64 out: R << up.sb | rid << up.bits_qy | (pos << 1 | strand), R = the low up.nbits - 8 bits of the
                            // hash's significance string (k_index.h: hash_to_sig; the low hash byte is the segment); `shift` addresses the packed value
#define RS_MODE_PACKQ 4     // the same, and the pass's digit -- the top e = up.sh_q bits of the second hash byte, `shift` = 16 - e addresses the HASH -- is
                            // left out of the word as well: R = the low up.nbits - 8 - e bits (the pass makes the digit part of the segment:
                            // index_sort_segpacked with e > 0; e <= 7)

// SEGW entries (round 5; k_sketch.h, sketch_write_chunk PK == 2): the sketch hands the sort a WORD per entry -- [the low nbits - 16 bits of the
// hash's significance string | rid | pos << 1 | strand] -- and, in a u32 array, the string's top 16 bits DIG = b0 << 8 | b1: 12 bytes instead
// of the pair's 16.  keys_in / keys_out are that u32 array (passed as u64 pointers), vals_in / vals_out the words.
#define RS_MODE_DW 5        // pass A: (DIG, word) by b0 = DIG >> 8 (`shift` = 8), both members move
#define RS_MODE_DWQ 6       // pass A2 (e = up.sh_q > 0): by the top e bits of b1 (`shift` = 8 - e, dmask = 2^e - 1); ONE packed u64 out, the word with the
                            // low 8 - e bits of b1 put on top of its hash field: R << up.sb | y, exactly RS_MODE_PACKQ's output
#define RS_MODE_DWP 7       // e = 0: the first LSD pass over R: the word takes all of b1 on top of its hash field (RS_MODE_PACK's output) and is ranked
                            // by its own digit (`shift` addresses the packed value)

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
// the DIG member of a wave-dense SEGW entry (k_sketch.h: k_sketch_wave): 16 bits suffice, but 2-byte loads and stores ... see DESIGN section 9
#ifndef WAVE_DIG_BITS
#define WAVE_DIG_BITS 32
#endif
#if WAVE_DIG_BITS == 16
typedef u16 wdig_t;
#else
typedef u32 wdig_t;
#endif
struct SlotSrc { const u64 *slots; const u32 *offs; const u32 *tile_chunk; u32 n_chunks, cap, total; const wdig_t *dig; const u32 *tile_desc; };      // dig: the 16-bit DIG members beside the words (the wave-dense sketch's slots: RS_MODE_DW); tile_desc: see k_tile_desc
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
// tile_desc[8 t ..]: the slot that holds dense index t * RS_TILE and the offsets of it and the six slots behind it (`total` beyond the last
// slot) -- ONE 32-byte load tells a tile where its entries lie when it spans at most six slots (the wave-dense sketch's slots hold ~2 000-2 800
// entries: a tile spans two or three).  The general path below fetches tile_chunk[t], then the offsets into LDS, then synchronises: three
// dependent trips to memory in front of the tile's own loads, and pass A ran at half its dense speed through it (26 against 13 ms per
// H. sapiens-scale part).
__global__ __launch_bounds__(256) void k_tile_desc(const u32 *__restrict__ offs, u32 n_chunks, u32 total, u32 n_tiles, u32 *__restrict__ desc) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const u64 target = (u64)t * RS_TILE;
    u32 lo = 0, hi = n_chunks - 1;
    while (lo < hi) { const u32 mid = lo + (hi - lo + 1) / 2; if ((u64)offs[mid] <= target) lo = mid; else hi = mid - 1; }
    uint4 a, b;
    auto off_of = [&](u32 c) -> u32 { return c >= n_chunks ? total : offs[c]; };
    a.x = lo; a.y = off_of(lo); a.z = off_of(lo + 1); a.w = off_of(lo + 2);
    b.x = off_of(lo + 3); b.y = off_of(lo + 4); b.z = off_of(lo + 5); b.w = off_of(lo + 6);
    ((uint4 *)desc)[2 * (size_t)t] = a; ((uint4 *)desc)[2 * (size_t)t + 1] = b;
}

// f(r, src): item r of this lane lives at index `src` of the slot arrays (called for the lane's valid items only, r a compile-time constant
// after unrolling)
template <typename F>
__device__ __forceinline__ void rs_for_slot_items(const SlotSrc &S, u32 bid, u64 tile0, u32 n_tile, u32 *s_offs, F &&f) {
    if (S.tile_desc) {
        const uint4 a = ((const uint4 *)S.tile_desc)[2 * (size_t)bid], b = ((const uint4 *)S.tile_desc)[2 * (size_t)bid + 1];
        const u32 last_ = (u32)tile0 + n_tile - 1;
        if (b.w > last_) {                                               // (block-uniform) the tile lies inside slots a.x .. a.x + 5
            const u32 l0_ = (threadIdx.x >> 6) * (RS_ITEMS * 64) + lane_id();
#pragma unroll
            for (int r = 0; r < RS_ITEMS; ++r) {
                const u32 il = l0_ + (u32)r * 64;
                if (il < n_tile) {
                    const u32 i = (u32)tile0 + il;
                    u32 j = 0, o = a.y;
                    if (i >= a.z) { j = 1; o = a.z; }
                    if (i >= a.w) { j = 2; o = a.w; }
                    if (i >= b.x) { j = 3; o = b.x; }
                    if (i >= b.y) { j = 4; o = b.y; }
                    if (i >= b.z) { j = 5; o = b.z; }
                    f(r, (u64)(a.x + j) * S.cap + (i - o));
                }
            }
            return;
        }
    }
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
            if (i0 > last) continue;                                     // (wave-uniform)
            while (s_offs[jr + 1] <= i0) ++jr;                           // wave-uniform: the sentinel stops it
            u32 jm = jr, jc = jr;
            while (s_offs[jc + 1] < i0 + 64 && jc + 1 < n_l - 1) { ++jc; if (i >= s_offs[jc]) jm = jc; }   // chunks that begin inside the row
            if (i <= last) f(r, (u64)(c_lo + jm) * S.cap + (i - s_offs[jm]));
        }
        return;
    }
    auto off_of = [&](u32 c) -> u32 { return c >= S.n_chunks ? S.total : S.offs[c]; };
    u32 c = c_lo;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const u32 il = w * (RS_ITEMS * 64) + (u32)r * 64 + lane;
        if (il < n_tile) {
            const u32 i = (u32)tile0 + il;
            u32 lo = c, hi = S.n_chunks - 1;                             // last chunk with off <= i
            while (lo < hi) { const u32 mid = lo + (hi - lo + 1) / 2; if (off_of(mid) <= i) lo = mid; else hi = mid - 1; }
            c = lo;
            f(r, (u64)c * S.cap + (i - off_of(c)));
        }
    }
}
__device__ __forceinline__ void rs_load_from_slots(const SlotSrc &S, u32 bid, u64 tile0, u32 n_tile, u32 *s_offs, u64 (&k)[RS_ITEMS], u64 fill) {
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) k[r] = fill;
    rs_for_slot_items(S, bid, tile0, n_tile, s_offs, [&](int r, u64 src) { k[r] = S.slots[src]; });
}

// DB: digit bits.  The library launches 8 only; the 10-bit instantiation lives with the measurements that ruled it out
// (tools/micro/sort_forms.h: one 10-bit pass 0.58 + 0.15 + 1.33 ms against 0.43 + 0.05 + 0.88, the whole sort 5.8 against 5.0 ms)
// sig_nbits != 0: the keys are raw hashes of that many bits and the digit is cut from their significance string (k_index.h:
// hash_to_sig) -- the first LSD pass of index_sort_segpacked without an A2 pass reads the pairs' hashes
template <bool SEG, int DB = 8, bool SLOTS = false>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const u64 *__restrict__ keys, u64 n, int shift, u32 nb,
                                                        u32 *__restrict__ hist, const SegTile *__restrict__ tiles, u32 dmask = 255, SlotSrc src = SlotSrc(),
                                                        u32 use_src = 0, u32 sig_nbits = 0, u32 key32 = 0) {       // key32 = 1: `keys` is a u32 array (the DIG member of SEGW entries); 3: a u16 array (the same, wave-dense form); 2: a u8 array of ready-made digits
    constexpr u32 ND = 1u << DB;
    static_assert(!SLOTS || !SEG, "slots feed whole (unsegmented) sorts only");
    __shared__ u32 h[ND];
    __shared__ u32 s_offs[SLOTS ? SLOT_LDS : 1];
    for (u32 i = threadIdx.x; i < ND; i += RS_THREADS) h[i] = 0;
    __syncthreads();
    const u32 bid = xcd_tile(blockIdx.x, gridDim.x);
    const u64 tile0 = SEG ? (u64)(use_src ? tiles[bid].src : tiles[bid].start) : (u64)bid * RS_TILE;
    const u32 n_tile = SEG ? tiles[bid].len : (u32)((n - tile0) < (u64)RS_TILE ? (n - tile0) : (u64)RS_TILE);
    // the whole tile in flight before the first count (the 4-deep unrolled load -> atomic loop ran at 2.7 TB/s)
    if (!SLOTS && key32 == 2) {
        // `keys` is the digit-byte array the pass before left (UnpackParams::nd_out): 16 bytes per lane and load, whatever the tile's alignment
        // (the array has 64 spare bytes behind its last entry; bytes outside [tile0, tile0 + n_tile) are skipped)
        const u8 *d8 = (const u8 *)keys;
        const u64 a0 = tile0 & ~15ULL, end = tile0 + n_tile;
        for (u64 g = a0 + 16ULL * threadIdx.x; g < end; g += 16ULL * RS_THREADS) {
            const uint4 q = *(const uint4 *)(d8 + g);
            const u32 v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const u64 i = g + 4 * j + t;
                    if (i >= tile0 && i < end) atomicAdd(&h[(v[j] >> (8 * t)) & 0xffu], 1u);
                }
            }
        }
        __syncthreads();
        for (u32 d = threadIdx.x; d < ND; d += RS_THREADS) {
            const u64 hi = SEG ? (u64)tiles[bid].hbase + (u64)d * tiles[bid].hstride : (u64)d * nb + bid;
            hist[hi] = h[d];
        }
        return;
    }
    const u32 l0 = (threadIdx.x >> 6) * (RS_ITEMS * 64) + lane_id();
    u64 kk[RS_ITEMS];
    if (SLOTS && src.dig) {
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) kk[r] = 0;
        rs_for_slot_items(src, bid, tile0, n_tile, s_offs, [&](int r, u64 si) { kk[r] = (u64)src.dig[si]; });
    } else if (SLOTS) rs_load_from_slots(src, bid, tile0, n_tile, s_offs, kk, 0ULL);
    else {
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r)
            kk[r] = l0 + (u32)r * 64 < n_tile ? (key32 == 3 ? (u64)((const u16 *)keys)[tile0 + l0 + (u32)r * 64] : key32 ? (u64)((const u32 *)keys)[tile0 + l0 + (u32)r * 64] : keys[tile0 + l0 + (u32)r * 64]) : 0;
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r)
        if (l0 + (u32)r * 64 < n_tile) atomicAdd(&h[(u32)((sig_nbits ? hash_to_sig(kk[r], sig_nbits) : kk[r]) >> shift) & dmask], 1u);
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
                                                           const SegTile *__restrict__ tiles, UnpackParams up, SlotSrc src = SlotSrc(), u32 use_src = 0) {
    static_assert(!SLOTS || (!SEG && (MODE == RS_MODE_KEYS || MODE == RS_MODE_DW)), "slots feed the first pass of whole sorts only: packed keys, or SEGW (DIG, word) entries");
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
    // PACKQ: the digit is not in the staged word; it is staged beside it, one byte per item, in the rank counters' LDS (free by then)
    static_assert(sizeof(u32) * RS_WAVES * ND >= RS_TILE || (MODE != RS_MODE_PACKQ && MODE != RS_MODE_DWQ), "the digit bytes alias the rank counters");
    constexpr bool DWIN = MODE == RS_MODE_DW || MODE == RS_MODE_DWQ || MODE == RS_MODE_DWP;      // the key member comes from a u32 array
    u8 *sdig = (u8 *)&cnt[0][0];
    const u32 w = threadIdx.x >> 6, lane = lane_id();
    const u32 bid = xcd_tile(blockIdx.x, gridDim.x);
    const u32 dmask = MODE == RS_MODE_PAIRS ? 255u : up.dmask;
    for (u32 i = threadIdx.x; i < RS_WAVES * ND; i += RS_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const u64 tile0 = SEG ? (u64)(use_src ? tiles[bid].src : tiles[bid].start) : (u64)bid * RS_TILE;
    const u32 n_tile = SEG ? tiles[bid].len : (u32)((n - tile0) < (u64)RS_TILE ? (n - tile0) : (u64)RS_TILE);
    const u32 l0 = w * (RS_ITEMS * 64) + lane;           // tile-local index of my first item
    const u64 base = tile0 + l0;
    u64 k[RS_ITEMS], v[RS_ITEMS];
    u32 rank[RS_ITEMS];
    // all loads of the tile are issued up front: the values arrive while the keys are being ranked
    if (SLOTS && MODE == RS_MODE_DW) {       // the wave-dense sketch's slots: 16-bit DIG members and the words, the same index in both arrays
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) { k[r] = ~0ULL; v[r] = 0; }
        rs_for_slot_items(src, bid, tile0, n_tile, s_offs, [&](int r, u64 si) { k[r] = (u64)src.dig[si]; v[r] = src.slots[si]; });
    } else if (SLOTS) rs_load_from_slots(src, bid, tile0, n_tile, s_offs, k, ~0ULL);
    else {
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r)
            k[r] = l0 + (u32)r * 64 < n_tile ? (DWIN ? (up.dig16 ? (u64)((const u16 *)keys_in)[base + (u64)r * 64] : (u64)((const u32 *)keys_in)[base + (u64)r * 64]) : keys_in[base + (u64)r * 64]) : ~0ULL;
    }
    if ((MODE == RS_MODE_PAIRS || MODE == RS_MODE_PACK || MODE == RS_MODE_PACKQ || DWIN) && !(SLOTS && MODE == RS_MODE_DW)) {
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) v[r] = l0 + (u32)r * 64 < n_tile ? vals_in[base + (u64)r * 64] : 0;
    }
    if (MODE == RS_MODE_PACK) {
        // the low hash byte is implied by the segment the tile lies in (index_sort_segpacked): what is left of the hash and
        // the position fit one word, and every later pass moves 8 bytes per entry instead of 16
        const u64 pmask = (1ULL << up.bits_qy) - 1, rmask = (1ULL << (up.nbits - 8)) - 1;
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r)
            if (l0 + (u32)r * 64 < n_tile) k[r] = (hash_to_sig(k[r], up.nbits) & rmask) << up.sb | (v[r] >> 32) << up.bits_qy | (v[r] & pmask);
    }
    if (MODE == RS_MODE_DWP) {       // the word takes b1 on top of its hash field and is the key from here on
        const u32 hs = up.sb + up.nbits - 16;
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r)
            if (l0 + (u32)r * 64 < n_tile) k[r] = v[r] | (k[r] & 0xffULL) << hs;
    }
    // (PACKQ / DWQ: the digit comes from the hash itself, so the word is packed only when it is staged)
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
        if (MODE != RS_MODE_PACKQ && MODE != RS_MODE_DWQ && l0 + (u32)r * 64 < n_tile) stage[lpos[r]] = k[r];
    }
    if (MODE == RS_MODE_DWQ) {
        __syncthreads();                               // every local position is known: the counters' LDS takes the digit bytes
        const u32 hs = up.sb + up.nbits - 16;
        const u64 lowb = (1ULL << (8 - up.sh_q)) - 1;
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) {
            if (l0 + (u32)r * 64 < n_tile) {
                sdig[lpos[r]] = (u8)((u32)(k[r] >> shift) & dmask);
                stage[lpos[r]] = v[r] | (k[r] & lowb) << hs;
            }
        }
    }
    if (MODE == RS_MODE_PACKQ) {
        __syncthreads();                               // every local position is known: the counters' LDS takes the digit bytes
        const u64 pmask = (1ULL << up.bits_qy) - 1, rmask = (1ULL << (up.nbits - 8 - up.sh_q)) - 1;
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) {
            if (l0 + (u32)r * 64 < n_tile) {
                sdig[lpos[r]] = (u8)((u32)(k[r] >> shift) & dmask);
                stage[lpos[r]] = (hash_to_sig(k[r], up.nbits) & rmask) << up.sb | (v[r] >> 32) << up.bits_qy | (v[r] & pmask);
            }
        }
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
        if (p < n_tile) {
            ko[r] = stage[p];
            u32 d = (u32)(ko[r] >> shift) & dmask;
            if (MODE == RS_MODE_PACKQ || MODE == RS_MODE_DWQ) d = sdig[p];
            if (MODE == RS_MODE_DW) { if (up.dig16) ((u16 *)keys_out)[gbase[d] + p] = (u16)ko[r]; else ((u32 *)keys_out)[gbase[d] + p] = (u32)ko[r]; }
            else keys_out[gbase[d] + p] = ko[r];
            if (MODE != RS_MODE_DW && MODE != RS_MODE_PAIRS && up.nd_out) up.nd_out[gbase[d] + p] = (u8)((u32)(ko[r] >> up.nd_shift) & up.nd_mask);
        }
    }
    if (MODE == RS_MODE_KEYS || MODE == RS_MODE_PACK || MODE == RS_MODE_PACKQ || MODE == RS_MODE_DWQ || MODE == RS_MODE_DWP) return;
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
// src_first: the first pass reads tile t at tiles[t].src of pk0 (a sparse layout), everything after it is dense.
static int radix_sort_packed_seg(lrge_hip_ctx *ctx, Scratch &sc, u64 *pk0, u64 *pk1, u64 *out_k, u64 *out_v, u64 n, int nbits,
                                 const SegTile *d_tiles, u32 n_tiles, UnpackParams up, u64 n_items, bool src_first = false) {
    if (n == 0 || n_tiles == 0) return LRGE_OK;
    if (n >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "radix sort limited to < 2^32 items (got %llu)", (unsigned long long)n); return LRGE_ERR_INVALID; }
    const u32 nb = n_tiles;
    ALLOC_OR_FAIL(hist, sc, u32, (u64)256 * nb);
    const int passes = nbits > 0 ? (nbits + 7) / 8 : 1;
    u64 *ki = pk0, *ko = pk1;
    // next-pass digit bytes (UnpackParams::nd_out; option NO_DIGIT_BYTES: every histogram reads the packed anchors)
    u8 *nd = (ctx->opt("NO_DIGIT_BYTES") || passes < 2) ? nullptr : sc.get<u8>(n + 64);
    if (!nd) { (void)hipGetLastError(); ctx->err.clear(); }
    bool nd_valid = false;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * 8;
        up.dmask = nbits - shift >= 8 ? 255u : (1u << (nbits - shift)) - 1u;   // bits above nbits are payload, not key
        up.nd_out = nullptr;
        if (nd && p + 2 < passes + 1 && p + 1 < passes) { up.nd_out = nd; up.nd_shift = (u32)shift + 8u; up.nd_mask = nbits - (shift + 8) >= 8 ? 255u : (1u << (nbits - (shift + 8))) - 1u; }
        const u32 us = (src_first && p == 0) ? 1u : 0u;
        if (nd_valid) hipLaunchKernelGGL(k_rs_hist<true>, dim3(nb), dim3(RS_THREADS), 0, ctx->stream, (const u64 *)nd, n, 0, nb, hist, d_tiles, up.dmask, SlotSrc(), 0u, 0u, 2u);
        else hipLaunchKernelGGL(k_rs_hist<true>, dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, n, shift, nb, hist, d_tiles, up.dmask, SlotSrc(), us);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, hist, hist, (u64)256 * nb, nullptr);
        if (rc) return rc;
        {
            StageTimer ts(ctx, LRGE_T_RS_SCATTER);
            if (p + 1 < passes)
                hipLaunchKernelGGL((k_rs_scatter<true, RS_MODE_KEYS>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, (const u64 *)nullptr, ko, (u64 *)nullptr, n, shift, nb, hist, d_tiles, up, SlotSrc(), us);
            else
                hipLaunchKernelGGL((k_rs_scatter<true, RS_MODE_UNPACK>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, (const u64 *)nullptr, out_k, out_v, n, shift, nb, hist, d_tiles, up, SlotSrc(), us);
            KCHK(ctx);
            ts.stop();
            ctx->counters[LRGE_C_RS_SCATTER_LAUNCHES] += 1;
            ctx->counters[LRGE_C_RS_SCATTER_ITEMS] += n_items;
            ctx->counters[LRGE_C_RS_SCATTER_BYTES] += (p + 1 < passes ? 16 : 24) * n_items;
        }
        nd_valid = up.nd_out != nullptr;
        u64 *t = ki; ki = ko; ko = t;
    }
    if (nd) sc.drop(nd);
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
    // large slots (the wave-dense sketch's: ~2 000-2 800 entries each): one 32-byte descriptor per tile instead of the offsets-through-LDS path
    u32 *tile_desc = nullptr;
    if (cap >= 1024) {
        tile_desc = sc.get<u32>((size_t)nb * 8 + 8);
        if (!tile_desc) return LRGE_ERR_DEVICE;
        hipLaunchKernelGGL(k_tile_desc, dim3((u32)div_up(nb, 256)), dim3(256), 0, ctx->stream, offs, n_chunks, (u32)n, nb, tile_desc);
        KCHK(ctx);
    }
    SlotSrc src{slots, offs, tile_chunk, n_chunks, cap, (u32)n, nullptr, tile_desc};
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
    sc.drop(hist); sc.drop(tile_chunk); if (tile_desc) sc.drop(tile_desc);
    return LRGE_OK;
}

// Stable keys-only LSD sort on bits [begin_bit, begin_bit + nbits) (k0 / k1 ping-pong, *res = buffer holding the result), 8 bits
// per pass.  (Other forms of this sort -- 10-bit digits, one-sweep, two MSD passes + LDS -- were built and measured in rounds 2-3
// and are kept with their bench under tools/micro/sort_forms.h; none beat these three-kernel passes inside the step.)
static int radix_sort_keys(lrge_hip_ctx *ctx, Scratch &sc, u64 *k0, u64 *k1, u64 n, int begin_bit, int nbits, u64 **res,
                           bool reverse_digits, int pass_begin = 0, int pass_end = -1) {
    *res = k0;
    if (n <= 1 || nbits <= 0) return LRGE_OK;
    if (n >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "radix sort limited to < 2^32 items (got %llu)", (unsigned long long)n); return LRGE_ERR_INVALID; }
    const u32 nb = (u32)div_up(n, RS_TILE);
    ALLOC_OR_FAIL(hist, sc, u32, (u64)256 * nb);
    const int passes = (nbits + 7) / 8;
    u64 *ki = k0, *ko = k1;
    const int p_end = pass_end < 0 ? passes : std::min(pass_end, passes);
    // next-pass digit bytes (UnpackParams::nd_out; option NO_DIGIT_BYTES: every histogram reads the keys)
    u8 *nd = (ctx->opt("NO_DIGIT_BYTES") || p_end - pass_begin < 2) ? nullptr : sc.get<u8>(n + 64);
    if (!nd) { (void)hipGetLastError(); ctx->err.clear(); }
    bool nd_valid = false;
    for (int p = pass_begin; p < p_end; ++p) {
        const int d = reverse_digits ? passes - 1 - p : p;
        const int shift = begin_bit + d * 8;
        UnpackParams up{0, 0, 0, nbits - d * 8 >= 8 ? 255u : (1u << (nbits - d * 8)) - 1u, 0, nullptr, 0, 0};
        if (nd && p + 1 < p_end) {
            const int d2 = reverse_digits ? passes - 2 - p : p + 1;
            up.nd_out = nd; up.nd_shift = (u32)(begin_bit + d2 * 8); up.nd_mask = nbits - d2 * 8 >= 8 ? 255u : (1u << (nbits - d2 * 8)) - 1u;
        }
        if (nd_valid) hipLaunchKernelGGL((k_rs_hist<false, 8>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, (const u64 *)nd, n, 0, nb, hist, (const SegTile *)nullptr, up.dmask, SlotSrc(), 0u, 0u, 2u);
        else hipLaunchKernelGGL((k_rs_hist<false, 8>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, n, shift, nb, hist, (const SegTile *)nullptr, up.dmask);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, hist, hist, (u64)256 * nb, nullptr);
        if (rc) return rc;
        {
            StageTimer ts(ctx, LRGE_T_RS_SCATTER);
            hipLaunchKernelGGL((k_rs_scatter<false, RS_MODE_KEYS, 8>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, (const u64 *)nullptr, ko, (u64 *)nullptr, n, shift, nb,
                               hist, (const SegTile *)nullptr, up);
            KCHK(ctx);
            ts.stop();
            ctx->counters[LRGE_C_RS_SCATTER_LAUNCHES] += 1;
            ctx->counters[LRGE_C_RS_SCATTER_ITEMS] += n;
            ctx->counters[LRGE_C_RS_SCATTER_BYTES] += 16 * n;
        }
        nd_valid = up.nd_out != nullptr;
        u64 *t = ki; ki = ko; ko = t;
    }
    if (nd) sc.drop(nd);
    sc.drop(hist);
    *res = ki;
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

// (Round 6 tried the keys read COALESCED -- a wavefront owning 1 024 consecutive keys as 16 rows of 64, the predecessor by a DPP shift, the
// head bits through one ballot per row into the thread-owned layout below: exact, and SLOWER -- 4.5-4.8 ms per launch against 3.7 for a
// 20-GB part, with LDS permutes or with DPP alike.  Each thread reading its own 128-byte line with eight 16-byte loads stays.)
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

// flags: the 16 head bits of every thread's line, kept for k_heads_fill (instead of re-reading 128 bytes of keys); stored as a dword per
// thread since round 6 (HFLAG_BITS=16: the u16 array of rounds 2-5 -- 2-byte stores and loads are slow on this chip)
#ifndef HFLAG_BITS
#define HFLAG_BITS 32
#endif
#if HFLAG_BITS == 16
typedef u16 hflag_t;
#else
typedef u32 hflag_t;
#endif
__global__ __launch_bounds__(HC_THREADS) void k_heads_count(const u64 *__restrict__ keys, u64 n, u32 shift, u32 *__restrict__ bcount,
                                                            hflag_t *__restrict__ flags, const u32 *__restrict__ seg_start = nullptr, u32 n_seg = 0) {
    __shared__ u32 ws[HC_THREADS / 64];
    const u64 base = (u64)blockIdx.x * HC_TILE + (u64)threadIdx.x * HC_ITEMS;
    u32 f = hc_load_flags(keys, n, shift, base);
    if (seg_start && base < n) f |= hc_boundary_flags(seg_start, n_seg, n, base);
    flags[(u64)blockIdx.x * HC_THREADS + threadIdx.x] = (hflag_t)f;
    u32 c = (u32)__popc(f);
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if (lane_id() == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { u32 t = 0; for (int w = 0; w < HC_THREADS / 64; ++w) t += ws[w]; bcount[blockIdx.x] = t; }
}

__global__ __launch_bounds__(HC_THREADS) void k_heads_fill(const hflag_t *__restrict__ flags, const u32 *__restrict__ boff,
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
    ALLOC_OR_FAIL(fl, sc, hflag_t, (size_t)nb * HC_THREADS);
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
    ALLOC_OR_FAIL(fl, sc, hflag_t, (size_t)nb * HC_THREADS);
    hipLaunchKernelGGL(k_heads_count, dim3(nb), dim3(HC_THREADS), 0, ctx->stream, keys, n, shift, bc, fl, seg_start, n_seg);
    KCHK(ctx);
    int rc = scan_exclusive_u32(ctx, sc, bc, bc, nb, d_tot);
    if (rc) return rc;
    HIPCHK(ctx, ctx->d2h(n_heads, d_tot, 4, ctx->stream));
    HIPCHK(ctx, ctx->d2h_sync(ctx->stream));             // (also lands what the caller queued with ctx->d2h: the segment starts of a segment-packed index)
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
struct SegDesc { u32 start, len, seg, src; };     // src: offset of the segment in the INPUT array (the output goes to start)
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
    const u64 *src = pk_in + sd.src;                         // (input offset: the output goes to sd.start)
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

// out[i] = src[idx[i]] (idx[i] = ~0: left alone)
__global__ void k_gather_index_u32(const u32 *__restrict__ src, const u32 *__restrict__ idx, u32 n, u32 *__restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && idx[i] != 0xFFFFFFFFu) out[i] = src[idx[i]];
}

// out[i] = src[i * stride] (the starts of the 256 first-digit segments out of a scanned histogram)
__global__ void k_gather_strided_u32(const u32 *__restrict__ src, u64 stride, u32 n, u32 *__restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[(u64)i * stride];
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
// Tile bookkeeping of the segmented passes, on the device: the sort has no host round trip (a part of full-size C5 has 610 000
// tiles -- building them on the host and copying them cost two stream drains and ~10 ms per part).
// tb[s] = tiles in front of segment s (exclusive prefix of ceil(count / RS_TILE)), tb[n_seg] = their number.  One block; n_seg <= 4096.
__global__ __launch_bounds__(1024) void k_seg_tile_scan(const u32 *__restrict__ starts, u32 n_seg, u32 *__restrict__ tb) {
    __shared__ u32 wsum[16];
    const u32 per = (n_seg + 1023) / 1024, s0 = threadIdx.x * per;
    u32 mine = 0;
    for (u32 j = 0; j < per; ++j) { const u32 s = s0 + j; if (s < n_seg) mine += (starts[s + 1] - starts[s] + RS_TILE - 1) / RS_TILE; }
    const u32 inc = wave_incl_scan_u32(mine);
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    u32 run = inc - mine;
    for (u32 w = 0; w < (threadIdx.x >> 6); ++w) run += wsum[w];
    for (u32 j = 0; j < per; ++j) { const u32 s = s0 + j; if (s < n_seg) { tb[s] = run; run += (starts[s + 1] - starts[s] + RS_TILE - 1) / RS_TILE; } }
    if (threadIdx.x == 1023) tb[n_seg] = run;
}
// tiles[t] for t < max_tiles (the grid of every segmented pass): beyond tb[n_seg] an empty tile whose counts go to the spare
// histogram slot 256 * max_tiles
__global__ __launch_bounds__(256) void k_seg_tile_fill(const u32 *__restrict__ starts, const u32 *__restrict__ tb, u32 n_seg, u32 max_tiles, SegTile *__restrict__ tiles) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= max_tiles) return;
    if (t >= tb[n_seg]) { tiles[t] = SegTile{0u, 0u, 256u * max_tiles, 0u, 0u, 0u, 0u, 0u}; return; }
    u32 lo = 0, hi = n_seg;                            // the last s with tb[s] <= t (it has tiles: tb[s + 1] > t)
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (tb[mid] <= t) lo = mid; else hi = mid; }
    const u32 lt = t - tb[lo], c = starts[lo + 1] - starts[lo], nt = tb[lo + 1] - tb[lo];
    const u32 left = c - lt * RS_TILE;
    tiles[t] = SegTile{starts[lo] + lt * RS_TILE, left < RS_TILE ? left : (u32)RS_TILE, 256u * tb[lo] + lt, nt, lo, 0u, 0u, 0u};
}
// where segment (b0, q) begins after pass A2: the scanned count of digit q in the first tile of b0 (hist index 256 tb + q nt); a
// b0 without tiles is empty and its segments begin where it does.  fine[256 << e] = n.
__global__ __launch_bounds__(256) void k_seg_fine_starts(const u32 *__restrict__ hist_scanned, const u32 *__restrict__ starts, const u32 *__restrict__ tb, u32 e, u32 n,
                                                         u32 *__restrict__ fine) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x, nf = 256u << e;
    if (i > nf) return;
    if (i == nf) { fine[i] = n; return; }
    const u32 s = i >> e, q = i & ((1u << e) - 1), nt = tb[s + 1] - tb[s];
    fine[i] = nt ? hist_scanned[256u * tb[s] + q * nt] : starts[s];
}
__global__ void k_store_u32(u32 *p, u32 v) { *p = v; }

static int index_sort_segpacked(lrge_hip_ctx *ctx, Scratch &sc, u64 *kx, u64 *ky, u64 *k1, u64 *v1, u64 n, int nbits, u32 ybits, u32 pos1, u32 e,
                                u64 **res, u32 **d_seg_start) {
    // e > 0: e more hash bits are implied by the segment -- the NEXT bits of the byte-reversed order, the top e bits q of the second
    // byte -- because the word would be too narrow without (large parts: read ids of 19-20 bits) or because what is left of the hash then
    // takes a pass less.  Pass A2 is a most-significant-digit pass on q inside each of A's 256 segments (pairs in, packed words out:
    // RS_MODE_PACKQ), leaving 256 << e segments numbered b0 << e | q.  The entry keeps R, the low nr = nbits - 8 - e bits of the hash's
    // significance string (hash_to_sig): R compares like the order itself, so the keys-only LSD passes behind cut it into 8-bit
    // digits from the bottom, the last one taking what is left.  k = 19, e = 6: 32 + 24 + 3 x 16 = 104 bytes per entry through the
    // scatters and five histograms (round 4, e = 2 with byte-aligned digits: 120 and six; the plain pair sort: 160).
    const int nr = nbits - 8 - (int)e, passes = (nr + 7) / 8;      // LSD passes over R
    const u32 nb = (u32)div_up(n, RS_TILE), n_seg = 256u << e;
    const u32 max_tiles = nb + n_seg;                  // every segment ends in at most one partial tile
    ALLOC_OR_FAIL(hist, sc, u32, (u64)256 * (max_tiles + 1) + 256);
    // next-pass digit bytes (UnpackParams::nd_out): every scatter that writes packed words also leaves the byte the NEXT pass ranks by, and
    // that pass's histogram reads 1 byte per entry instead of 8 (option NO_DIGIT_BYTES: the histograms read the words, rounds 3-5)
    u8 *nd = ctx->opt("NO_DIGIT_BYTES") ? nullptr : sc.get<u8>(n + 64);
    if (!nd) { (void)hipGetLastError(); ctx->err.clear(); }
    bool nd_valid = false;                             // nd holds the digits of the pass about to run
    auto nd_of = [&](int j_next) -> UnpackParams {      // what the scatter in front of LSD pass j_next adds to its parameters
        UnpackParams q{0, 0, 0, 0, 0, nullptr, 0, 0};
        if (nd && j_next < passes) { const int w_ = nr - 8 * j_next >= 8 ? 8 : nr - 8 * j_next; q.nd_out = nd; q.nd_shift = ybits + 8u * (u32)j_next; q.nd_mask = (1u << w_) - 1u; }
        return q;
    };
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
    u32 *d_b = sc.get<u32>(n_seg + 1), *d_c = e ? sc.get<u32>(257) : nullptr, *d_tb = sc.get<u32>(n_seg + 1);
    if (!d_b || !d_tb || (e && !d_c)) return LRGE_ERR_DEVICE;
    u32 *coarse = e ? d_c : d_b;                       // the 256 segments of pass A
    hipLaunchKernelGGL(k_gather_strided_u32, dim3(1), dim3(256), 0, ctx->stream, hist, (u64)nb, 256u, coarse);
    hipLaunchKernelGGL(k_store_u32, dim3(1), dim3(1), 0, ctx->stream, coarse + 256, (u32)n);
    KCHK(ctx);
    ALLOC_OR_FAIL(d_tiles, sc, u32, (size_t)max_tiles * (sizeof(SegTile) / 4) + 8);
    u32 cur_tiles = nb + 256;                          // the grid of the passes over `cur_seg` segments
    hipLaunchKernelGGL(k_seg_tile_scan, dim3(1), dim3(1024), 0, ctx->stream, (const u32 *)coarse, 256u, d_tb);
    hipLaunchKernelGGL(k_seg_tile_fill, dim3(div_up(cur_tiles, 256)), dim3(256), 0, ctx->stream, (const u32 *)coarse, (const u32 *)d_tb, 256u, cur_tiles, (SegTile *)d_tiles);
    KCHK(ctx);
    u64 *pi = kx, *po = ky;          // (the sketch's pair buffers are free once pass A has read them: they carry the packed words)
    if (e) {
        // ---- pass A2: inside every segment by q, packing ----
        const u32 qm = (1u << e) - 1;
        hipLaunchKernelGGL(k_rs_hist<true>, dim3(cur_tiles), dim3(RS_THREADS), 0, ctx->stream, k1, n, 16 - (int)e, cur_tiles, hist, (const SegTile *)d_tiles, qm);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, hist, hist, (u64)256 * cur_tiles, nullptr); if (rc) return rc;
        hipLaunchKernelGGL(k_seg_fine_starts, dim3(div_up(n_seg + 1, 256)), dim3(256), 0, ctx->stream, (const u32 *)hist, (const u32 *)coarse, (const u32 *)d_tb, e, (u32)n, d_b);
        KCHK(ctx);
        {
            StageTimer ts(ctx, LRGE_T_RS_SCATTER);
            const UnpackParams nq = nd_of(0);
            hipLaunchKernelGGL((k_rs_scatter<true, RS_MODE_PACKQ>), dim3(cur_tiles), dim3(RS_THREADS), 0, ctx->stream, k1, v1, pi, (u64 *)nullptr, n, 16 - (int)e, cur_tiles, hist,
                               (const SegTile *)d_tiles, UnpackParams{ybits, pos1, e, qm, (u32)nbits, nq.nd_out, nq.nd_shift, nq.nd_mask});
            nd_valid = nq.nd_out != nullptr;
            KCHK(ctx);
            ts.stop();
        }
        ctx->counters[LRGE_C_RS_SCATTER_LAUNCHES] += 1; ctx->counters[LRGE_C_RS_SCATTER_ITEMS] += n; ctx->counters[LRGE_C_RS_SCATTER_BYTES] += 24 * n;
        cur_tiles = max_tiles;
        hipLaunchKernelGGL(k_seg_tile_scan, dim3(1), dim3(1024), 0, ctx->stream, (const u32 *)d_b, n_seg, d_tb);
        hipLaunchKernelGGL(k_seg_tile_fill, dim3(div_up(cur_tiles, 256)), dim3(256), 0, ctx->stream, (const u32 *)d_b, (const u32 *)d_tb, n_seg, cur_tiles, (SegTile *)d_tiles);
        KCHK(ctx);
    }
    // ---- the digits of R, least significant first (e == 0: the first of them reads the pairs and packs) ----
    for (int j = 0; j < passes; ++j) {
        const bool first = !e && j == 0;
        const int w = nr - 8 * j >= 8 ? 8 : nr - 8 * j;
        const u32 dm = (1u << w) - 1u;
        const int pshift = (int)ybits + 8 * j;                           // where digit j sits in the packed word
        if (first) hipLaunchKernelGGL(k_rs_hist<true>, dim3(cur_tiles), dim3(RS_THREADS), 0, ctx->stream, k1, n, 0, cur_tiles, hist, (const SegTile *)d_tiles, dm, SlotSrc(), 0u, (u32)nbits);
        else if (nd_valid) hipLaunchKernelGGL(k_rs_hist<true>, dim3(cur_tiles), dim3(RS_THREADS), 0, ctx->stream, (const u64 *)nd, n, 0, cur_tiles, hist, (const SegTile *)d_tiles, dm, SlotSrc(), 0u, 0u, 2u);
        else hipLaunchKernelGGL(k_rs_hist<true>, dim3(cur_tiles), dim3(RS_THREADS), 0, ctx->stream, pi, n, pshift, cur_tiles, hist, (const SegTile *)d_tiles, dm);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, hist, hist, (u64)256 * cur_tiles, nullptr); if (rc) return rc;
        StageTimer ts(ctx, LRGE_T_RS_SCATTER);
        const UnpackParams nq = nd_of(j + 1);
        if (first) hipLaunchKernelGGL((k_rs_scatter<true, RS_MODE_PACK>), dim3(cur_tiles), dim3(RS_THREADS), 0, ctx->stream, k1, v1, pi, (u64 *)nullptr, n, pshift, cur_tiles, hist,
                                      (const SegTile *)d_tiles, UnpackParams{ybits, pos1, 0, dm, (u32)nbits, nq.nd_out, nq.nd_shift, nq.nd_mask});
        else hipLaunchKernelGGL((k_rs_scatter<true, RS_MODE_KEYS>), dim3(cur_tiles), dim3(RS_THREADS), 0, ctx->stream, pi, (const u64 *)nullptr, po, (u64 *)nullptr, n, pshift, cur_tiles,
                                hist, (const SegTile *)d_tiles, UnpackParams{0, 0, 0, dm, 0, nq.nd_out, nq.nd_shift, nq.nd_mask});
        KCHK(ctx);
        ts.stop();
        nd_valid = nq.nd_out != nullptr;
        ctx->counters[LRGE_C_RS_SCATTER_LAUNCHES] += 1; ctx->counters[LRGE_C_RS_SCATTER_ITEMS] += n; ctx->counters[LRGE_C_RS_SCATTER_BYTES] += (first ? 24 : 16) * n;
        if (!first) { u64 *t = pi; pi = po; po = t; }
    }
    if (d_c) sc.drop(d_c);
    sc.drop(d_tb);
    if (nd) sc.drop(nd);
    sc.drop(hist); sc.drop((u32 *)d_tiles);
    *res = pi; *d_seg_start = d_b;
    return LRGE_OK;
}

// The same sort over SEGW entries (RS_MODE_DW* above): wx = the words, dy = the u32 DIG array, both as the sketch left them; k1 / d1 =
// buffers of n + 1 u64 / u32.  Pass A moves 12 bytes per entry instead of 16 (and its histogram reads 4 instead of 8), pass A2 / the
// first LSD pass reads 12 instead of 16; from there on the entries are the packed words of index_sort_segpacked, bit for bit.
// WaveSrc (round 6): pass A reads the wave-dense sketch's slots (k_sketch.h: k_sketch_wave) in place of the dense (wx, dy) arrays -- through
// the slot-source form of its two kernels (SlotSrc) --, DIG members of 16 bits (d1 is then a u16 array too).  The slots are
// released as soon as pass A has read them and the second word buffer is taken in their place; *spare = the word buffer that does not
// hold the result.
struct WaveSrc { u64 *wx; wdig_t *wd; u32 *cnt, *offs; u32 n_waves, cap; };      // offs: exclusive scan of cnt
static int index_sort_segw(lrge_hip_ctx *ctx, Scratch &sc, u64 *wx, u32 *dy, u64 *k1, u32 *d1, u64 n, int nbits, u32 ybits, u32 pos1, u32 e,
                           u64 **res, u32 **d_seg_start, const WaveSrc *ws = nullptr, u64 **spare = nullptr) {
    (void)pos1;
    const u32 dig16 = (ws && sizeof(wdig_t) == 2) ? 1u : 0u, dkey = dig16 ? 3u : 1u;      // width of the DIG arrays (UnpackParams::dig16; k_rs_hist's key32)
    const int nr = nbits - 8 - (int)e, passes = (nr + 7) / 8;      // LSD passes over R
    const u32 nb = (u32)div_up(n, RS_TILE), n_seg = 256u << e;
    const u32 max_tiles = nb + n_seg;
    ALLOC_OR_FAIL(hist, sc, u32, (u64)256 * (max_tiles + 1) + 256);
    // next-pass digit bytes (UnpackParams::nd_out): every scatter that writes packed words also leaves the byte the NEXT pass ranks by, and
    // that pass's histogram reads 1 byte per entry instead of 8 (option NO_DIGIT_BYTES: the histograms read the words, rounds 3-5)
    u8 *nd = ctx->opt("NO_DIGIT_BYTES") ? nullptr : sc.get<u8>(n + 64);
    if (!nd) { (void)hipGetLastError(); ctx->err.clear(); }
    bool nd_valid = false;                             // nd holds the digits of the pass about to run
    auto nd_of = [&](int j_next) -> UnpackParams {      // what the scatter in front of LSD pass j_next adds to its parameters
        UnpackParams q{0, 0, 0, 0, 0, nullptr, 0, 0};
        if (nd && j_next < passes) { const int w_ = nr - 8 * j_next >= 8 ? 8 : nr - 8 * j_next; q.nd_out = nd; q.nd_shift = ybits + 8u * (u32)j_next; q.nd_mask = (1u << w_) - 1u; }
        return q;
    };
    // ---- pass A: (DIG, word) by b0 ----
    const u32 nbA = nb;
    if (ws) {
        // dense tiles of RS_TILE entries over the VIRTUAL dense array: entry i lives in the wavefront slot v = the last one with offs[v] <= i,
        // at index v * cap + i - offs[v] (k_prims.h: SlotSrc / rs_for_slot_items; a tile spans two or three slots of ~2 000-2 800 entries).
        // (One tile per slot -- half-full tiles, twice as many, digit runs of 8 entries -- took 34 ms per part instead of 16.)
        ALLOC_OR_FAIL(tile_chunk, sc, u32, (size_t)nb + 1);
        hipLaunchKernelGGL(k_tile_chunks, dim3((u32)div_up(nb, 256)), dim3(256), 0, ctx->stream, (const u32 *)ws->offs, ws->n_waves, nb, tile_chunk);
        KCHK(ctx);
        ALLOC_OR_FAIL(tile_desc, sc, u32, (size_t)nb * 8 + 8);
        hipLaunchKernelGGL(k_tile_desc, dim3((u32)div_up(nb, 256)), dim3(256), 0, ctx->stream, (const u32 *)ws->offs, ws->n_waves, (u32)n, nb, tile_desc);
        KCHK(ctx);
        SlotSrc src{ws->wx, ws->offs, tile_chunk, ws->n_waves, ws->cap, (u32)n, ws->wd, tile_desc};
        hipLaunchKernelGGL((k_rs_hist<false, 8, true>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, (const u64 *)nullptr, n, 8, nb, hist, (const SegTile *)nullptr, 255u, src, 0u, 0u, dkey);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, hist, hist, (u64)256 * nb, nullptr); if (rc) return rc;
        StageTimer ts(ctx, LRGE_T_RS_SCATTER);
        hipLaunchKernelGGL((k_rs_scatter<false, RS_MODE_DW, 8, true>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, (const u64 *)nullptr, (const u64 *)nullptr, (u64 *)d1, k1, n, 8, nb, hist,
                           (const SegTile *)nullptr, UnpackParams{0, 0, 0, 255, 0, nullptr, 0, 0, dig16}, src);
        KCHK(ctx);
        ts.stop();
        ctx->counters[LRGE_C_RS_SCATTER_LAUNCHES] += 1; ctx->counters[LRGE_C_RS_SCATTER_ITEMS] += n; ctx->counters[LRGE_C_RS_SCATTER_BYTES] += (dig16 ? 20 : 24) * n;
        sc.drop(tile_chunk); sc.drop(tile_desc);
    } else {
        hipLaunchKernelGGL(k_rs_hist<false>, dim3(nb), dim3(RS_THREADS), 0, ctx->stream, (const u64 *)dy, n, 8, nb, hist, (const SegTile *)nullptr, 255u, SlotSrc(), 0u, 0u, 1u);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, hist, hist, (u64)256 * nb, nullptr); if (rc) return rc;
        StageTimer ts(ctx, LRGE_T_RS_SCATTER);
        hipLaunchKernelGGL((k_rs_scatter<false, RS_MODE_DW>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, (const u64 *)dy, (const u64 *)wx, (u64 *)d1, k1, n, 8, nb, hist,
                           (const SegTile *)nullptr, UnpackParams{0, 0, 0, 255, 0});
        KCHK(ctx);
        ts.stop();
        ctx->counters[LRGE_C_RS_SCATTER_LAUNCHES] += 1; ctx->counters[LRGE_C_RS_SCATTER_ITEMS] += n; ctx->counters[LRGE_C_RS_SCATTER_BYTES] += 24 * n;
    }
    u32 *d_b = sc.get<u32>(n_seg + 1), *d_c = e ? sc.get<u32>(257) : nullptr, *d_tb = sc.get<u32>(n_seg + 1);
    if (!d_b || !d_tb || (e && !d_c)) return LRGE_ERR_DEVICE;
    u32 *coarse = e ? d_c : d_b;                       // the 256 segments of pass A
    hipLaunchKernelGGL(k_gather_strided_u32, dim3(1), dim3(256), 0, ctx->stream, hist, (u64)nbA, 256u, coarse);
    hipLaunchKernelGGL(k_store_u32, dim3(1), dim3(1), 0, ctx->stream, coarse + 256, (u32)n);
    KCHK(ctx);
    ALLOC_OR_FAIL(d_tiles, sc, u32, (size_t)max_tiles * (sizeof(SegTile) / 4) + 8);
    u32 cur_tiles = nb + 256;
    hipLaunchKernelGGL(k_seg_tile_scan, dim3(1), dim3(1024), 0, ctx->stream, (const u32 *)coarse, 256u, d_tb);
    hipLaunchKernelGGL(k_seg_tile_fill, dim3(div_up(cur_tiles, 256)), dim3(256), 0, ctx->stream, (const u32 *)coarse, (const u32 *)d_tb, 256u, cur_tiles, (SegTile *)d_tiles);
    KCHK(ctx);
    if (ws) {        // the slots have been read: they go, the second word buffer comes (in stream order: the arena recycles on ctx->stream)
        sc.drop(ws->wx); sc.drop(ws->wd); sc.drop(ws->cnt); sc.drop(ws->offs);
        wx = sc.get<u64>(n + 1);
        if (!wx) return LRGE_ERR_DEVICE;
    }
    u64 *pi = wx, *po = k1;          // (the sketch's word buffer is free once pass A has read it; k1 once the packing pass has)
    if (e) {
        // ---- pass A2: inside every segment by the top e bits of b1, packing ----
        const u32 qm = (1u << e) - 1;
        hipLaunchKernelGGL(k_rs_hist<true>, dim3(cur_tiles), dim3(RS_THREADS), 0, ctx->stream, (const u64 *)d1, n, 8 - (int)e, cur_tiles, hist, (const SegTile *)d_tiles, qm, SlotSrc(), 0u, 0u, dkey);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, hist, hist, (u64)256 * cur_tiles, nullptr); if (rc) return rc;
        hipLaunchKernelGGL(k_seg_fine_starts, dim3(div_up(n_seg + 1, 256)), dim3(256), 0, ctx->stream, (const u32 *)hist, (const u32 *)coarse, (const u32 *)d_tb, e, (u32)n, d_b);
        KCHK(ctx);
        {
            StageTimer ts(ctx, LRGE_T_RS_SCATTER);
            const UnpackParams nq = nd_of(0);
            hipLaunchKernelGGL((k_rs_scatter<true, RS_MODE_DWQ>), dim3(cur_tiles), dim3(RS_THREADS), 0, ctx->stream, (const u64 *)d1, (const u64 *)k1, pi, (u64 *)nullptr, n, 8 - (int)e, cur_tiles, hist,
                               (const SegTile *)d_tiles, UnpackParams{ybits, pos1, e, qm, (u32)nbits, nq.nd_out, nq.nd_shift, nq.nd_mask, dig16});
            nd_valid = nq.nd_out != nullptr;
            KCHK(ctx);
            ts.stop();
        }
        ctx->counters[LRGE_C_RS_SCATTER_LAUNCHES] += 1; ctx->counters[LRGE_C_RS_SCATTER_ITEMS] += n; ctx->counters[LRGE_C_RS_SCATTER_BYTES] += 20 * n;
        cur_tiles = max_tiles;
        hipLaunchKernelGGL(k_seg_tile_scan, dim3(1), dim3(1024), 0, ctx->stream, (const u32 *)d_b, n_seg, d_tb);
        hipLaunchKernelGGL(k_seg_tile_fill, dim3(div_up(cur_tiles, 256)), dim3(256), 0, ctx->stream, (const u32 *)d_b, (const u32 *)d_tb, n_seg, cur_tiles, (SegTile *)d_tiles);
        KCHK(ctx);
    }
    // ---- the digits of R, least significant first (e == 0: the first of them reads (DIG, word) and packs) ----
    for (int j = 0; j < passes; ++j) {
        const bool first = !e && j == 0;
        const int w = nr - 8 * j >= 8 ? 8 : nr - 8 * j;
        const u32 dm = (1u << w) - 1u;
        const int pshift = (int)ybits + 8 * j;                           // where digit j sits in the packed word
        // (first: digit 0 of R lies in the word's own hash field -- nbits - 16 >= 8 is the caller's condition -- so the histogram reads the words)
        if (!first && nd_valid) hipLaunchKernelGGL(k_rs_hist<true>, dim3(cur_tiles), dim3(RS_THREADS), 0, ctx->stream, (const u64 *)nd, n, 0, cur_tiles, hist, (const SegTile *)d_tiles, dm, SlotSrc(), 0u, 0u, 2u);
        else hipLaunchKernelGGL(k_rs_hist<true>, dim3(cur_tiles), dim3(RS_THREADS), 0, ctx->stream, first ? k1 : pi, n, pshift, cur_tiles, hist, (const SegTile *)d_tiles, dm);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, hist, hist, (u64)256 * cur_tiles, nullptr); if (rc) return rc;
        StageTimer ts(ctx, LRGE_T_RS_SCATTER);
        const UnpackParams nq = nd_of(j + 1);
        if (first) hipLaunchKernelGGL((k_rs_scatter<true, RS_MODE_DWP>), dim3(cur_tiles), dim3(RS_THREADS), 0, ctx->stream, (const u64 *)d1, (const u64 *)k1, pi, (u64 *)nullptr, n, pshift, cur_tiles, hist,
                                      (const SegTile *)d_tiles, UnpackParams{ybits, pos1, 0, dm, (u32)nbits, nq.nd_out, nq.nd_shift, nq.nd_mask, dig16});
        else hipLaunchKernelGGL((k_rs_scatter<true, RS_MODE_KEYS>), dim3(cur_tiles), dim3(RS_THREADS), 0, ctx->stream, pi, (const u64 *)nullptr, po, (u64 *)nullptr, n, pshift, cur_tiles,
                                hist, (const SegTile *)d_tiles, UnpackParams{0, 0, 0, dm, 0, nq.nd_out, nq.nd_shift, nq.nd_mask});
        KCHK(ctx);
        ts.stop();
        nd_valid = nq.nd_out != nullptr;
        ctx->counters[LRGE_C_RS_SCATTER_LAUNCHES] += 1; ctx->counters[LRGE_C_RS_SCATTER_ITEMS] += n; ctx->counters[LRGE_C_RS_SCATTER_BYTES] += (first ? 20 : 16) * n;
        if (!first) { u64 *t = pi; pi = po; po = t; }
    }
    if (d_c) sc.drop(d_c);
    sc.drop(d_tb);
    if (nd) sc.drop(nd);
    sc.drop(hist); sc.drop((u32 *)d_tiles);
    *res = pi; *d_seg_start = d_b;
    if (spare) *spare = po;
    return LRGE_OK;
}
