// k_route.h -- the index of a multi-GPU run whose TARGET SKETCH is sharded too (lrge_hip_index_build_sharded).
//
// k_restrict.h keeps every rank's index down to the keys its own streamed reads carry, but every rank still sketches
// all targets and probes its key set with every target minimizer: a fixed 7 of 14 ms per rank at C4 / 8 ranks, which
// caps the strong scaling of the forward strategy at ~2.4x.  Here rank r sketches only ITS contiguous share of the target
// reads; what the ranks need of each other travels in three exchanges (comm.h):
//   1. an all-gather of the ranks' key sets (blocked Bloom filters of one agreed size; afterwards transposed so that
//      word w of all ranks is one contiguous run: ONE line answers "which ranks ask for this key");
//   2. a variable-size all-to-all of the kept entries: an entry goes to every rank whose key set holds its key
//      (complete position lists on every rank that asks, hence identical mm_idx_get answers);
//   3. a variable-size all-to-all of the bare hashes, each to the rank that OWNS it (a 1/world share of the hash space):
//      the owner run-length counts them, and the all-reduce of k_restrict.h's statistics vector makes mid_occ global.
// Entries leave a rank in sketch order (stable compaction per destination) and arrive concatenated in rank order; the
// ranks hold contiguous ascending ranges of the target reads, so every rank's kept entries are in the order the one
// index would hold them: the rest of the build (stable sort by hash, table) is the single-GPU code.
#pragma once
#include "internal.h"
#include "k_prims.h"
#include "k_restrict.h"

#define ROUTE_MAX_WORLD 16

struct KeySetAll {
    const u64 *bits;    // interleaved: word w of rank r at bits[w * STRIDE + r], STRIDE = 8 (world <= 8) or 16 slots per word, unused slots zero
    u64 word_mask;      // n_words - 1
    u32 world;
};

// the Bloom form of k_restrict.h's KeySet with no slicing by the top digit (top_shift = 2k): same word and bit choice
__device__ __forceinline__ void ksa_locate(u64 word_mask, u64 hash, u64 *word, u64 *mask) {
    const u64 m = ks_mix(hash);
    *word = (m >> 20) & word_mask;
    *mask = 1ULL << (m & 63) | 1ULL << ((m >> 6) & 63) | 1ULL << ((m >> 12) & 63);
}

// [rank][word] (what the all-gather delivers) -> [word][STRIDE slots, rank r in slot r, the rest zero]: one thread per word
// (coalesced reads of every rank's filter, one contiguous STRIDE * 8 byte run written per thread)
template <int STRIDE>
__global__ __launch_bounds__(256) void k_keyset_interleave(const u64 *__restrict__ in, u64 n_words, u32 world, u64 *__restrict__ out) {
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    u64 v[STRIDE];
#pragma unroll
    for (int r = 0; r < STRIDE; ++r) v[r] = r < (int)world ? in[(u64)r * n_words + w] : 0;
#pragma unroll
    for (int r = 0; r < STRIDE; r += 2) *(ulonglong2 *)(out + w * STRIDE + r) = make_ulonglong2(v[r], v[r + 1]);
}

// entries of a shard carry the read's index inside the shard: make it the index in the whole target set
__global__ __launch_bounds__(256) void k_add_u64(u64 *__restrict__ a, u64 n, u64 add) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += add;
}

struct RouteArgs {
    const u64 *x; const u64 *y;     // entries: packed (y null, hash = x >> kshift) or (hash, y) pairs (kshift 0)
    u64 n; u32 kshift;
    KeySetAll ks;
    u32 n_tiles;
};

// flags[i] = mask of the ranks that ask for entry i's key | owner << 16; cnt[s * n_tiles + tile] = entries of the tile
// that go to stream s (s < world: kept for rank s; s >= world: owned by rank s - world).
// STRIDE (8 or 16) words per filter line, a compile-time constant: every entry's line is fetched with STRIDE / 2 16-byte
// loads issued back to back, and all rows' loads are in flight before the first is tested (with a run-time rank count the
// compiler made it one load and one wait per rank and row: 3.5 ms per 30 M entries, now latency-hidden).
template <int STRIDE>
__global__ __launch_bounds__(RF_THREADS) void k_route_count(RouteArgs A, u32 *__restrict__ flags, u32 *__restrict__ cnt) {
    __shared__ u32 sc[2 * ROUTE_MAX_WORLD];
    const u32 W = A.ks.world;
    if (threadIdx.x < 2 * W) sc[threadIdx.x] = 0;
    __syncthreads();
    const u64 tile_base = (u64)blockIdx.x * RF_TILE;
    u64 hs[RF_ITEMS];
#pragma unroll
    for (int r = 0; r < RF_ITEMS; ++r) {
        const u64 i = tile_base + (u64)r * RF_THREADS + threadIdx.x;
        hs[r] = i < A.n ? A.x[i] >> A.kshift : 0;
    }
    u32 keep[STRIDE], own[STRIDE];
#pragma unroll
    for (int s = 0; s < STRIDE; ++s) { keep[s] = 0; own[s] = 0; }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        ulonglong2 ln[RF_ITEMS / 2][STRIDE / 2];
        u64 ms[RF_ITEMS / 2];
#pragma unroll
        for (int rr = 0; rr < RF_ITEMS / 2; ++rr) {
            u64 w;
            ksa_locate(A.ks.word_mask, hs[half * (RF_ITEMS / 2) + rr], &w, &ms[rr]);
            const ulonglong2 *line = (const ulonglong2 *)(A.ks.bits + w * STRIDE);
#pragma unroll
            for (int q = 0; q < STRIDE / 2; ++q) ln[rr][q] = line[q];
        }
#pragma unroll
        for (int rr = 0; rr < RF_ITEMS / 2; ++rr) {
            const int r = half * (RF_ITEMS / 2) + rr;
            const u64 i = tile_base + (u64)r * RF_THREADS + threadIdx.x;
            if (i < A.n) {
                const u64 m = ms[rr];
                u32 want = 0;
#pragma unroll
                for (int q = 0; q < STRIDE / 2; ++q) {
                    want |= ((ln[rr][q].x & m) == m ? 1u : 0u) << (2 * q);
                    want |= ((ln[rr][q].y & m) == m ? 1u : 0u) << (2 * q + 1);
                }
                const u32 owner = (u32)(((ks_mix(hs[r]) >> 32) * (u64)W) >> 32);
                flags[i] = want | owner << 16;
#pragma unroll
                for (int s = 0; s < STRIDE; ++s) { keep[s] += (want >> s) & 1u; own[s] += owner == (u32)s; }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < STRIDE; ++s) {
        if (s < (int)W) {
            u32 a = keep[s], b = own[s];
            for (int d = 32; d > 0; d >>= 1) { a += __shfl_down(a, d, 64); b += __shfl_down(b, d, 64); }
            if (lane_id() == 0) { if (a) atomicAdd(&sc[s], a); if (b) atomicAdd(&sc[W + s], b); }
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * W) cnt[(u64)threadIdx.x * A.n_tiles + blockIdx.x] = sc[threadIdx.x];
}

// one block per stream: exclusive scan of its row of cnt[] in place, the row's total to tot[s]
__global__ __launch_bounds__(1024) void k_route_scan(u32 *__restrict__ cnt, u32 n_tiles, u32 *__restrict__ tot) {
    __shared__ u32 ws[16];
    __shared__ u32 carry;
    u32 *row = cnt + (u64)blockIdx.x * n_tiles;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u32 base = 0; base < n_tiles; base += 1024) {
        const u32 i = base + threadIdx.x;
        const u32 v = i < n_tiles ? row[i] : 0;
        u32 inc = v;
        for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, d, 64); if ((int)lane_id() >= d) inc += t; }
        if (lane_id() == 63) ws[threadIdx.x >> 6] = inc;
        __syncthreads();
        u32 before = carry;
        for (u32 w = 0; w < (threadIdx.x >> 6); ++w) before += ws[w];
        if (i < n_tiles) row[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) tot[blockIdx.x] = carry;
}

struct RouteBases { u64 keep[ROUTE_MAX_WORLD]; u64 own[ROUTE_MAX_WORLD]; };   // first element of every destination's run in the send buffers

// order-preserving scatter of the entries into the send buffers (grouped by destination)
__global__ __launch_bounds__(RF_THREADS) void k_route_write(RouteArgs A, const u32 *__restrict__ flags, const u32 *__restrict__ off, RouteBases B,
                                                            u64 *__restrict__ out_x, u64 *__restrict__ out_y, u64 *__restrict__ out_hash,
                                                            u32 *__restrict__ out_hash32) {
    // out_hash32 != null: the hash has at most 32 bits (k <= 16) and travels as 4 bytes (out_hash unused)
    __shared__ u32 cw[RF_ITEMS][RF_THREADS / 64];
    const u32 W = A.ks.world;
    const u64 tile_base = (u64)blockIdx.x * RF_TILE;
    const u32 w = threadIdx.x >> 6, lane = lane_id();
    u64 xs[RF_ITEMS]; u32 fl[RF_ITEMS];
#pragma unroll
    for (int r = 0; r < RF_ITEMS; ++r) {
        const u64 i = tile_base + (u64)r * RF_THREADS + threadIdx.x;
        xs[r] = i < A.n ? A.x[i] : 0;
        fl[r] = i < A.n ? flags[i] : 0xFFFF0000u;        // (no rank asks, nobody owns)
    }
    for (u32 s = 0; s < 2 * W; ++s) {
        u32 pos[RF_ITEMS];
        bool p[RF_ITEMS];
#pragma unroll
        for (int r = 0; r < RF_ITEMS; ++r) {
            p[r] = s < W ? ((fl[r] >> s) & 1u) != 0 : (fl[r] >> 16) == s - W;
            const u64 b = __ballot(p[r]);
            pos[r] = (u32)__popcll(b & lanemask_lt());
            if (lane == 0) cw[r][w] = (u32)__popcll(b);
        }
        __syncthreads();
        u64 o = (s < W ? B.keep[s] : B.own[s - W]) + off[(u64)s * A.n_tiles + blockIdx.x];
#pragma unroll
        for (int r = 0; r < RF_ITEMS; ++r) {
            u32 before = 0, total = 0;
#pragma unroll
            for (u32 ww = 0; ww < RF_THREADS / 64; ++ww) { const u32 c = cw[r][ww]; if (ww < w) before += c; total += c; }
            if (p[r]) {
                const u64 d = o + before + pos[r];
                if (s < W) { out_x[d] = xs[r]; if (out_y) out_y[d] = A.y[tile_base + (u64)r * RF_THREADS + threadIdx.x]; }
                else if (out_hash32) out_hash32[d] = (u32)(xs[r] >> A.kshift);
                else out_hash[d] = xs[r] >> A.kshift;
            }
            o += total;
        }
        __syncthreads();
    }
}
