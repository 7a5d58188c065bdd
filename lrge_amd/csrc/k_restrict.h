// k_restrict.h -- the index of a multi-GPU run: restricted to the minimizers one rank's streamed reads carry, with
// occurrence statistics that are still those of the WHOLE target set.
//
// mm_idx_gen indexes every target minimizer and mm_idx_cal_max_occ takes mid_occ from all distinct keys (mm2:index.c;
// aligner.rs:144-197).  A rank that maps only its own range of the streamed reads asks mm_idx_get about the keys THOSE
// reads carry and about no others, so its index only has to hold the entries of those keys -- complete lists, hence the
// same answers -- provided mid_occ stays global.  Per rank:
//   * a key set of the streamed reads' minimizers: a direct bitmap over the 2k-bit hash space when that fits (k = 15:
//     2^30 bits = 128 MB, no false positives), else a blocked Bloom filter (3 bits inside one 64-bit word; a false positive
//     only keeps entries nobody asks for);
//   * one order-preserving pass over the target entries (two sweeps: count per tile, scan, write) that keeps the entries
//     whose key is in the set, and beside them emits the bare hashes of the keys this rank OWNS (a 1/world share of the
//     hash space) for the statistics.  It runs between the first and the second LSD pass of the index sort: the first
//     pass groups the entries by the top digit of the hash and a key's bit lives in the slice of the set that belongs
//     to its top digit, so a group's tests stay inside L2; the remaining passes only move what was kept;
//   * the owned hashes are sorted and run-length counted into the occurrence histogram; one all-reduce (comm.h) of
//     [distinct keys, minimizers, histogram] makes it the histogram of the whole target set, and mid_occ follows with the
//     reference's arithmetic.
#pragma once
#include "internal.h"
#include "k_prims.h"
#include "k_sketch.h"

struct KeySet {
    u64 *bits;          // n_words 64-bit words
    u64 word_mask;      // n_words - 1 (power of two)
    int direct;         // 1: bit index = hash itself (n_words * 64 >= 2^(2k))
    u32 top_shift;      // hash >> top_shift = the top digit of the hash (the first LSD pass of the index sort groups by it)
    u32 low_bits;       // Bloom form: word = top digit << low_bits | low_bits bits of a mix of the hash
};

__device__ __forceinline__ u64 ks_mix(u64 h) { h *= 0x9E3779B97F4A7C15ULL; return h ^ (h >> 29); }

__device__ __forceinline__ void ks_locate(const KeySet &ks, u64 hash, u64 *word, u64 *mask) {
    if (ks.direct) { *word = hash >> 6; *mask = 1ULL << (hash & 63); return; }
    const u64 m = ks_mix(hash);
    *word = ((hash >> ks.top_shift) << ks.low_bits | ((m >> 20) & ((1ULL << ks.low_bits) - 1))) & ks.word_mask;
    *mask = 1ULL << (m & 63) | 1ULL << ((m >> 6) & 63) | 1ULL << ((m >> 12) & 63);
}
__device__ __forceinline__ bool ks_test(const KeySet &ks, u64 hash) {
    u64 w, m;
    ks_locate(ks, hash, &w, &m);
    return (ks.bits[w] & m) == m;
}

// one lane per streamed minimizer (x = hash << 8 | span)
__global__ __launch_bounds__(256) void k_keyset_build(const u64 *__restrict__ qx, const u32 *__restrict__ d_n, KeySet ks) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *d_n) return;
    u64 w, m;
    ks_locate(ks, qx[i] >> 8, &w, &m);
    atomicOr((unsigned long long *)&ks.bits[w], (unsigned long long)m);
}

// which rank owns a hash for the occurrence statistics
__device__ __forceinline__ bool own_hash(u64 hash, u32 rank, u32 world) {
    return world <= 1 || (u32)(((ks_mix(hash) >> 32) * (u64)world) >> 32) == rank;    // (multiply-shift: no integer division)
}

// The filter inside the sketch (the one-pass form of k_sketch.h): a minimizer is tested the moment mm_sketch's loop emits
// it, so the entries nobody will ask for never reach memory -- the kernel is bound by its hash / window arithmetic, and the
// scattered key-set probes ride along under it.  Kept entries go to the chunk's slot of tmp_x (tmp_y: the (hash, y) pair
// layout), the bare hashes this rank owns to its slot of tmp_h; k_sketch_compact closes the gaps of each stream.
template <int K, int W, bool HPC, bool PK>
__global__ __launch_bounds__(SK_THREADS) void k_sketch_restrict(const u64 *__restrict__ pack, const u32 *__restrict__ nmask,
                                                               const u64 *__restrict__ woff, const u32 *__restrict__ lens,
                                                               ChunkMap cm, u32 n_chunks, u32 *__restrict__ cnt_keep, u32 *__restrict__ cnt_own,
                                                               u32 *__restrict__ overflow, u64 *__restrict__ tmp_x, u64 *__restrict__ tmp_y,
                                                               u64 *__restrict__ tmp_h, u32 pk_pos1, u32 pk_ybits, u32 cap, KeySet ks,
                                                               u32 rank, u32 world) {
    const u32 c = blockIdx.x * SK_THREADS + threadIdx.x;
    if (c >= n_chunks) return;
    const u32 r = cm.find(c);
    const i32 len = (i32)lens[r];
    const i32 s = (i32)(c - cm.chunk_start[r]) * SK_CHUNK;
    const i32 e = s + SK_CHUNK < len ? s + SK_CHUNK : len;
    const u64 base = (u64)c * SK_CAP;
    u32 nk = 0, no = 0;
    sketch_chunk<K, W, HPC>(pack, nmask, woff[r], len, r, s, e, [&](u64 x, u64 y) {
        const u64 h = x >> 8;
        if (ks_test(ks, h)) {
            if (nk < cap) {
                if (PK) tmp_x[base + nk] = h << pk_ybits | (y >> 32) << pk_pos1 | (u64)(u32)y;
                else { tmp_x[base + nk] = h; tmp_y[base + nk] = y; }
            }
            ++nk;
        }
        if (own_hash(h, rank, world)) { if (no < cap) tmp_h[base + no] = h; ++no; }
    });
    cnt_keep[c] = nk; cnt_own[c] = no;
    if (nk > cap || no > cap) *overflow = 1u;
}

#define RF_THREADS 256
#define RF_ITEMS 8
#define RF_TILE (RF_THREADS * RF_ITEMS)

struct RestrictArgs {
    const u64 *x; const u64 *y;     // entries: packed (y null, hash = x >> kshift) or (hash, y) pairs
    u64 n; u32 kshift;
    KeySet ks;
    u32 rank, world;
};

// A tile = RF_ITEMS rows of RF_THREADS consecutive entries: lane t of a row reads entry tile_base + row * RF_THREADS + t,
// so every load is one contiguous 512-byte run per wavefront, and (row, wave, lane) order is memory order -- the
// compaction stays order-preserving with per-(row, wave) ballot counts instead of per-thread runs.
// flags: bit row = keep, bit 16 + row = owned
__device__ __forceinline__ u32 rf_flags(const RestrictArgs &A, u64 tile_base, u64 *xs) {
    u32 f = 0;
#pragma unroll
    for (int r = 0; r < RF_ITEMS; ++r) {
        const u64 i = tile_base + (u64)r * RF_THREADS + threadIdx.x;
        xs[r] = i < A.n ? A.x[i] : 0;
    }
#pragma unroll
    for (int r = 0; r < RF_ITEMS; ++r) {
        const u64 i = tile_base + (u64)r * RF_THREADS + threadIdx.x;
        if (i < A.n) {
            const u64 h = xs[r] >> A.kshift;
            if (ks_test(A.ks, h)) f |= 1u << r;
            if (own_hash(h, A.rank, A.world)) f |= 1u << (16 + r);
        }
    }
    return f;
}

__global__ __launch_bounds__(RF_THREADS) void k_restrict_count(RestrictArgs A, u32 *__restrict__ bc_keep, u32 *__restrict__ bc_own,
                                                               u32 *__restrict__ flags) {
    __shared__ u32 wk[RF_THREADS / 64], wo[RF_THREADS / 64];
    u64 xs[RF_ITEMS];
    const u32 f = rf_flags(A, (u64)blockIdx.x * RF_TILE, xs);
    flags[(u64)blockIdx.x * RF_THREADS + threadIdx.x] = f;      // the write sweep does not probe the key set again
    u32 ck = (u32)__popc(f & 0xffffu), co = (u32)__popc(f >> 16);
    for (int d = 32; d > 0; d >>= 1) { ck += __shfl_down(ck, d, 64); co += __shfl_down(co, d, 64); }
    if (lane_id() == 0) { wk[threadIdx.x >> 6] = ck; wo[threadIdx.x >> 6] = co; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 a = 0, b = 0;
        for (int w = 0; w < RF_THREADS / 64; ++w) { a += wk[w]; b += wo[w]; }
        bc_keep[blockIdx.x] = a; bc_own[blockIdx.x] = b;
    }
}

__global__ __launch_bounds__(RF_THREADS) void k_restrict_write(RestrictArgs A, const u32 *__restrict__ off_keep, const u32 *__restrict__ off_own,
                                                               const u32 *__restrict__ flags, u64 *__restrict__ out_x, u64 *__restrict__ out_y,
                                                               u64 *__restrict__ out_hash) {
    __shared__ u32 ck[RF_ITEMS][RF_THREADS / 64], co[RF_ITEMS][RF_THREADS / 64];
    const u64 tile_base = (u64)blockIdx.x * RF_TILE;
    u64 xs[RF_ITEMS];
    const u32 f = flags[(u64)blockIdx.x * RF_THREADS + threadIdx.x];
#pragma unroll
    for (int r = 0; r < RF_ITEMS; ++r) {
        const u64 i = tile_base + (u64)r * RF_THREADS + threadIdx.x;
        xs[r] = i < A.n ? A.x[i] : 0;
    }
    const u32 w = threadIdx.x >> 6, lane = lane_id();
    u32 pk[RF_ITEMS], po[RF_ITEMS];          // my rank inside my (row, wave)
#pragma unroll
    for (int r = 0; r < RF_ITEMS; ++r) {
        const u64 bk = __ballot((f >> r) & 1u), bo = __ballot((f >> (16 + r)) & 1u);
        pk[r] = (u32)__popcll(bk & lanemask_lt()); po[r] = (u32)__popcll(bo & lanemask_lt());
        if (lane == 0) { ck[r][w] = (u32)__popcll(bk); co[r][w] = (u32)__popcll(bo); }
    }
    __syncthreads();
    u32 ok = off_keep[blockIdx.x], oo = off_own[blockIdx.x];
#pragma unroll
    for (int r = 0; r < RF_ITEMS; ++r) {
        u32 bk = 0, bo = 0, tk = 0, to = 0;   // before my wave in this row / the whole row
#pragma unroll
        for (u32 ww = 0; ww < RF_THREADS / 64; ++ww) {
            const u32 a = ck[r][ww], b = co[r][ww];
            if (ww < w) { bk += a; bo += b; }
            tk += a; to += b;
        }
        if ((f >> r) & 1u) {
            const u32 d = ok + bk + pk[r];
            out_x[d] = xs[r];
            if (out_y) out_y[d] = A.y[tile_base + (u64)r * RF_THREADS + threadIdx.x];
        }
        if ((f >> (16 + r)) & 1u) out_hash[oo + bo + po[r]] = xs[r] >> A.kshift;
        ok += tk; oo += to;
    }
}

// occurrence histogram of the runs of a sorted hash stream (run r = [start[r], start[r + 1])), gathered in LDS and
// flushed once per block from a small grid (k_place_apply's lesson); *d_n_runs lives on the device
#define OH_BINS 2048
__global__ __launch_bounds__(256) void k_occ_hist_runs(const u32 *__restrict__ start, const u32 *__restrict__ d_n_runs, u64 n,
                                                       u32 *__restrict__ hist, u32 max_bin) {
    __shared__ u32 lh[OH_BINS];
    for (u32 i = threadIdx.x; i < OH_BINS; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const u32 n_runs = *d_n_runs;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 r0 = (u64)blockIdx.x * blockDim.x; r0 < n_runs; r0 += stride) {
        const u64 r = r0 + threadIdx.x;
        const bool in = r < n_runs;
        u32 hb = 0;
        if (in) {
            const u64 en = (r + 1 < n_runs) ? start[r + 1] : n;
            const u64 c = en - start[r];
            hb = c < max_bin ? (u32)c : max_bin;
        }
        // most runs have length 1 or 2: count those per wave with a ballot instead of 64 conflicting LDS atomics
        const u64 m1 = __ballot(in && hb == 1), m2 = __ballot(in && hb == 2);
        if (lane_id() == 0) {
            if (m1) atomicAdd(&lh[1], (u32)__popcll(m1));
            if (m2) atomicAdd(&lh[2], (u32)__popcll(m2));
        }
        if (in && hb != 1 && hb != 2) { if (hb < OH_BINS) atomicAdd(&lh[hb], 1u); else atomicAdd(&hist[hb], 1u); }
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < OH_BINS && i <= max_bin; i += blockDim.x) if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// [0] = distinct keys, [1] = minimizers, [2 ...) = the first `head` histogram bins, as u64 for the all-reduce
__global__ void k_stats_pack(const u32 *__restrict__ d_n_runs, u64 n_mz, const u32 *__restrict__ hist, u32 head, u64 *__restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { out[0] = *d_n_runs; out[1] = n_mz; }
    if (i < head) out[2 + i] = hist[i];
}
__global__ void k_u32_to_u64(const u32 *__restrict__ in, u64 n, u64 *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}
