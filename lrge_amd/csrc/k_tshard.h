// k_tshard.h -- the forward strategy over several GPUs with the TARGETS sharded (lrge_hip_index_build_tsharded; round 4).
//
// twoset.rs:286-317 counts, per query, the distinct targets it overlaps; aligner.rs:111-120 builds ONE index over all targets.
// Rounds 2-3 sharded the QUERIES and gave every rank an index restricted to its queries' keys (k_restrict.h, k_route.h): exact,
// but every kept entry, every key set and every minimizer hash crosses the links, and at H. sapiens-scale HiFi (2^30.5 possible
// HPC 19-mers against 7.5 G minimizers: every key recurs all over the genome) a rank's "restricted" index still holds a tenth of
// all entries -- 3.3x at 8 ranks (profiles/r04_emulated_world8_c5_half_fwd.json).  Here rank r indexes ITS contiguous share of the
// target reads and maps ALL queries against it: the shards hold disjoint targets, so a query's distinct-target count is the sum
// of its counts over the ranks (the argument of the partitioned index, host_index_parts.inl) -- ONE all-reduce of u32[Q] closes the job.
// What must be global is mm_idx_cal_max_occ's statistic and the mid_occ filter: a key is dropped by its occurrence count over
// ALL targets.  Each rank therefore sends, per distinct key of its table, (hash, local count) to the rank that owns the hash
// (ONE 8-byte word per distinct key -- hash << 24 | count -- instead of 8-16 bytes per minimizer: the table has done the run-length counting already); the owner adds
// the counts up, the all-reduce of k_restrict.h's statistics vector makes n_keys / n_minimizers / mid_occ those of the one index,
// and the (few: mid_occ_frac = 2e-4 of the keys) too-frequent keys travel back to everybody, who lifts them above the threshold in
// its own table exactly as k_part_drop does for the parts of a partitioned index.  No index entry ever crosses a link.
#pragma once
#include "internal.h"
#include "k_index.h"
#include "k_restrict.h"

#define TS_THREADS 256
#define TS_ITEMS 8
#define TS_MAX_WORLD 16

__device__ __forceinline__ u32 ts_owner(u64 hash, u32 world) { return (u32)(((ks_mix(hash) >> 32) * (u64)world) >> 32); }

// (key, count) pairs of one table per owner rank: totals (tot[o] += ...)
__global__ __launch_bounds__(TS_THREADS) void k_ts_count(const u64 *__restrict__ ht, u64 n_slots, u32 world, unsigned long long *__restrict__ tot) {
    __shared__ u32 c[TS_MAX_WORLD];
    if (threadIdx.x < TS_MAX_WORLD) c[threadIdx.x] = 0;
    __syncthreads();
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += stride) {
        const u64 key = ht[2 * s];
        if (key != HT_EMPTY) atomicAdd(&c[ts_owner(key, world)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < world && c[threadIdx.x]) atomicAdd(&tot[threadIdx.x], (unsigned long long)c[threadIdx.x]);
}

// the pairs, grouped by owner: a block counts its tile per owner in LDS, reserves one run per owner (cursor[o] starts at the
// owner's offset in the send buffers) and fills it -- the order inside an owner's share does not matter (the owner sorts)
// a pair is ONE word: hash << 24 | local count (2k <= 38 bits of hash, and the table caps a count at 2^24 - 1)
#define TS_CNT_BITS HT_CNT_BITS
__global__ __launch_bounds__(TS_THREADS) void k_ts_emit(const u64 *__restrict__ ht, u64 n_slots, u32 world, unsigned long long *__restrict__ cursor,
                                                       u64 *__restrict__ out) {
    __shared__ u32 c[TS_MAX_WORLD], fill[TS_MAX_WORLD];
    __shared__ unsigned long long base[TS_MAX_WORLD];
    if (threadIdx.x < TS_MAX_WORLD) { c[threadIdx.x] = 0; fill[threadIdx.x] = 0; }
    __syncthreads();
    const u64 s0 = (u64)blockIdx.x * (TS_THREADS * TS_ITEMS);
    u64 key[TS_ITEMS]; u32 cnt[TS_ITEMS], own[TS_ITEMS];
#pragma unroll
    for (int r = 0; r < TS_ITEMS; ++r) {
        const u64 s = s0 + (u64)r * TS_THREADS + threadIdx.x;
        key[r] = HT_EMPTY; cnt[r] = 0; own[r] = 0;
        if (s < n_slots) {
            const ulonglong2 e = *(const ulonglong2 *)(ht + 2 * s);
            key[r] = e.x;
            if (e.x != HT_EMPTY) { cnt[r] = ht_count(e.y); own[r] = ts_owner(e.x, world); atomicAdd(&c[own[r]], 1u); }
        }
    }
    __syncthreads();
    if (threadIdx.x < world && c[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], (unsigned long long)c[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TS_ITEMS; ++r)
        if (key[r] != HT_EMPTY) {
            const unsigned long long d = base[own[r]] + atomicAdd(&fill[own[r]], 1u);
            out[d] = key[r] << TS_CNT_BITS | cnt[r];
        }
}

// owner side: the received pairs sorted by hash, run r = [start[r], start[r + 1]): the key's count over all ranks (and parts),
// the occurrence histogram (mm_idx_cal_max_occ's input), the number of minimizers
__global__ __launch_bounds__(256) void k_ts_reduce(const u64 *__restrict__ pairs /* hash << 24 | count, sorted by hash */, const u32 *__restrict__ start,
                                                   const u32 *__restrict__ d_n_runs, u64 n, u32 *__restrict__ gcnt, u32 *__restrict__ hist, u32 max_bin,
                                                   unsigned long long *__restrict__ n_mz) {
    __shared__ u32 lh[OH_BINS];
    __shared__ unsigned long long lsum;
    for (u32 i = threadIdx.x; i < OH_BINS; i += blockDim.x) lh[i] = 0;
    if (threadIdx.x == 0) lsum = 0;
    __syncthreads();
    const u32 n_runs = *d_n_runs;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n_runs; r += stride) {
        const u64 en = (r + 1 < n_runs) ? start[r + 1] : n;
        u64 sum = 0;
        for (u64 i = start[r]; i < en; ++i) sum += pairs[i] & HT_CNT_MAX;
        mine += sum;
        const u32 g = sum > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)sum;
        gcnt[r] = g;
        const u32 hb = g < max_bin ? g : max_bin;
        if (hb < OH_BINS) atomicAdd(&lh[hb], 1u); else atomicAdd(&hist[hb], 1u);
    }
    atomicAdd(&lsum, mine);
    __syncthreads();
    for (u32 i = threadIdx.x; i < OH_BINS && i <= max_bin; i += blockDim.x) if (lh[i]) atomicAdd(&hist[i], lh[i]);
    if (threadIdx.x == 0 && lsum) atomicAdd(n_mz, lsum);
}

// owner side: the keys whose count over all targets exceeds mid_occ (list[] has room for cap of them; *n counts all)
__global__ __launch_bounds__(256) void k_ts_frequent(const u64 *__restrict__ keys, const u32 *__restrict__ start, const u32 *__restrict__ d_n_runs,
                                                     const u32 *__restrict__ gcnt, u32 mid_occ, u64 *__restrict__ list, u32 cap, u32 *__restrict__ n) {
    const u32 n_runs = *d_n_runs;
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_runs || gcnt[r] <= mid_occ) return;
    const u32 d = atomicAdd(n, 1u);
    if (d < cap) list[d] = keys[start[r]] >> TS_CNT_BITS;
}

// every rank: a key that is too frequent over ALL targets is lifted above the threshold in this table (k_part_drop's marking:
// all k_lookup's consumers test is count > mid_occ; an inline singleton becomes an ordinary entry whose list is never expanded)
__global__ __launch_bounds__(256) void k_ts_mark(u64 *__restrict__ ht, u64 cap, u32 fix, const u64 *__restrict__ keys, const u32 *__restrict__ n_of_rank,
                                                 u32 world, u32 stride /* keys of rank r at keys[r * stride ...) */, u32 mid_occ) {
    const u32 r = blockIdx.y;
    if (r >= world) return;
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_of_rank[r]) return;
    const u64 key = keys[(u64)r * stride + i];
    u64 slot = ht_home(key, cap, fix);
    const u64 bk = __builtin_bswap64(key);
    for (;; ++slot) {
        const ulonglong2 e = *(const ulonglong2 *)(ht + 2 * slot);
        if (__builtin_bswap64(e.x) >= bk) {
            if (e.x == key && ht_count(e.y) <= mid_occ) ht[2 * slot + 1] = ((e.y & HT_INLINE) ? 0ULL : (e.y & ~(u64)HT_CNT_MAX)) | (u64)(mid_occ + 1);
            return;
        }
    }
}
