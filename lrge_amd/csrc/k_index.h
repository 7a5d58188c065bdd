// k_index.h -- K2 index build on top of the sorted (hash, y) stream, K2b occurrence threshold,
// K3 lookup.  Restates mm2:index.c worker_post / mm_idx_get / mm_idx_cal_max_occ as:
//   sorted keys -> run heads -> open-addressing table  hash -> (start, count)  into pos[].
// (A bucket directory over the sorted distinct keys was tried instead of the table: minimizer hashes
// are window minima, i.e. heavily skewed toward small values, which starves the top buckets and
// crowds the bottom ones; the hash table is indifferent to the key distribution.)
// The position list of a key is the y values in ascending order (the reference re-sorts every list
// by y, and the sketch stream is already ascending in y, so a STABLE key sort yields that order).
#pragma once
#include "internal.h"
#include "k_prims.h"

#define HT_EMPTY (~0ULL)
#define HT_CNT_BITS 24
#define HT_CNT_MAX ((1u << HT_CNT_BITS) - 1)

__device__ __forceinline__ u64 ht_slot_hash(u64 key) {
    // keys are already outputs of an invertible mixer; one multiply spreads the low bits
    return (key * 0x9E3779B97F4A7C15ULL) >> 20;
}

__global__ void k_run_heads(const u64 *__restrict__ skey, u64 n, u32 *__restrict__ head) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || skey[i] != skey[i - 1]) ? 1u : 0u;
}

// run_start[run_id] = i for every head (run_id from the exclusive scan of head flags)
__global__ void k_run_starts(const u32 *__restrict__ head, const u32 *__restrict__ run_id, u64 n,
                             u32 *__restrict__ run_start) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (head[i]) run_start[run_id[i]] = (u32)i;
}

#define OCC_LDS_BINS 2048
// per-run: insert into the table, histogram the run length (clamped to max_bin)
// table entry = {key, start<<24 | count} in one 16-byte slot: insert and lookup touch one line
__global__ __launch_bounds__(256) void k_table_insert(const u64 *__restrict__ skey, const u32 *__restrict__ run_start,
                                                      u32 n_runs, u64 n, u64 *__restrict__ ht, u64 ht_mask,
                                                      u32 *__restrict__ occ_hist, u32 max_bin) {
    __shared__ u32 lh[OCC_LDS_BINS];
    for (u32 i = threadIdx.x; i < OCC_LDS_BINS; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    // grid-stride with a small grid: the histogram flush below hits the same few global addresses from
    // every block, so the number of blocks (not of runs) sets that serialised cost
    for (u64 rr = (u64)blockIdx.x * blockDim.x + threadIdx.x; rr < n_runs; rr += (u64)gridDim.x * blockDim.x) {
        const u32 r = (u32)rr;
        u32 st = run_start[r];
        u64 en = (r + 1 < n_runs) ? run_start[r + 1] : n;
        u32 cnt = (u32)(en - st);
        u64 key = skey[st];
        u64 slot = ht_slot_hash(key) & ht_mask;
        for (;;) {
            u64 prev = atomicCAS((unsigned long long *)&ht[2 * slot], (unsigned long long)HT_EMPTY, (unsigned long long)key);
            if (prev == HT_EMPTY) break;  // keys are distinct per run, so no "already present" case
            slot = (slot + 1) & ht_mask;
        }
        ht[2 * slot + 1] = (u64)st << HT_CNT_BITS | (cnt < HT_CNT_MAX ? cnt : HT_CNT_MAX);
        const u32 hb = cnt < max_bin ? cnt : max_bin;
        // most runs have length 1 or 2: count those per wave with a ballot instead of 64 conflicting
        // LDS atomics on one address
        const u64 m1 = __ballot(hb == 1), m2 = __ballot(hb == 2);
        const u32 leader = (u32)__ffsll((unsigned long long)__ballot(true)) - 1;
        if (lane_id() == leader) {
            if (m1) atomicAdd(&lh[1], (u32)__popcll(m1));
            if (m2) atomicAdd(&lh[2], (u32)__popcll(m2));
        }
        if (hb > 2) { if (hb < OCC_LDS_BINS) atomicAdd(&lh[hb], 1u); else atomicAdd(&occ_hist[hb], 1u); }
        else if (hb == 0) atomicAdd(&lh[0], 1u);
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < OCC_LDS_BINS && i <= max_bin; i += blockDim.x)
        if (lh[i]) atomicAdd(&occ_hist[i], lh[i]);
}

__device__ __forceinline__ bool ht_lookup(const u64 *__restrict__ ht, u64 ht_mask, u64 key, u64 *start, u32 *cnt) {
    u64 slot = ht_slot_hash(key) & ht_mask;
    for (;;) {
        const ulonglong2 e = *(const ulonglong2 *)(ht + 2 * slot);
        if (e.x == key) { *start = e.y >> HT_CNT_BITS; *cnt = (u32)(e.y & HT_CNT_MAX); return true; }
        if (e.x == HT_EMPTY) return false;
        slot = (slot + 1) & ht_mask;
    }
}
