// k_index.h -- K2 index build on top of the sorted (hash, y) stream, K2b occurrence threshold,
// K3 lookup.  Restates mm2:index.c worker_post / mm_idx_get / mm_idx_cal_max_occ as:
//   sorted keys -> run heads -> ORDERED open-addressing table  hash -> (start, count)  into pos[].
// The minimizer stream is sorted by the BYTE-REVERSED hash (low byte most significant): minimizer
// hashes are window minima, heavily skewed toward small values, but their low bytes are uniform.  A
// key's home slot is  umulhi(bswap64(key), cap) (with the partial top byte stretched: ht_home), monotone in that order, so inserting the distinct
// keys in sorted order with linear probing has a closed form -- slot(r) = max(home(r), slot(r-1) + 1)
// = r + prefix-max(home(r') - r') -- i.e. one max-scan and one streaming pass with nearly sequential
// writes; no atomics, no random CAS traffic.  Lookups probe linearly from the home slot as usual (the
// table does not wrap: it has slack slots behind `cap`, and the last one always stays empty).
// The position list of a key is the y values in ascending order (the reference re-sorts every list
// by y, and the sketch stream is already ascending in y, so a STABLE key sort yields that order).
#pragma once
#include "internal.h"
#include "k_prims.h"

#define HT_EMPTY (~0ULL)
#define HT_CNT_BITS 24
#define HT_CNT_MAX ((1u << HT_CNT_BITS) - 1)
// A key that occurs ONCE -- most keys of a noisy read set: 242 M entries over ~190 M distinct keys at C4 -- carries its one
// position in the slot itself: value = HT_INLINE | y (y = rid << 32 | pos << 1 | strand < 2^63).  The expansion then never
// touches pos[] for it; before, every such key cost one random 64-byte sector for 8 useful bytes (minimap2 itself keeps
// singletons inside its khash: mm2:index.c mm_idx_get).  Every other key: value = start << 24 | min(count, 2^24 - 1).
#define HT_INLINE (1ULL << 63)
__device__ __forceinline__ u32 ht_count(u64 v) { return (v & HT_INLINE) ? 1u : (u32)(v & HT_CNT_MAX); }

// Home slot = position of the key in the byte-reversed order, scaled to [0, cap).  A hash of B = 2k bits fills its top byte only
// partly (k = 15: 6 of 8 bits; k = 19 likewise), and byte-reversed that byte sits in the MIDDLE of the 64-bit value: left as it
// is, the keys of every (b0, b1, b2) prefix would all fall into the first quarter of that prefix's slot range -- runs of ~11
// occupied slots followed by ~11 empty ones at load 1/2, ~5 slots and ~1.8 sectors per probe instead of ~1.5 and ~1.15
// (measured: 117 bytes fetched per probe; tools/micro/random_access.hip reaches 52 G single-sector probes per second on the same
// 6 GB).  `fix` = position of that byte in the reversed value | its missing bits << 8 (ht_fix_of): the partial byte is
// stretched to a full one, which keeps the order (it is the least significant digit of it) and makes the homes uniform.
__host__ __device__ __forceinline__ u32 ht_fix_of(int k) {
    const int B = 2 * k, part = B % 8;
    if (part == 0) return 0;
    const int nb = (B + 7) / 8;                       // bytes of the hash; the partial one is byte nb - 1 -> bits [64 - 8 nb, +8)
    return (u32)(64 - 8 * nb) | (u32)(8 - part) << 8;
}
// + the exponent of the distribution correction of the partial byte (0: linear stretch only)
// pw is clamped to what ht_home's arithmetic holds: u^pw must span at least the 8 output bits and fit 56 (bits * pw in
// [8, 56]); a value outside -- option HT_POWER is the caller's -- would shift by a negative amount or overflow and make the
// map non-monotone, i.e. an ordered table whose lookups miss present keys
__host__ __device__ __forceinline__ u32 ht_fix_with_power(int k, u32 pw) {
    const u32 f = ht_fix_of(k);
    if (!f) return 0u;
    const u32 bits = 8 - ((f >> 8) & 0xff);
    if (pw) { const u32 lo = (8 + bits - 1) / bits, hi = 56 / bits; pw = pw < lo ? lo : pw > hi ? hi : pw; }
    return f | pw << 16;
}
__device__ __forceinline__ u64 ht_home(u64 key, u64 cap, u32 fix) {
    u64 bs = __builtin_bswap64(key);
    const u32 pos = fix & 0xff, tsh = (fix >> 8) & 0xff, pw = fix >> 16;
    const u64 pb = bs & (0xFFull << pos);
    u64 nb = pb << tsh;                                // (tsh == 0: unchanged)
    if (pw) {
        // The partial byte holds the TOP bits of the hash, and minimizer hashes are window minima: P(h > x) ~ (1 - x)^w, three
        // quarters of them lie in the first quarter of the range.  Any monotone map of that byte keeps the table's order, so it
        // is sent through the distribution function 1 - (1 - t)^pw, which spreads the keys of a stretch evenly over it.
        const u32 bits = 8 - tsh, v = (u32)(pb >> pos);                // v in [0, 2^bits)
        const u32 u = (1u << bits) - v;                                  // 1 .. 2^bits
        u64 p = u;
        for (u32 e = 1; e < pw; ++e) p *= u;                             // u^pw <= 2^(bits * pw) <= 2^48
        const u32 sh = bits * pw - 8;
        u32 f = 256u - (u32)(p >> sh);                                   // 256 * (1 - (u / 2^bits)^pw), 0 .. 256
        f = f > 255u ? 255u : f;
        nb = (u64)f << pos;
    }
    bs = (bs ^ pb) | nb;
    return __umul64hi(bs, cap);
}

// (run heads of the sorted stream: compact_heads() in k_prims.h; kshift = bits below the hash in a packed entry)

#define OCC_LDS_BINS 2048
#define PLACE_THREADS 256
#define PLACE_ROWS 8
#define PLACE_TILE (PLACE_THREADS * PLACE_ROWS)
#define PLACE_COMP 384      // slots a wavefront composes in LDS (64 runs at load 1/2 span ~128)

// The hash of entry `st` of the sorted stream.  seg.start != null: a segment-packed index (k_prims.h: index_sort_segpacked).
// The stream is ordered by the byte-reversed hash; written as ONE number, that order is the hash's SIGNIFICANCE STRING
//     S = b0 . b1 . b2 ... b(m-1)          (b0 = low byte, most significant; the top byte b(m-1) holds tb = nbits - 8 (m - 1) bits)
// (hash_to_sig: a byte swap with the partial top byte squeezed).  A segment-packed entry keeps the low nr = nbits - 8 - e bits R of
// S -- the top 8 + e bits are the NUMBER of the segment it lies in (256 << e segments) -- so its hash field compares like the order
// itself and the LSD passes of the sort may cut R wherever they like (round 5; rounds 3-4 stored the hash's own bytes, which tied
// the digits to byte boundaries: k = 19 with e = 2 took four keys-only passes over 6 + 8 + 8 + 6 bits, now e = 6 leaves 24 = 3 x 8).
struct SegStarts { const u32 *start; u32 e, nbits; };
// (hash_to_sig / sig_to_hash: k_prims.h, beside the sort that packs the entries)
__host__ __device__ __forceinline__ u64 seg_hash(u64 R, u32 sgm, u32 e, u32 nbits) {        // R = the stored bits, sgm = the segment's number
    return sig_to_hash((u64)sgm << (nbits - 8 - e) | R, nbits);
}
// the last segment whose start is <= st (empty segments share their start with the next)
__device__ __forceinline__ u32 seg_of(SegStarts seg, u32 st) {
    const u32 n_seg = 256u << seg.e;
    u32 lo = 0, hi = n_seg;
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (seg.start[mid] <= st) lo = mid; else hi = mid; }
    while (lo + 1 < n_seg && seg.start[lo + 1] <= st) ++lo;
    return lo;
}
// hint: a segment at or in front of the entry's (the callers walk runs in ascending order: the segment of a block's first run, found
// once, is that of nearly all of them -- 10 dependent loads per run otherwise)
__device__ __forceinline__ u64 entry_hash(const u64 *__restrict__ skey, u32 st, u32 kshift, SegStarts seg, u32 hint = 0xFFFFFFFFu) {
    const u64 h = skey[st] >> kshift;
    if (!seg.start) return h;
    u32 lo;
    if (hint == 0xFFFFFFFFu) lo = seg_of(seg, st);
    else { const u32 n_seg = 256u << seg.e; lo = hint; while (lo + 1 < n_seg && seg.start[lo + 1] <= st) ++lo; }
    return seg_hash(h, lo, seg.e, seg.nbits);
}

// d(r) = home(r) + (n_runs - r): slot(r) = prefix-max(d)(r) - (n_runs - r).  Needs cap + n_runs < 2^32.
__device__ __forceinline__ u32 place_d(const u64 *__restrict__ skey, const u32 *__restrict__ run_start, u32 r, u32 n_runs, u64 cap, u32 kshift, u32 fix,
                                       SegStarts seg_start, u32 hint = 0xFFFFFFFFu) {
    return (u32)ht_home(entry_hash(skey, run_start[r], kshift, seg_start, hint), cap, fix) + (n_runs - r);
}

__global__ __launch_bounds__(PLACE_THREADS) void k_place_reduce(const u64 *__restrict__ skey, const u32 *__restrict__ run_start,
                                                                u32 n_runs, u64 cap, u32 *__restrict__ bmax, u32 kshift, u32 fix,
                                                                SegStarts seg_start) {
    __shared__ u32 wm[PLACE_THREADS / 64];
    __shared__ u32 seg0;
    u32 m = 0;
    const u32 base = blockIdx.x * PLACE_TILE;
    if (threadIdx.x == 0) seg0 = seg_start.start && base < n_runs ? seg_of(seg_start, run_start[base]) : 0xFFFFFFFFu;
    __syncthreads();
    const u32 hint = seg0;
#pragma unroll
    for (int i = 0; i < PLACE_ROWS; ++i) {
        const u32 r = base + i * PLACE_THREADS + threadIdx.x;
        if (r < n_runs) { const u32 d = place_d(skey, run_start, r, n_runs, cap, kshift, fix, seg_start, hint); m = d > m ? d : m; }
    }
    for (int d = 32; d > 0; d >>= 1) { const u32 o = (u32)__shfl_xor((i32)m, d, 64); m = o > m ? o : m; }
    if (lane_id() == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 t = 0;
        for (int w = 0; w < PLACE_THREADS / 64; ++w) t = wm[w] > t ? wm[w] : t;
        bmax[blockIdx.x] = t;
    }
}

// single block: in-place EXCLUSIVE max-scan of the block maxima (identity 0)
__global__ __launch_bounds__(1024) void k_place_scan(u32 *data, u32 n) {
    __shared__ u32 wm[16];
    __shared__ u32 carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (u32 base = 0; base < n; base += 1024) {
        const u32 i = base + threadIdx.x;
        const u32 v = i < n ? data[i] : 0;
        u32 inc = v;
        for (int d = 1; d < 64; d <<= 1) { const u32 o = (u32)__shfl_up((i32)inc, d, 64); if ((int)lane_id() >= d) inc = o > inc ? o : inc; }
        if (lane_id() == 63) wm[threadIdx.x >> 6] = inc;
        __syncthreads();
        u32 pre = carry_s;
        for (u32 w = 0; w < (threadIdx.x >> 6); ++w) pre = wm[w] > pre ? wm[w] : pre;
        u32 exc = (u32)__shfl_up((i32)inc, 1, 64);
        exc = lane_id() == 0 ? 0 : exc;
        if (i < n) data[i] = exc > pre ? exc : pre;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = inc > pre ? inc : pre;
        __syncthreads();
    }
}

// per run: its slot from the max-scan, the table entry {key, start<<24 | count} (one 16-byte slot: insert
// and lookup touch one line), and the histogram of run lengths (clamped to max_bin) for mm_idx_cal_max_occ
__global__ __launch_bounds__(PLACE_THREADS) void k_place_apply(const u64 *__restrict__ skey, const u32 *__restrict__ run_start,
                                                               u32 n_runs, u64 n, u64 cap, u64 n_slots, const u32 *__restrict__ bpre,
                                                               u64 *__restrict__ ht, u32 *__restrict__ occ_hist, u32 max_bin,
                                                               u32 *__restrict__ overflow, u32 kshift, u32 fix, u32 *__restrict__ last_slot,
                                                               const u64 *__restrict__ ypos, u32 pk_pos1, u32 inline_single,
                                                               SegStarts seg_start) {
    // ypos: the y values of the (hash, y) pair layout (null: packed entries, y is decoded from the entry itself)
    // last_slot != null: the table has NOT been cleared.  The runs of a wavefront occupy increasing slots, and the slot of the
    // run in front of the wavefront's first one is known from the same max-scan: every wavefront owns the contiguous slot
    // range (slot of the run before its first, slot of its last], composes it in LDS -- empty slots and entries -- and writes
    // it as whole lines.  The table is written once (6 GB at C4) instead of cleared and then written into with 16-byte
    // stores; *last_slot receives the slot of the last run (k_fill_tail clears what lies behind it).
    __shared__ ulonglong2 comp[PLACE_THREADS / 64][PLACE_COMP];
    __shared__ u32 lh[OCC_LDS_BINS];
    __shared__ u32 wm[PLACE_THREADS / 64];
    __shared__ u32 carry_s, seg0_s;
    for (u32 i = threadIdx.x; i < OCC_LDS_BINS; i += blockDim.x) lh[i] = 0;
    const u32 w = threadIdx.x >> 6, lane = lane_id();
    u32 disp_sum = 0;      // (per lane: at most a few thousand runs x small displacements)
    const u32 n_tiles = (n_runs + PLACE_TILE - 1) / PLACE_TILE;
    // grid-stride over tiles with a small grid: the histogram flush at the end hits the same few global
    // addresses from every block, so the number of blocks (not of runs) sets that serialised cost
    for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();
    const u32 base = tile * PLACE_TILE;
    if (threadIdx.x == 0) { carry_s = bpre[tile]; seg0_s = seg_start.start && base < n_runs ? seg_of(seg_start, run_start[base]) : 0xFFFFFFFFu; }
    __syncthreads();
    const u32 hint = seg0_s;
    for (int row = 0; row < PLACE_ROWS; ++row) {
        const u32 r = base + row * PLACE_THREADS + threadIdx.x;
        const bool in = r < n_runs;
        u32 st = 0, cnt = 0; u64 key = 0; u32 d = 0;
        if (in) {
            st = run_start[r];
            const u64 en = (r + 1 < n_runs) ? run_start[r + 1] : n;
            cnt = (u32)(en - st);
            key = entry_hash(skey, st, kshift, seg_start, hint);
            d = (u32)ht_home(key, cap, fix) + (n_runs - r);
        }
        u32 inc = d;
        for (int s = 1; s < 64; s <<= 1) { const u32 o = (u32)__shfl_up((i32)inc, s, 64); if ((int)lane >= s) inc = o > inc ? o : inc; }
        if (lane == 63) wm[w] = inc;
        __syncthreads();
        u32 pre = carry_s;
        for (u32 ww = 0; ww < w; ++ww) pre = wm[ww] > pre ? wm[ww] : pre;
        const u32 m = inc > pre ? inc : pre;
        __syncthreads();
        if (threadIdx.x == PLACE_THREADS - 1) carry_s = m;
        const u64 slot = in ? (u64)m - (n_runs - r) : 0;
        if (in) disp_sum += (u32)(slot - (u64)(d - (n_runs - r)));      // slot - home
        ulonglong2 e; e.x = key; e.y = (u64)st << HT_CNT_BITS | (cnt < HT_CNT_MAX ? cnt : HT_CNT_MAX);
        if (in && cnt == 1 && inline_single) {
            u64 y;
            if (ypos) y = ypos[st];
            else { const u64 yb = skey[st] & ((1ULL << kshift) - 1); y = (yb >> pk_pos1) << 32 | (yb & ((1ULL << pk_pos1) - 1)); }
            e.y = HT_INLINE | y;
        }
        if (in && slot + 1 >= n_slots) *overflow = 1u;      // the last slot must stay empty
        if (!last_slot) {
            if (in && slot + 1 < n_slots) *(ulonglong2 *)(ht + 2 * slot) = e;
        } else {
            // first slot this lane is responsible for: one behind the slot of run r - 1 (m of the lane before; `pre` for lane 0)
            u32 m_prev = (u32)__shfl_up((i32)m, 1, 64);
            m_prev = lane == 0 ? pre : m_prev;
            const u64 fill0 = (in && r > 0) ? (u64)m_prev - (n_runs - (r - 1)) + 1 : 0;
            const u64 vmask = __ballot(in);
            if (vmask) {
                const int nv = (int)__popcll(vmask);                           // valid lanes are a prefix
                const u64 F = (u64)__shfl((i32)(u32)fill0, 0, 64) | (u64)(u32)__shfl((i32)(u32)(fill0 >> 32), 0, 64) << 32;
                const u64 L = (u64)(u32)__shfl((i32)(u32)slot, nv - 1, 64) | (u64)(u32)__shfl((i32)(u32)(slot >> 32), nv - 1, 64) << 32;
                const u64 R = L - F + 1;
                ulonglong2 empty; empty.x = HT_EMPTY; empty.y = HT_EMPTY;
                if (R <= PLACE_COMP) {
                    for (u32 i = lane; i < (u32)R; i += 64) comp[w][i] = empty;
                    if (in) comp[w][(u32)(slot - F)] = e;
                    for (u32 i = lane; i < (u32)R; i += 64) if (F + i < n_slots) *(ulonglong2 *)(ht + 2 * (F + i)) = comp[w][i];
                } else if (in) {                                              // a sparse stretch: every lane clears its own gap
                    for (u64 q = fill0; q < slot; ++q) if (q < n_slots) *(ulonglong2 *)(ht + 2 * q) = empty;
                    if (slot < n_slots) *(ulonglong2 *)(ht + 2 * slot) = e;
                }
                if (in && r == n_runs - 1) *last_slot = (u32)(slot < n_slots ? slot : n_slots - 1);
            }
        }
        const u32 hb = cnt < max_bin ? cnt : max_bin;
        // most runs have length 1 or 2: count those per wave with a ballot instead of 64 conflicting LDS atomics
        const u64 m1 = __ballot(in && hb == 1), m2 = __ballot(in && hb == 2);
        if (lane == 0) {
            if (m1) atomicAdd(&lh[1], (u32)__popcll(m1));
            if (m2) atomicAdd(&lh[2], (u32)__popcll(m2));
        }
        if (in) {
            if (hb > 2) { if (hb < OCC_LDS_BINS) atomicAdd(&lh[hb], 1u); else atomicAdd(&occ_hist[hb], 1u); }
            else if (hb == 0) atomicAdd(&lh[0], 1u);
        }
        __syncthreads();
    }
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < OCC_LDS_BINS && i <= max_bin; i += blockDim.x)
        if (lh[i]) atomicAdd(&occ_hist[i], lh[i]);
    // total displacement (slot - home) of the block's runs -> the two words at occ_hist[max_bin + 2] (LRGE_C_TABLE_DISP_SUM)
    for (int s2 = 32; s2 > 0; s2 >>= 1) disp_sum += (u32)__shfl_xor((i32)disp_sum, s2, 64);
    if (lane == 0 && disp_sum) atomicAdd((unsigned long long *)(occ_hist + max_bin + 2 + ((max_bin + 2) & 1)), (unsigned long long)disp_sum);
}

// clears the slots behind the last run (slack + whatever the homes left free at the end); *last_slot from k_place_apply
__global__ __launch_bounds__(256) void k_fill_tail(u64 *__restrict__ ht, u64 n_slots, const u32 *__restrict__ last_slot) {
    ulonglong2 empty; empty.x = HT_EMPTY; empty.y = HT_EMPTY;
    for (u64 q = (u64)*last_slot + 1 + (u64)blockIdx.x * blockDim.x + threadIdx.x; q < n_slots; q += (u64)gridDim.x * blockDim.x)
        *(ulonglong2 *)(ht + 2 * q) = empty;
}

// The table is ORDERED: keys were placed in ascending byte-reversed order, each at its home slot or right behind its
// predecessor, so a probe can stop at the first entry that is not smaller in that order -- an absent key (most query
// minimizers of noisy reads) costs no more than a present one instead of a walk to the next empty slot.  An empty slot
// (all ones) compares as the largest key.
__device__ __forceinline__ bool ht_lookup(const u64 *__restrict__ ht, u64 cap, u32 fix, u64 key, u64 *value) {
    u64 slot = ht_home(key, cap, fix);
    const u64 bk = __builtin_bswap64(key);
    for (;; ++slot) {
        const ulonglong2 e = *(const ulonglong2 *)(ht + 2 * slot);
#ifndef HT_LAZY_VALUE
        // keep the slot ONE 16-byte load: left alone the compiler fetches the key first and the value only on a match --
        // a second dependent trip to the cache for every present key
        asm volatile("" : : "v"((u32)e.y), "v"((u32)(e.y >> 32)));
#endif
        if (__builtin_bswap64(e.x) >= bk) {
            if (e.x != key) return false;
            *value = e.y;
            return true;
        }
    }
}

// ---- partitioned index: occurrence statistics over all parts (mm_idx_cal_max_occ sees ONE index) ----
#define MAX_INDEX_PARTS 32
struct PartTables { const u64 *ht[MAX_INDEX_PARTS]; u64 cap[MAX_INDEX_PARTS]; int n; u32 fix; };

// One lane per slot of part `self` (grid-stride): the key's occurrence count summed over every part; each distinct key
// enters the histogram once, in the lowest part that holds it.  The
// histogram is gathered in LDS and flushed once per block from a small grid (k_place_apply's lesson: millions of
// atomics on the same few dozen bins serialise at L2).
__global__ __launch_bounds__(256) void k_part_global_occ(const u64 *__restrict__ ht, u64 n_slots, PartTables T, int self,
                                                         u32 *__restrict__ hist, u32 max_bin,
                                                         unsigned long long *__restrict__ n_distinct, u32 *__restrict__ gsum) {
    // gsum (may be null): the key's occurrence count over all parts, per slot of THIS part -- k_part_drop then needs no
    // second round of probes (round 3: the two kernels were 272 of 1 848 ms of a full-size C5 step, half of it the re-probing)
    __shared__ u32 lh[OCC_LDS_BINS];
    __shared__ u32 l_first;
    for (u32 i = threadIdx.x; i < OCC_LDS_BINS; i += blockDim.x) lh[i] = 0;
    if (threadIdx.x == 0) l_first = 0;
    __syncthreads();
    const u64 rounds = (n_slots + (u64)gridDim.x * blockDim.x - 1) / ((u64)gridDim.x * blockDim.x);
    for (u64 rd = 0; rd < rounds; ++rd) {
        const u64 slot = (rd * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        bool first = false;
        u32 total = 0;
        if (slot < n_slots) {
            const ulonglong2 e = *(const ulonglong2 *)(ht + 2 * slot);
            if (e.x != HT_EMPTY) {
                u64 sum = ht_count(e.y);
                first = true;
                for (int o = 0; o < T.n; ++o) {
                    if (o == self) continue;
                    u64 v;
                    if (ht_lookup(T.ht[o], T.cap[o], T.fix, e.x, &v)) { sum += ht_count(v); if (o < self) first = false; }
                }
                total = sum > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)sum;
            }
            if (gsum) gsum[slot] = total;
        }
        const u32 hb = total < max_bin ? total : max_bin;
        const u64 mf = __ballot(first);
        if (lane_id() == 0 && mf) atomicAdd(&l_first, (u32)__popcll(mf));
        if (first) { if (hb < OCC_LDS_BINS) atomicAdd(&lh[hb], 1u); else atomicAdd(&hist[hb], 1u); }
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < OCC_LDS_BINS && i <= max_bin; i += blockDim.x) if (lh[i]) atomicAdd(&hist[i], lh[i]);
    if (threadIdx.x == 0 && l_first) atomicAdd(n_distinct, (unsigned long long)l_first);
}

// A key whose GLOBAL count exceeds mid_occ must be dropped in every part: lift its local count above the threshold, which
// is all k_lookup's consumers test (a dropped list is never expanded, so its true length is not needed any more).
// gsum: the global counts k_part_global_occ left per slot (null: memory was short -- the sum over the parts is taken again)
__global__ __launch_bounds__(256) void k_part_drop(u64 *__restrict__ ht, u64 n_slots, PartTables T, int self, u32 mid_occ, const u32 *__restrict__ gsum) {
    const u64 slot = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots) return;
    if (gsum) {
        if (gsum[slot] <= mid_occ) return;                // (0 for an empty slot)
        const u64 v = ht[2 * slot + 1];
        if (ht_count(v) > mid_occ) return;
        ht[2 * slot + 1] = ((v & HT_INLINE) ? 0ULL : (v & ~(u64)HT_CNT_MAX)) | (u64)(mid_occ + 1);
        return;
    }
    const ulonglong2 e = *(const ulonglong2 *)(ht + 2 * slot);
    if (e.x == HT_EMPTY) return;
    const u32 local = ht_count(e.y);
    if (local > mid_occ) return;
    u64 sum = local;
    for (int o = 0; o < T.n && sum <= mid_occ; ++o) {
        if (o == self) continue;
        u64 v;
        if (ht_lookup(T.ht[o], T.cap[o], T.fix, e.x, &v)) sum += ht_count(v);
    }
    // (an inline singleton becomes an ordinary entry whose list is never expanded)
    if (sum > mid_occ) ht[2 * slot + 1] = ((e.y & HT_INLINE) ? 0ULL : (e.y & ~(u64)HT_CNT_MAX)) | (u64)(mid_occ + 1);
}
