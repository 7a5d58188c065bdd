// k_seed.h -- query-side seeding: K4a query occurrence filter (mm_seed_mz_flt), K3 lookup +
// mid_occ filter (mm_seed_collect_all / mm_collect_matches with occ_dist = 0), K4 expansion into
// anchors with skip_seed (mm2:map.c collect_seed_hits), group discovery on the sorted anchors.
#pragma once
#include "internal.h"
#include "k_index.h"

struct KeyLayout {      // composite anchor sort key: qlocal | rid | rev | rpos
    u32 bits_rpos, bits_rid, bits_q;
    __host__ __device__ u32 sh_rev() const { return bits_rpos; }
    __host__ __device__ u32 sh_rid() const { return bits_rpos + 1; }
    __host__ __device__ u32 sh_q() const { return bits_rpos + 1 + bits_rid; }
    __host__ __device__ u32 total() const { return bits_rpos + 1 + bits_rid + bits_q; }
};

struct SeedParams {
    const u64 *ht; u64 ht_cap; u32 ht_fix;   // (ht_fix: see ht_home)
    const u64 *pos;            // index position lists (plain y values, or packed entries when pk_ybits != 0)
    u32 pk_pos1, pk_ybits;     // packed index entry: hash << pk_ybits | rid << pk_pos1 | (pos << 1 | strand)
    const u32 *t_len, *t_rank; // indexed reads
    const u32 *q_len, *q_rank; // query reads
    int mid_occ;
    int check_names;           // both sets carry ranks
    int no_dual;               // MM_F_NO_DUAL (AVA)
};

// y value (rid << 32 | pos << 1 | strand) of index entry i
__device__ __forceinline__ u64 index_y_of(const SeedParams &sp, u64 r) {        // ... from the entry's word
    if (sp.pk_ybits == 0) return r;
    const u64 yb = r & ((1ULL << sp.pk_ybits) - 1);
    return (yb >> sp.pk_pos1) << 32 | (yb & ((1ULL << sp.pk_pos1) - 1));
}
__device__ __forceinline__ u64 index_y(const SeedParams &sp, u64 i) { return index_y_of(sp, sp.pos[i]); }

// hit j of a list that lives at `st` (k_lookup): the inline position of a singleton, or entry st + j of pos[]
__device__ __forceinline__ u64 list_y(const SeedParams &sp, u64 st, u32 j) {
    return (st & HT_INLINE) ? (st & ~HT_INLINE) : index_y(sp, st + j);
}
__device__ __forceinline__ u64 shfl_u64(u64 v, int l) {
    return (u64)(u32)__shfl((i32)(u32)v, l, 64) | (u64)(u32)__shfl((i32)(u32)(v >> 32), l, 64) << 32;
}

// K3: one lane per query minimizer: probe the index.  hs = list start, hc = raw list length (0 when
// the hash is absent).  The mid_occ filter and skip_seed are applied by k_seed_counts, after the query
// occurrence filter had its say.
// hn (may be null): the kept list length k_seed_counts would derive from hc -- written here when no name checks are needed
// (two-set runs without shared reads), so that the common path, in which mm_seed_mz_flt removes nothing, needs no second
// pass over the 10^8 counters.
// hs: where the list lives -- its start in pos[], or HT_INLINE | y for a key that occurs once (k_index.h)
__global__ __launch_bounds__(256) void k_lookup(const u64 *__restrict__ qx, u64 n_mz, SeedParams sp,
                                                u64 *__restrict__ hs, u32 *__restrict__ hc, u32 *__restrict__ hn) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_mz) return;
    u64 v = 0; u64 st = 0; u32 cnt = 0;
    if (ht_lookup(sp.ht, sp.ht_cap, sp.ht_fix, qx[i] >> 8, &v)) { cnt = ht_count(v); st = (v & HT_INLINE) ? v : v >> HT_CNT_BITS; }
    hs[i] = st; hc[i] = cnt;
    if (hn) hn[i] = (cnt != 0 && (i64)cnt <= (i64)sp.mid_occ) ? cnt : 0;   // m[i].n > max_occ -> flt (as in k_seed_counts)
}

// hn = list length of a KEPT seed (0 when absent, removed by mm_seed_mz_flt, or n > mid_occ -> flt),
// hv = hits that survive skip_seed.  Without name checks hv = hn.  With them (AVA, or sets that share reads) one
// wavefront owns 64 consecutive minimizers and walks the concatenation of their hit lists 64 hits at a time, like
// k_expand: the position-list reads are consecutive across the wave and every lane does useful work (one lane per
// minimizer looping over its own list diverged on the list lengths and gathered 8 bytes per lane per step).
__global__ __launch_bounds__(256) void k_seed_counts(const u64 *__restrict__ qy, u64 n_mz, SeedParams sp,
                                                     const u64 *__restrict__ hs, const u32 *__restrict__ hc,
                                                     u32 *__restrict__ hn, u32 *__restrict__ hv) {
    __shared__ u32 kept[4][64];
    const u32 lane = lane_id(), w = threadIdx.x >> 6;
    const u64 w0 = ((u64)blockIdx.x * (blockDim.x >> 6) + w) * 64;
    if (w0 >= n_mz) return;
    const u64 i = w0 + lane;
    const bool in = i < n_mz;
    const u32 c = in ? hc[i] : 0;
    const u32 n = (c != 0 && (i64)c <= (i64)sp.mid_occ) ? c : 0;   // m[i].n > max_occ -> flt
    if (!sp.check_names) {
        if (in) { hn[i] = n; hv[i] = n; }
        return;
    }
    const u32 incl = wave_incl_scan_u32(n);
    const u32 total = (u32)__builtin_amdgcn_readlane((i32)incl, 63);
    const u32 rs = incl - n;
    u32 m_qpos = 0, m_ql = 0, m_qr = 0; u64 m_st = 0;
    if (n) {
        const u64 y = qy[i];
        const u32 q = (u32)(y >> 32);
        m_qpos = (u32)y >> 1; m_ql = sp.q_len[q]; m_qr = sp.q_rank[q]; m_st = hs[i];
    }
    kept[w][lane] = 0;
    // (a wavefront only touches its own row: program order is enough, no barrier)
    for (u32 c0 = 0; c0 < total; c0 += 64) {
        const u32 r = c0 + lane;
        u32 l = 0;
#pragma unroll
        for (u32 step = 32; step > 0; step >>= 1) {
            const u32 v = (u32)__shfl((i32)rs, (int)(l + step), 64);
            l = v <= r ? l + step : l;
        }
        const u32 j = r - (u32)__shfl((i32)rs, (int)l, 64);
        const u64 st = shfl_u64(m_st, (int)l); const u32 qpos = (u32)__shfl((i32)m_qpos, (int)l, 64);
        const u32 ql = (u32)__shfl((i32)m_ql, (int)l, 64), qr = (u32)__shfl((i32)m_qr, (int)l, 64);
        if (r < total) {
            const u64 h = list_y(sp, st, j);
            const u32 rid = (u32)(h >> 32);
            const u32 tr = sp.t_rank[rid];
            bool skip = false;
            if (qr == tr && sp.t_len[rid] == ql && ((u32)h >> 1) == qpos) skip = true;  // NO_DIAG, exact diagonal
            if (sp.no_dual && qr > tr) skip = true;                                     // NO_DUAL, cmp > 0
            if (!skip) atomicAdd(&kept[w][l], 1u);
        }
    }
    if (in) { hn[i] = n; hv[i] = kept[w][lane]; }
}

// per-query anchor totals from the scanned hit counts (aoff has n_mz + 1 entries; differences are exact modulo 2^32, and a
// query's total is far below that)
__global__ void k_query_totals_from_scan(const u32 *__restrict__ aoff, const u32 *__restrict__ qmz_off, u32 nq, u32 *__restrict__ totals) {
    const u32 q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq) totals[q] = aoff[qmz_off[q + 1]] - aoff[qmz_off[q]];
}

// per-query sum of hv (one wave per query)
__global__ __launch_bounds__(256) void k_query_anchor_totals(const u32 *__restrict__ hv, const u32 *__restrict__ qmz_off,
                                                             u32 nq, u32 *__restrict__ totals) {
    u32 q = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (q >= nq) return;
    u32 b = qmz_off[q], e = qmz_off[q + 1];
    u32 s = 0;
    for (u32 i = b + lane_id(); i < e; i += 64) s += hv[i];
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if (lane_id() == 0) totals[q] = s;
}

// anchor value layout: rank<<44 | SELF<<43 | span<<32 | qpos, where rank = index of the seed among the
// query's kept seeds = its index in minimap2's mini_pos[] (what mm_est_err needs for dv)
#define AVAL_RANK_SHIFT 44
#define AVAL_LOW_MASK ((1ULL << AVAL_RANK_SHIFT) - 1)

// K4: expand the kept seeds of the batch into anchors.  One wavefront owns 64 consecutive query
// minimizers; their hit lists are concatenated into one "raw" index space (wave scan of hn) and the
// lanes walk that space 64 hits at a time: each lane finds its (minimizer, hit) by a 6-step search
// through the scanned counts, applies skip_seed, and the survivors are compacted with a ballot so that
// both the reads of the position lists and the (key, val) writes are consecutive across the wave.
// Output order = minimizer order, then list order, exactly as collect_seed_hits emits them.
//
// MARK (k_expand_q below): every emitted anchor also counts its (target, strand) pair in the block's unary bit planes.
#define EXPQ_PLANES 3
#define EXPQ_ITEMS 8
// size classes of k_expand_q (threads per workgroup, log2 of the census buckets): a query's slot of <= EXPQ_SMALL_MAX anchors /
// <= EXPQ_MID_MAX / anything.  The census is LDS, and LDS is residency: 3 planes of 2^14 bits are 6 KB (the chip stays full of
// small queries: C4), of 2^17 bits 48 KB (two 1024-thread workgroups per CU: H. sapiens-scale HiFi, ~36 000 hits per query and part)
#define EXPQ_SMALL_MAX 8192u
#define EXPQ_MID_MAX 32768u
template <int LOG2B> __device__ __forceinline__ u32 pair_bucket(u32 rid_rev) { return (rid_rev * 0x9E3779B1u) >> (32 - LOG2B); }

template <int LOG2B>      // LOG2B != 0: mark the pair census (k_expand_q)
__device__ __forceinline__ void expand_wave_chunk(const u64 w0, const u64 mz_begin, const u64 mz_end, const u64 *__restrict__ qx, const u64 *__restrict__ qy,
                                                  const SeedParams &sp, const u64 *__restrict__ hs, const u32 *__restrict__ hn, const u32 *__restrict__ aoff,
                                                  const u32 *__restrict__ krank, const u32 *__restrict__ qmz_off, const u32 q0, const KeyLayout &kl,
                                                  u64 *__restrict__ akey, u64 *__restrict__ aval, const u32 packed_bits_qy, u32 *planes, const u32 n_planes) {
    // packed_bits_qy != 0: count-only run, one packed u64 per anchor (UnpackParams in k_prims.h), aval unused
    const u32 lane = lane_id();
    const u64 i = w0 + lane;
    const bool in = i < mz_end;
    // the per-minimizer streams are read for every lane, kept seed or not, TOGETHER with the counts: a third of the seeds are
    // kept, so every line of them is fetched anyway, and loads that wait for hn[] cost the wavefront one more trip to memory
    // (this kernel is bound by the length of its chain of dependent loads, not by bytes)
    const u32 n = in ? hn[i] : 0;
    const u64 x = in ? qx[i] : 0, y = in ? qy[i] : 0;
    const u64 st_i = in ? hs[i] : 0;
    const u32 incl = wave_incl_scan_u32(n);
    const u32 total = (u32)__builtin_amdgcn_readlane((i32)incl, 63);
    if (total == 0) return;
    const u32 rs = incl - n;                                     // first raw index of my minimizer
    // per-minimizer fields, fetched by the lanes that expand its hits
    u64 m_st = 0; u32 m_q = 0, m_qpos = 0, m_flags = 0, m_ql = 0, m_qr = 0, m_rank = 0;
    if (n) {
        m_q = (u32)(y >> 32); m_qpos = (u32)y >> 1;
        m_flags = ((u32)y & 1) | ((u32)x & 0xff) << 8;          // strand | span << 8
        m_ql = sp.q_len[m_q];
        m_qr = sp.check_names ? sp.q_rank[m_q] : 0;
        m_st = st_i;
        if (krank) m_rank = (krank[i] - krank[qmz_off[m_q]]) & 0xFFFFFu;    // (null: count-only run, the packed anchor has no rank field)
    }
    u32 o = aoff[w0] - aoff[mz_begin];                           // anchors written so far (wave-uniform); aoff: scan over ALL query minimizers
    o = (u32)__builtin_amdgcn_readfirstlane((i32)o);
    // ONE ROUND AHEAD (round 6): the kernel is bound by its chain of dependent trips to memory (position list -> target's name rank ->
    // its length), so the position-list load of round c0 + 64 is issued before round c0 is worked on: its (minimizer, hit) pair needs
    // only the scanned counts, which are in registers.
    // locate: the lane's hit of the round that starts at c0 -- the largest lane l with rs[l] <= r (zero-length lists share their rs with
    // the next one and lose) -- and its position-list entry, on its way
    // (h: the entry's WORD -- what a packed index needs to make a y of it waits until the word is used, or the load would be waited for here)
    auto locate = [&](const u32 c0, u32 &l, bool &inr, bool &inl, u64 &h) {
        const u32 r = c0 + lane;
        l = 0;
#pragma unroll
        for (u32 step = 32; step > 0; step >>= 1) {
            const u32 v = (u32)__shfl((i32)rs, (int)(l + step), 64);
            l = v <= r ? l + step : l;
        }
        const u32 j = r - (u32)__shfl((i32)rs, (int)l, 64);
        const u64 st = shfl_u64(m_st, (int)l);
        inr = r < total;
        h = !inr ? 0 : (st & HT_INLINE) ? st : sp.pos[(st & ~HT_INLINE) + j];
        inl = (st & HT_INLINE) != 0;
    };
    u32 l_next; bool in_next, inl_next; u64 h_next;
    locate(0, l_next, in_next, inl_next, h_next);
    for (u32 c0 = 0; c0 < total; c0 += 64) {
        const u32 r = c0 + lane;
        const u32 l = l_next; bool keep = in_next; const bool inl = inl_next; const u64 hw = h_next;
        if (c0 + 64 < total) locate(c0 + 64, l_next, in_next, inl_next, h_next);      // (wave-uniform)
        const u32 q = (u32)__shfl((i32)m_q, (int)l, 64);
        const u32 qpos = (u32)__shfl((i32)m_qpos, (int)l, 64), fl = (u32)__shfl((i32)m_flags, (int)l, 64);
        const u32 ql = (u32)__shfl((i32)m_ql, (int)l, 64), qr = (u32)__shfl((i32)m_qr, (int)l, 64);
        const u32 rk = (u32)__shfl((i32)m_rank, (int)l, 64);
        (void)r;
        u64 key = 0, val = 0;
        if (keep) {
            const u64 h = inl ? (hw & ~HT_INLINE) : index_y_of(sp, hw);
            const u32 rid = (u32)(h >> 32), rpos = (u32)h >> 1, qstrand = fl & 1, span = fl >> 8;
            u64 self = 0;
            if (sp.check_names) {
                const u32 tr = sp.t_rank[rid];
                if (qr == tr && sp.t_len[rid] == ql) {
                    if (rpos == qpos) keep = false;
                    if (((u32)h & 1) == qstrand) self = 1ULL << 43;  // MM_SEED_SELF (unused by chaining)
                }
                if (sp.no_dual && qr > tr) keep = false;
            }
            const bool rev = ((u32)h & 1) != qstrand;
            const u32 yq_rev = ql - (qpos + 1 - span) - 1;
            key = (u64)(q - q0) << kl.sh_q() | (u64)rid << kl.sh_rid() | (u64)(rev ? 1 : 0) << kl.sh_rev() | rpos;
            val = (u64)rk << AVAL_RANK_SHIFT | self | (u64)span << 32 | (rev ? yq_rev : qpos);
            if (LOG2B != 0 && keep) {
                // unary count of the pair's bucket: plane j is set by the (j + 1)-th anchor that arrives (each atomicOr hands exactly
                // one arrival the "was clear" answer), so plane n_planes - 1 set <=> at least n_planes anchors hashed here
                constexpr u32 WORDS = LOG2B ? 1u << (LOG2B > 5 ? LOG2B - 5 : 0) : 1u;
                const u32 b = pair_bucket<LOG2B ? LOG2B : 5>(rid << 1 | (rev ? 1u : 0u)), wd = b >> 5, bit = 1u << (b & 31);
                if (!(planes[(n_planes - 1) * WORDS + wd] & bit))
                    for (u32 lvl = 0; lvl < n_planes; ++lvl)
                        if (!(atomicOr(&planes[lvl * WORDS + wd], bit) & bit)) break;
            }
        }
        const u64 km = __ballot(keep);
        if (keep) {
            const u32 d = o + (u32)__popcll(km & lanemask_lt());
            if (packed_bits_qy) {
                const u32 sb = kl.sh_q();
                akey[d] = (key & ((1ULL << sb) - 1)) | (val & 0xFFFFFFFFULL) << sb | ((val >> 32) & 0xff) << (sb + packed_bits_qy) |
                          ((val >> 43) & 1) << (sb + packed_bits_qy + 8);
            } else { akey[d] = key; aval[d] = val; }
        }
        o += (u32)__popcll(km);
    }
}

__global__ __launch_bounds__(256) void k_expand(const u64 *__restrict__ qx, const u64 *__restrict__ qy, u64 mz_begin,
                                                u64 mz_end, SeedParams sp, const u64 *__restrict__ hs,
                                                const u32 *__restrict__ hn, const u32 *__restrict__ aoff,
                                                const u32 *__restrict__ krank, const u32 *__restrict__ qmz_off, u32 q0,
                                                KeyLayout kl, u64 *__restrict__ akey, u64 *__restrict__ aval, u32 packed_bits_qy) {
    const u64 w0 = mz_begin + ((u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64;
    if (w0 >= mz_end) return;
    expand_wave_chunk<0>(w0, mz_begin, mz_end, qx, qy, sp, hs, hn, aoff, krank, qmz_off, q0, kl, akey, aval, packed_bits_qy, nullptr, 0);
}

// K4 with the DEAD-PAIR FILTER (count-only runs; round 4).  A chain needs min_cnt anchors on ONE (target, strand) pair, and the
// group stage only ever chains pairs with at least min_n of them (OverlapRun::plan) -- but most anchors of a real job are
// singletons: chance k-mer matches on some unrelated read (H. sapiens-scale HiFi: 71 % of 10.9 G anchors per step, 7.4 G groups
// of which 41 M are chained; C4: 55 %).  They used to travel through the whole anchor sort before the group stage dropped them.
// Here ONE workgroup owns a query: while its wavefronts expand the hits (the code of k_expand, chunk by chunk of 64 minimizers,
// into the same slots), every anchor counts its pair in a hashed unary counter in LDS (n_planes = min(min_n, 3) bit planes of
// 2^17 buckets); a bucket's count bounds the size of every pair hashed into it from above, so an anchor whose bucket stayed below
// n_planes belongs to a pair the group stage would discard -- dropping it changes no chain, no count.  Then the block streams
// over the slot once more and compacts the survivors to its front, in order (the sort is stable: ties keep emission order).
// kept[q - q0] tells the host how many; the sort gathers them from the sparse slots into the dense layout (SegTile.src).
// qlist: the queries (relative to q0) of this launch's size class.
template <int THREADS, int LOG2B>
__global__ __launch_bounds__(THREADS) void k_expand_q(const u64 *__restrict__ qx, const u64 *__restrict__ qy, u64 mz_begin, SeedParams sp,
                                                      const u64 *__restrict__ hs, const u32 *__restrict__ hn, const u32 *__restrict__ aoff,
                                                      const u32 *__restrict__ qmz_off, u32 q0, const u32 *__restrict__ qlist, KeyLayout kl,
                                                      u64 *__restrict__ akey, u32 packed_bits_qy, u32 n_planes, u32 *__restrict__ kept) {
    constexpr u32 WORDS = 1u << (LOG2B - 5), NW = THREADS / 64;
    __shared__ u32 planes[EXPQ_PLANES * WORDS];
    __shared__ u32 wsum[2][NW];
    const u32 ql = qlist[blockIdx.x], q = q0 + ql;
    const u64 mb = qmz_off[q], me = qmz_off[q + 1];
    const u32 seg0 = aoff[mb] - aoff[mz_begin], tot = aoff[me] - aoff[mb];
    if (tot == 0) { if (threadIdx.x == 0) kept[ql] = 0; return; }      // (block-uniform)
    for (u32 i = threadIdx.x; i < n_planes * WORDS; i += THREADS) planes[i] = 0;
    __syncthreads();
    const u32 wv = threadIdx.x >> 6, lane = lane_id();
    for (u64 w0 = mb + 64ull * wv; w0 < me; w0 += 64ull * NW)
        expand_wave_chunk<LOG2B>(w0, mz_begin, me, qx, qy, sp, hs, hn, aoff, nullptr, qmz_off, q0, kl, akey, nullptr, packed_bits_qy, planes, n_planes);
    __syncthreads();                     // the slot is written (same workgroup: visible), the planes are final
    const u32 *top = planes + (n_planes - 1) * WORDS;
    const u32 sb = kl.sh_q(), sh_rev = kl.sh_rev();
    const u64 smask = (1ULL << sb) - 1;
    u64 *slot = akey + seg0;
    u32 base = 0, it = 0;
    for (u32 i0 = 0; i0 < tot; i0 += THREADS * EXPQ_ITEMS, ++it) {
        // wave-major item order (wave w: items i0 + w * 64 * ITEMS + r * 64 + lane), so that ranks follow the slot's order
        const u32 l0 = i0 + wv * (64 * EXPQ_ITEMS) + lane;
        u64 v[EXPQ_ITEMS]; u64 km[EXPQ_ITEMS];
        u32 wtot = 0;
#pragma unroll
        for (int r = 0; r < EXPQ_ITEMS; ++r) v[r] = l0 + (u32)r * 64 < tot ? slot[l0 + (u32)r * 64] : 0;
#pragma unroll
        for (int r = 0; r < EXPQ_ITEMS; ++r) {
            bool k = l0 + (u32)r * 64 < tot;
            if (k) { const u32 b = pair_bucket<LOG2B>((u32)((v[r] & smask) >> sh_rev)); k = (top[b >> 5] >> (b & 31)) & 1; }
            km[r] = __ballot(k);
            wtot += (u32)__popcll(km[r]);
        }
        if (lane == 0) wsum[it & 1][wv] = wtot;
        __syncthreads();                 // every load of this round has landed before any store of it (stores go to indices below i0 + round size)
        u32 run = base, all = 0;
#pragma unroll
        for (u32 w = 0; w < NW; ++w) { const u32 c = wsum[it & 1][w]; if (w < wv) run += c; all += c; }
#pragma unroll
        for (int r = 0; r < EXPQ_ITEMS; ++r) {
            if ((km[r] >> lane) & 1) slot[run + (u32)__popcll(km[r] & lanemask_lt())] = v[r];
            run += (u32)__popcll(km[r]);
        }
        base += all;
    }
    if (threadIdx.x == 0) kept[ql] = base;
}

// ------------------------------------------------------------------------------------------
// K4a: query occurrence filter (mm2:seed.c mm_seed_mz_flt), run on the minimizers that are present in
// the index.  (x, query|index) pairs are sorted by x and then stably by query; a run of identical
// (query, x) longer than both thresholds removes all its members (hc = 0: as if absent).
// ------------------------------------------------------------------------------------------
__global__ void k_flag_nonzero(const u32 *__restrict__ v, u64 n, u32 *__restrict__ flag) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = v[i] ? 1u : 0u;
}

__global__ void k_qocc_keys(const u64 *__restrict__ qx, const u64 *__restrict__ qy, const u32 *__restrict__ flag,
                            const u32 *__restrict__ fpos, u64 n, u64 *__restrict__ k_x, u64 *__restrict__ v_idx) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const u32 o = fpos[i];
    k_x[o] = qx[i];
    v_idx[o] = (qy[i] >> 32) << 32 | (u32)i;  // value carries (query, original index); needs n < 2^32
}

// Conservative pre-check for K4a: the filter can only remove a value that occurs more than mid_occ times
// inside one query.  One block per query counts its indexed minimizers into 8192 hashed LDS buckets; a
// bucket count is an upper bound of the multiplicity of every value that falls into it, so when no bucket
// of any query exceeds mid_occ the filter removes nothing and the exact (sort-based) pass is skipped.
#ifndef QOCC_BUCKETS
#define QOCC_BUCKETS 8192
#endif
__global__ __launch_bounds__(256) void k_qocc_check(const u64 *__restrict__ qx, const u32 *__restrict__ hc,
                                                    const u32 *__restrict__ qmz_off, u32 nq, int mid_occ, u32 *__restrict__ flag) {
    __shared__ u32 cnt[QOCC_BUCKETS];
    const u32 q = blockIdx.x;
    if (q >= nq) return;
    const u32 b = qmz_off[q], e = qmz_off[q + 1];
    if ((i64)(e - b) <= (i64)mid_occ) return;                              // mv->n <= q_occ_max: filter off
    // as many buckets as it takes to keep the load below 1/2 (a power of two, at most QOCC_BUCKETS): short queries
    // do not pay for clearing 32 KB
    u32 nbk = 256;
    while (nbk < QOCC_BUCKETS && nbk < 2 * (e - b)) nbk <<= 1;
    for (u32 i = threadIdx.x; i < nbk; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    bool hit = false;
    for (u32 i = b + threadIdx.x; i < e; i += blockDim.x) {
        if (hc[i] == 0) continue;
        const u32 h = (u32)(((qx[i] >> 8) * 0x9E3779B97F4A7C15ULL) >> 51) & (nbk - 1);  // up to 13 bits
        if ((i64)atomicAdd(&cnt[h], 1u) + 1 > (i64)mid_occ) hit = true;
    }
    if (hit) { atomicOr(flag, 1u); flag[1 + q] = 1u; }      // flag[0]: any query; flag[1 + q]: this one (benign same-value race)
}

// flag = present in the index AND the minimizer's query is one the pre-check could not clear (qsel[1 + q]; null = every query):
// the exact pass then sorts the minimizers of those queries only -- a false alarm on one very long read (its buckets fill up)
// costs microseconds, not a sort of every query's minimizers
__global__ void k_flag_present_sel(const u32 *__restrict__ hc, const u64 *__restrict__ qy, const u32 *__restrict__ qsel, u64 n, u32 *__restrict__ flag) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (hc[i] != 0 && (!qsel || qsel[1 + (u32)(qy[i] >> 32)] != 0)) ? 1u : 0u;
}

// One lane per element; run heads do the work (runs are short except for the pathological ones this
// filter exists for).
__global__ void k_qocc_mark(const u64 *__restrict__ sx, const u64 *__restrict__ sv, u64 n,
                            const u32 *__restrict__ qmz_off, int mid_occ, float q_occ_frac, u32 *__restrict__ hc) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 x = sx[i]; u32 q = (u32)(sv[i] >> 32);
    if (i > 0 && sx[i - 1] == x && (u32)(sv[i - 1] >> 32) == q) return;  // not a run head
    u64 j = i + 1;
    while (j < n && sx[j] == x && (u32)(sv[j] >> 32) == q) ++j;
    i32 cnt = (i32)(j - i);
    u32 nq = qmz_off[q + 1] - qmz_off[q];
    if ((i64)nq <= (i64)mid_occ) return;                                   // mv->n <= q_occ_max: filter off
    if (cnt > mid_occ && (float)cnt > (float)(u64)nq * q_occ_frac)
        for (u64 t = i; t < j; ++t) hc[(u32)sv[t]] = 0;
}

// Partitioned index: a query minimizer's occurrence count over all parts.  A key that is too frequent globally carries
// mid_occ + 1 in every part that holds it (k_part_drop), all other local counts add up to the global one, so the clamped sum
// tells kept (0 < n <= mid_occ) from repetitive (n > mid_occ) exactly as the one index would.
__global__ void k_hc_accumulate(const u32 *__restrict__ hc, u64 n, u32 mid_occ, u32 *__restrict__ acc) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 s = (u64)acc[i] + hc[i];
    acc[i] = s > mid_occ ? mid_occ + 1 : (u32)s;
}

// kept seeds by their occurrence count over all parts (see k_hc_accumulate): 0 < n <= mid_occ
__global__ void k_flag_kept(const u32 *__restrict__ g, u64 n, u32 mid_occ, u32 *__restrict__ flag) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 c = g[i];
    flag[i] = (c != 0 && c <= mid_occ) ? 1u : 0u;
}

// Per-query PAF statistics (mm2:seed.c mm_collect_matches, mm2:esterr.c mm_est_err): rep_len = length of
// the query covered by filtered (n > mid_occ) seeds, sum_span / n_kept -> avg_k of the kept seeds.
__global__ void k_query_paf_stats(const u64 *__restrict__ qx, const u64 *__restrict__ qy, const u32 *__restrict__ hc,
                                  const u32 *__restrict__ qmz_off, u32 nq, int mid_occ, i32 *__restrict__ rep_len,
                                  u64 *__restrict__ sum_span, u32 *__restrict__ n_kept) {
    u32 q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    i32 rep_st = 0, rep_en = 0, rl = 0; u64 ss = 0; u32 nk = 0;
    for (u32 i = qmz_off[q]; i < qmz_off[q + 1]; ++i) {
        const u32 c = hc[i];
        if (c == 0) continue;
        const i32 span = (i32)(qx[i] & 0xff);
        if ((i64)c > (i64)mid_occ) {
            const i32 en = (i32)((u32)qy[i] >> 1) + 1, st = en - span;
            if (st > rep_en) { rl += rep_en - rep_st; rep_st = st; rep_en = en; }
            else rep_en = en;
        } else { ss += (u64)span; ++nk; }
    }
    rl += rep_en - rep_st;
    rep_len[q] = rl; sum_span[q] = ss; n_kept[q] = nk;
}

// ------------------------------------------------------------------------------------------
// groups on the sorted anchors: a group = one (query, target, strand)
// ------------------------------------------------------------------------------------------
// (group boundaries: compact_heads() in k_prims.h on key >> bits_rpos)

#define GB_CHUNK 8192   // groups per block in k_group_fill
