// sort_forms.h -- three forms of the index sort that were built, measured and NOT kept (rounds 2-3), with the numbers in their
// headers; moved out of the product library (lrge_amd/csrc/k_prims.h) in round 4 so that it only holds what the default path
// launches.  Development record + bench material: tools/micro/sort_bench.hip includes this after k_prims.h.
//   * one-sweep (decoupled look-back): exact, 6.0-6.3 ms against the three-kernel passes' 4.9-5.2 on 242 M keys;
//   * hybrid (two most-significant-digit passes, the remaining digits inside LDS): 4.6-4.9 ms alone, slower inside the step;
//   * 10-bit digits (three passes instead of four over a 30-bit hash): 5.7-6.1 ms.
// Not compiled into liblrge_hip.so, not reachable through any option.
#pragma once
#include "../../lrge_amd/csrc/k_prims.h"

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

// keys-only LSD sort with 10-bit digits (k_rs_hist / k_rs_scatter are templates on the digit width)
static int radix_sort_keys_db10(lrge_hip_ctx *ctx, Scratch &sc, u64 *k0, u64 *k1, u64 n, int begin_bit, int nbits, u64 **res, bool reverse_digits) {
    *res = k0;
    if (n <= 1 || nbits <= 0) return LRGE_OK;
    if (n >= (1ULL << 32)) return LRGE_ERR_INVALID;
    const int DB = 10;
    const u32 nb = (u32)div_up(n, RS_TILE);
    ALLOC_OR_FAIL(hist, sc, u32, ((u64)1 << DB) * nb);
    const int passes = (nbits + DB - 1) / DB;
    u64 *ki = k0, *ko = k1;
    for (int p = 0; p < passes; ++p) {
        const int d = reverse_digits ? passes - 1 - p : p;
        const int shift = begin_bit + d * DB;
        UnpackParams up{0, 0, 0, nbits - d * DB >= DB ? (1u << DB) - 1u : (1u << (nbits - d * DB)) - 1u};
        hipLaunchKernelGGL((k_rs_hist<false, 10>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, n, shift, nb, hist, (const SegTile *)nullptr, up.dmask);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, hist, hist, ((u64)1 << DB) * nb, nullptr);
        if (rc) return rc;
        hipLaunchKernelGGL((k_rs_scatter<false, RS_MODE_KEYS, 10>), dim3(nb), dim3(RS_THREADS), 0, ctx->stream, ki, (const u64 *)nullptr, ko, (u64 *)nullptr, n, shift, nb,
                           hist, (const SegTile *)nullptr, up);
        KCHK(ctx);
        u64 *t = ki; ki = ko; ko = t;
    }
    sc.drop(hist);
    *res = ki;
    return LRGE_OK;
}
