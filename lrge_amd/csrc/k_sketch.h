// k_sketch.h -- K0 pack (ASCII -> 2-bit + ambiguity mask) and K1 minimizer sketch.
//
// K1 restates minimap2's mm_sketch (mm2:sketch.c; called from mm_idx_gen for targets and from
// collect_minimizers inside mm_map for queries -- aligner.rs:181-185, :231-241) as a data-parallel
// kernel: a read is cut into chunks of SK_CHUNK bases, one lane per chunk.  The sequential state of
// mm_sketch at position i is a pure function of the last w+k-1 steps, so each lane replays a halo of
// w+k-1 steps in front of its chunk and then emits exactly the minimizers the scalar loop would emit
// for the steps it owns.  The w-entry window lives in registers as a shift register (w is a
// template constant, so every index is static), which replaces the scalar code's ring buffer.
#pragma once
#include <type_traits>
#include "internal.h"
#include "k_prims.h"

#ifndef SK_CHUNK
#define SK_CHUNK 128            // bases per lane
#endif
#define SK_THREADS 256
#define ST_REDO 0xFFFFFFFFu     // counts[c] of a chunk the tile form (k_sketch_tile.h) left to k_sketch_direct(only_marked)

// ------------------------------------------------------------------------------------------
// K0: one thread per 32-base word of the packed image
// ------------------------------------------------------------------------------------------
// A/a=0 C/c=1 G/g=2 T/t/U/u=3 else 4 (mm2: seq_nt4_table).  For the five letters (either case) bits 1..2 of the ASCII code
// are A 00, C 01, G 11, T/U 10, so the 2-bit code is b ^ (b >> 1) of those two bits; a byte is one of the letters iff
// it lies in [0x40, 0x7f] and bit (c & 31) of {1, 3, 7, 20, 21} is set.
__device__ __forceinline__ u32 nt4_code(u32 c) {
    const u32 b = (c >> 1) & 3;
    const bool letter = (c & 0xC0u) == 0x40u && ((0x0030008Au >> (c & 31)) & 1u);
    return letter ? (b ^ (b >> 1)) : 4u;
}

#define PACK_THREADS 256
#define PACK_ITEMS 4                                       // 16-base pieces per lane
#define PACK_WORDS (PACK_THREADS / 2 * PACK_ITEMS)         // words per block
#define PACK_LREADS 63                                     // reads whose offsets a block keeps in LDS
// Two lanes per 32-base word of the packed image, 16 bases each: the lanes of a wavefront read consecutive 16-byte pieces
// of the ASCII (one coalesced 1 KB run per load wherever a read continues), convert them to 32 bits + a 16-bit mask, and
// the even lane writes the word after one exchange with its neighbour.  `blk_read[b]` (computed on the host with the
// word offsets) is the read that holds the first word of block b; the block fetches the offsets of that read and the
// PACK_LREADS after it into LDS in ONE round of loads, every lane finds the reads of its PACK_ITEMS pieces there, and all its
// ASCII loads are in flight together.  (One piece per lane behind a chain of four dependent global loads -- block's read,
// next read's first word, base offsets, bases -- ran at 0.66 TB/s: latency, not bytes.)
__global__ __launch_bounds__(PACK_THREADS) void k_pack(const u8 *__restrict__ ascii, const u64 *__restrict__ boff,
                                                       const u64 *__restrict__ woff, const u32 *__restrict__ blk_read, u32 n_reads,
                                                       u64 n_words, u64 *__restrict__ pack, u32 *__restrict__ nmask) {
    __shared__ u64 s_woff[PACK_LREADS + 1], s_boff[PACK_LREADS + 1];
    const u32 r0 = blk_read[blockIdx.x];
    const u32 nl = n_reads + 1 - r0 < PACK_LREADS + 1 ? n_reads + 1 - r0 : PACK_LREADS + 1;      // entries r0 .. r0 + nl - 1 of the offset arrays
    if (threadIdx.x < nl) { s_woff[threadIdx.x] = woff[r0 + threadIdx.x]; s_boff[threadIdx.x] = boff[r0 + threadIdx.x]; }
    __syncthreads();
    const u32 half = threadIdx.x & 1;
    u64 wid[PACK_ITEMS]; const u8 *src[PACK_ITEMS]; u32 cnt[PACK_ITEMS];
    uint4 q[PACK_ITEMS];
#pragma unroll
    for (int it = 0; it < PACK_ITEMS; ++it) {
        wid[it] = (u64)blockIdx.x * PACK_WORDS + (u32)it * (PACK_THREADS / 2) + (threadIdx.x >> 1);
        cnt[it] = 0; src[it] = ascii; q[it] = make_uint4(0, 0, 0, 0);
        if (wid[it] < n_words) {
            // the read that holds the word: the last cached one whose first word is <= wid (empty reads own no word and are stepped over)
            u32 lo = 0, hi = nl - 1;                               // (entry nl - 1 may be the terminator woff[n_reads] = n_words > wid)
            while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if (s_woff[mid] <= wid[it]) lo = mid; else hi = mid - 1; }
            u64 w0 = s_woff[lo], b0 = s_boff[lo], b1;
            if (lo + 1 < nl) b1 = s_boff[lo + 1];
            else {                                                  // past the cached reads (a run of tiny reads): the slow way
                u32 r = r0 + lo;
                while (r + 1 < n_reads && woff[r + 1] <= wid[it]) ++r;
                w0 = woff[r]; b0 = boff[r]; b1 = boff[r + 1];
            }
            const u64 pos0 = (wid[it] - w0) * 32 + 16 * half, len = b1 - b0;
            src[it] = ascii + b0 + pos0;
            cnt[it] = pos0 >= len ? 0u : (u32)(len - pos0 < 16 ? len - pos0 : 16);
            if (cnt[it] == 16) __builtin_memcpy(&q[it], src[it], 16);
        }
    }
#pragma unroll
    for (int it = 0; it < PACK_ITEMS; ++it) {
        const bool in = wid[it] < n_words;
        u32 bits = 0, m = 0xffffu;
        if (in) {
            u32 v[4] = {q[it].x, q[it].y, q[it].z, q[it].w};
            if (cnt[it] < 16) {                                     // the last word of a read: never read past its end
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) { const u32 i = 4 * j + t; if (i < cnt[it]) v[j] |= (u32)src[it][i] << (8 * t); }
                }
            }
            m = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const u32 c = nt4_code((v[j] >> (8 * t)) & 0xffu);
                    bits |= (c & 3) << (2 * (4 * j + t));
                    m |= (c >> 2) << (4 * j + t);
                }
            }
            if (cnt[it] < 16) { m |= 0xffffu & (~0u << cnt[it]); bits &= cnt[it] ? (~0u >> (32 - 2 * cnt[it])) : 0u; }   // padding past the end is "ambiguous"
        }
        const u32 obits = (u32)__shfl_xor((i32)bits, 1, 64), om = (u32)__shfl_xor((i32)m, 1, 64);
        if (in && half == 0) {
            pack[wid[it]] = (u64)bits | (u64)obits << 32;
            nmask[wid[it]] = (m & 0xffffu) | om << 16;
        }
    }
}

// ------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 mm_hash64(u64 key, u64 mask) {  // mm2:sketch.c:hash64
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

// the same mixer in 32-bit arithmetic: exact whenever the mask has <= 32 bits (every step is masked, so only the
// low bits of each sum matter; k = 15 -> 30 bits)
__device__ __forceinline__ u32 mm_hash32(u32 key, u32 mask) {
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

struct BaseReader {  // sequential reader over the packed image of one read
    const u64 *pack; const u32 *nmask;
    u64 wbase; u64 w; u32 m; i32 cur_word;
    __device__ __forceinline__ void init(const u64 *p, const u32 *nm, u64 word_base) { pack = p; nmask = nm; wbase = word_base; cur_word = -1; w = 0; m = 0; }
    __device__ __forceinline__ u32 get(i32 pos) {  // returns 0..3, or 4 for ambiguous
        i32 wi = pos >> 5;
        if (wi != cur_word) { cur_word = wi; w = pack[wbase + wi]; m = nmask[wbase + wi]; }
        u32 sh = pos & 31;
        return ((m >> sh) & 1) ? 4u : (u32)((w >> (2 * sh)) & 3);
    }
};

// XT / YT: u64 for the general (x, y) pairs; u32 for the narrow form (non-HPC, 2k <= 32), where x is the bare hash
// (the span is always k for a valid k-mer, so the order of hash<<8|span is the order of the hash) and y is
// pos<<1|strand (the read id is constant inside a read)
template <int W, typename XT = u64, typename YT = u64>
struct MinWindow {  // shift-register form of mm_sketch's ring buffer + running minimum
    static constexpr XT NONE = (XT)~(XT)0;
    XT wx[W]; YT wy[W];
    XT minx; YT miny;
    int mi;  // index of the current minimum inside the window (W-1 = newest), -1 = evicted
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < W; ++j) { wx[j] = NONE; wy[j] = (YT)~(YT)0; }
        minx = NONE; miny = (YT)~(YT)0; mi = 0;
    }
    // One step of the scalar loop after `info` was computed.  l is the valid-base count AFTER this
    // step's increment.  emit(x,y) is called in exactly the scalar order.
    template <int K, typename Emit>
    __device__ __forceinline__ void step(XT ix, YT iy, int l, Emit &&emit) {
        const bool evicted = (mi == 0);
#pragma unroll
        for (int j = 0; j + 1 < W; ++j) { wx[j] = wx[j + 1]; wy[j] = wy[j + 1]; }
        wx[W - 1] = ix; wy[W - 1] = iy;
        mi -= 1;
        if (l == W + K - 1 && minx != NONE) {  // first full window: flush identical minima
            // (identical minima = the same k-mer twice inside w positions: rare, so ONE branch guards the per-slot ones)
            bool any = false;
#pragma unroll
            for (int j = 0; j + 1 < W; ++j) any |= minx == wx[j] && wy[j] != miny;
            if (any) {
#pragma unroll
                for (int j = 0; j + 1 < W; ++j)
                    if (minx == wx[j] && wy[j] != miny) emit(wx[j], wy[j]);
            }
        }
        // "the old minimum leaves": one emission site for both ways it happens (a new element that is not larger, or the
        // minimum falling out of the window) -- the lanes of a wavefront take both in the same step all the time, and
        // every site carries the whole output-buffer code
        const bool le = ix <= minx;
        if (minx != NONE && ((le && l >= W + K) || (!le && evicted && l >= W + K - 1))) emit(minx, miny);
        if (le) {
            minx = ix; miny = iy; mi = W - 1;
        } else if (evicted) {
            minx = NONE;
#pragma unroll
            for (int j = 0; j < W; ++j)
                if (minx >= wx[j]) { minx = wx[j]; miny = wy[j]; mi = j; }
            if (l >= W + K - 1 && minx != NONE) {
                bool any = false;
#pragma unroll
                for (int j = 0; j < W; ++j) any |= minx == wx[j] && miny != wy[j];
                if (any) {
#pragma unroll
                    for (int j = 0; j < W; ++j)
                        if (minx == wx[j] && miny != wy[j]) emit(wx[j], wy[j]);
                }
            }
        }
    }
};

// ---- homopolymer-compressed sketch (ava-pb): the steps of mm_sketch's loop as bits ----
// With HPC a step of the scalar loop is a whole homopolymer run (or one ambiguous base).  Scanned base by base -- "while the
// next base equals this one" -- the lanes of a wavefront wait for the longest run among them at every step (~4 bases for
// random sequence) and pay a word lookup per base.  Here the steps of 16 bases are found at once: bit 2j of `starts`
// is set when base j begins a step (it differs from base j - 1, or either of them is ambiguous), ~10 operations per half word,
// and the loop goes from set bit to set bit: the run's length is the distance to the next one.
#define SK_EVEN 0x5555555555555555ULL
__device__ __forceinline__ u64 spread32(u32 m) {      // bit j -> bit 2j
    u64 x = m;
    x = (x | x << 16) & 0x0000FFFF0000FFFFULL;
    x = (x | x << 8) & 0x00FF00FF00FF00FFULL;
    x = (x | x << 4) & 0x0F0F0F0F0F0F0F0FULL;
    x = (x | x << 2) & 0x3333333333333333ULL;
    x = (x | x << 1) & SK_EVEN;
    return x;
}
// HALF WORDS (round 6): the lanes of a wavefront use up their words at different steps, so the block that fetches the next one and finds
// its starts runs in nearly every step of the wavefront whatever its share of the lanes -- its price per run counts, not how often a lane
// needs it.  On 16 bases at a time every operation is one 32-bit instruction (a 64-bit shift, compare, find-first-set is two to four).
#define SK_EVEN32 0x55555555u
__device__ __forceinline__ u32 spread16(u32 m) {      // bit j -> bit 2j, j < 16
    u32 x = m & 0xFFFFu;
    x = (x | x << 8) & 0x00FF00FFu;
    x = (x | x << 4) & 0x0F0F0F0Fu;
    x = (x | x << 2) & 0x33333333u;
    x = (x | x << 1) & SK_EVEN32;
    return x;
}
struct RunWords {            // "word" = 16 bases = one half of a packed word
    static constexpr int LOG = 4, N = 16;
    const u32 *pack; const u32 *nmask; i32 len;
    u32 w, ns, starts;      // the current half: codes, ambiguity mask (bit 2j), step starts (bit 2j), clipped to the read
    __device__ __forceinline__ void init(const u64 *p, const u32 *nm, u64 word_base, i32 l) { pack = (const u32 *)(p + word_base); nmask = nm + word_base; len = l; }
    __device__ __forceinline__ u32 amb(i32 wi) const { return (nmask[wi >> 1] >> ((wi & 1) * 16)) & 0xFFFFu; }
    // prev2 / prevN: code and ambiguity of the last base of half wi - 1 (wi == 0: prevN = true, base 0 always starts a step)
    __device__ __forceinline__ void set(i32 wi, u32 half, u32 m, u32 prev2, bool prevN) {
        w = half;
        ns = m ? spread16(m) : 0u;
        const u32 x = w ^ (w << 2 | prev2);
        u32 eq = ~(x | x >> 1) & SK_EVEN32;
        eq &= ~(ns | ns << 2 | (prevN ? 1u : 0u));
        starts = ~eq & SK_EVEN32;
        const i32 nvalid = len - wi * N;
        if (nvalid < N) starts &= (1u << (2 * nvalid)) - 1;
    }
    __device__ __forceinline__ void load(i32 wi) {                 // random access: fetches half wi - 1 too
        u32 prev2 = 0; bool prevN = true;
        if (wi > 0) { prev2 = pack[wi - 1] >> 30; prevN = (amb(wi - 1) >> 15) != 0; }
        set(wi, pack[wi], amb(wi), prev2, prevN);
    }
    __device__ __forceinline__ void next(i32 wi) {                 // half wi, the current one being wi - 1
        const u32 prev2 = w >> 30; const bool prevN = (ns >> 30) != 0;
        set(wi, pack[wi], amb(wi), prev2, prevN);
    }
    __device__ __forceinline__ u32 code(u32 bit) const { return ((ns >> bit) & 1) ? 4u : ((w >> bit) & 3); }
};

// POS_OWN = false: the chunk owns the loop steps that START in [s, e) and writes what mm_sketch writes DURING those steps (the
// minimizer written at a step lies up to w steps back -- possibly in the chunk before).  POS_OWN = true (k_sketch_redo, behind
// k_sketch_tile.h): the chunk owns the minimizers whose own step ENDS in [s, e), whenever they are written -- the attribution of the
// tile form, so that the two can share a read: the replay starts one step earlier, goes on for w steps past e, and filters by position.
template <int K, int W, bool POS_OWN = false, typename Emit>
__device__ __forceinline__ void sketch_chunk_hpc(const u64 *pack, const u32 *nmask, u64 word_base, i32 len, u32 rid, i32 s, i32 e, Emit &&emit) {
    constexpr u64 mask = (1ULL << (2 * K)) - 1;
    constexpr int shift1 = 2 * (K - 1);
    constexpr int HALO = W + K - 1;
    static_assert(K <= 24, "run-length shift register holds 24 runs");
    if (len <= 0) return;
    RunWords rw; rw.init(pack, nmask, word_base, len);
    // ---- the first owned step: the first one that STARTS in [s, e) ----
    i32 wi = s >> RunWords::LOG;
    rw.load(wi);
    u32 m = rw.starts & ~((1u << (2 * (s & (RunWords::N - 1)))) - 1);
    i32 p = len;
    for (;;) {
        if (m) { p = wi * RunWords::N + (i32)(__builtin_ctz(m) >> 1); break; }
        ++wi;
        if (wi * RunWords::N >= len) break;
        rw.next(wi); m = rw.starts;
    }
    if (!POS_OWN) {
        if (p >= e || p >= len) {
            // this chunk starts no step; it still owns the end-of-read flush if it holds the last base
            if (!(e == len && s < len)) return;
        }
        s = p;
    }
    const i32 s_pos = s;               // POS_OWN: the owned positions are [s_pos, e)
    // ---- replay start h: HALO steps before it ----
    i32 h = 0;
    if (p > 0) {
        i32 wb = (p - 1) >> RunWords::LOG;
        if (wb != wi || p >= len) rw.load(wb);
        u32 mb = rw.starts;
        if ((p >> RunWords::LOG) == wb) mb &= (1u << (2 * (p & (RunWords::N - 1)))) - 1;
        int need = POS_OWN ? HALO + 1 : HALO;      // (POS_OWN: the step that holds base s may start in front of it)
        for (;;) {
            const int c = __popc(mb);
            if (c >= need) {
                for (int t = 1; t < need; ++t) mb &= ~(1u << (31 - __builtin_clz(mb)));
                h = wb * RunWords::N + ((31 - __builtin_clz(mb)) >> 1);
                break;
            }
            need -= c;
            if (wb == 0) break;                              // fewer than HALO steps in front: from the start of the read
            --wb; rw.load(wb); mb = rw.starts;
        }
    }
    // ---- forward from h ----
    MinWindow<W, u64, u32> win; win.init();
    u64 kf = 0, kr = 0;
    int l = 0, kmer_span = 0;
    // lengths of the last K runs as a byte shift register in VGPRs (a ring indexed at run time would live in scratch memory);
    // byte 0 = newest, byte K-1 = the run that leaves the k-mer at the next push, 0 while fewer than K runs are held
    u32 hq0 = 0, hq1 = 0, hq2 = 0, hq3 = 0, hq4 = 0, hq5 = 0;
    (void)hq3; (void)hq4; (void)hq5;
    wi = h >> RunWords::LOG;
    rw.load(wi);
    m = rw.starts & ~((1u << (2 * (h & (RunWords::N - 1)))) - 1);          // (its lowest bit is h itself)
    i32 ppos = h; u32 pc = rw.code(2 * (h & (RunWords::N - 1)));
    m &= m - 1;
    // next_step: the start of the step behind the pending run (= the end of that run) and its code
    auto next_step = [&](i32 &nxt, u32 &ncode) {
        nxt = len; ncode = 4;
        for (;;) {
            if (m) { const u32 b = (u32)__builtin_ctz(m); nxt = wi * RunWords::N + (i32)(b >> 1); ncode = rw.code(b); m &= m - 1; break; }
            ++wi;
            if (wi * RunWords::N >= len) break;
            rw.next(wi); m = rw.starts;
        }
    };
    // Warm-up (round 5): the first K - 1 steps from h cannot yield a k-mer whatever they hold (l starts at 0 there and a k-mer needs l >= K),
    // so the window sees only "no minimizer" from them -- which leaves a window that holds nothing but "no minimizer" exactly as it is
    // (no emission needs minx != NONE, and mi is only read beside a valid minx).  They run without the hash, the window and the emission
    // sites: the run-length queue, the two k-mers and l alone.  18 of the 23 halo steps of every chunk.
    bool done = false;
#pragma unroll 1
    for (int t = 0; t < K - 1; ++t) {
        i32 nxt; u32 ncode;
        next_step(nxt, ncode);
        if (!POS_OWN && ppos >= e) { done = true; break; }
        if (pc < 4) {
            const i32 run = nxt - ppos;
            const int rl = run > 255 ? 255 : run;
            constexpr int OW = (K - 1) / 4, OB = ((K - 1) % 4) * 8;
            const u32 ow = OW == 0 ? hq0 : OW == 1 ? hq1 : OW == 2 ? hq2 : OW == 3 ? hq3 : OW == 4 ? hq4 : hq5;
            const int oldest = (int)((ow >> OB) & 0xff);
            hq5 = hq5 << 8 | hq4 >> 24; hq4 = hq4 << 8 | hq3 >> 24; hq3 = hq3 << 8 | hq2 >> 24;
            hq2 = hq2 << 8 | hq1 >> 24; hq1 = hq1 << 8 | hq0 >> 24; hq0 = hq0 << 8 | (u32)rl;
            kmer_span += rl - oldest;
            kf = (kf << 2 | pc) & mask;
            kr = (kr >> 2) | (u64)(3 ^ pc) << shift1;
            ++l;
        } else { l = 0; hq0 = hq1 = hq2 = hq3 = hq4 = hq5 = 0; kmer_span = 0; }
        if (nxt >= len) { done = true; break; }
        ppos = nxt; pc = ncode;
    }
    bool at_end = done && !POS_OWN ? false : done;             // POS_OWN: the replay ran into the end of the read
    int past = 0;                                              // POS_OWN: steps taken that start at or beyond e
    for (; !done;) {
        // the next step's start = the end of the pending run
        i32 nxt; u32 ncode;
        next_step(nxt, ncode);
        if (POS_OWN) { if (ppos >= e && ++past > W) break; }    // a minimizer is written at most w steps behind its own
        else if (ppos >= e) break;                             // steps starting at or beyond e belong to later chunks
        u64 ix = ~0ULL; u32 iy = ~0u;
        if (pc < 4) {
            const i32 run = nxt - ppos;
            const i32 i = nxt - 1;                               // last base of the run
            const int rl = run > 255 ? 255 : run;                // spans >= 256 invalidate the k-mer anyway
            {
                constexpr int OW = (K - 1) / 4, OB = ((K - 1) % 4) * 8;      // where byte K-1 sits
                const u32 ow = OW == 0 ? hq0 : OW == 1 ? hq1 : OW == 2 ? hq2 : OW == 3 ? hq3 : OW == 4 ? hq4 : hq5;
                const int oldest = (int)((ow >> OB) & 0xff);
                hq5 = hq5 << 8 | hq4 >> 24; hq4 = hq4 << 8 | hq3 >> 24; hq3 = hq3 << 8 | hq2 >> 24;
                hq2 = hq2 << 8 | hq1 >> 24; hq1 = hq1 << 8 | hq0 >> 24; hq0 = hq0 << 8 | (u32)rl;
                kmer_span += rl - oldest;
            }
            kf = (kf << 2 | pc) & mask;
            kr = (kr >> 2) | (u64)(3 ^ pc) << shift1;
            // K is odd for both presets, so kf == kr (strand-symmetric k-mer) cannot happen
            const u32 z = kf < kr ? 0 : 1;
            ++l; if (l > W + K) l = W + K;                       // only thresholds up to W+K are ever tested
            if (l >= K && kmer_span < 256) { ix = mm_hash64(z ? kr : kf, mask) << 8 | (u64)kmer_span; iy = (u32)i << 1 | z; }
        } else { l = 0; hq0 = hq1 = hq2 = hq3 = hq4 = hq5 = 0; kmer_span = 0; }
        const bool owned = ppos >= s;
        win.template step<K>(ix, iy, l, [&](u64 x, u32 y) {
            if (POS_OWN) { const i32 pb = (i32)(y >> 1); if (pb >= s_pos && pb < e) emit(x, (u64)rid << 32 | (u64)y); }
            else if (owned) emit(x, (u64)rid << 32 | (u64)y);
        });
        if (nxt >= len) { at_end = true; break; }
        ppos = nxt; pc = ncode;
    }
    if (POS_OWN) {             // the final flush belongs to the chunk that holds the minimizer
        if (at_end && win.minx != win.NONE) { const i32 pb = (i32)(win.miny >> 1); if (pb >= s_pos && pb < e) emit(win.minx, (u64)rid << 32 | (u64)win.miny); }
    } else if (e == len && win.minx != win.NONE) emit(win.minx, (u64)rid << 32 | (u64)win.miny);  // final flush by the last chunk
}

// Runs the state machine over one chunk.  Emission callback is only invoked for owned steps.
template <int K, int W, bool HPC, bool POS_OWN = false, typename Emit>
__device__ __forceinline__ void sketch_chunk(const u64 *pack, const u32 *nmask, u64 word_base, i32 len, u32 rid,
                                             i32 s, i32 e, Emit &&emit) {
    static_assert(HPC || !POS_OWN, "position ownership exists for the HPC form only (k_sketch_redo)");
    if constexpr (HPC) { sketch_chunk_hpc<K, W, POS_OWN>(pack, nmask, word_base, len, rid, s, e, emit); return; }
    constexpr u64 mask = (1ULL << (2 * K)) - 1;
    constexpr int shift1 = 2 * (K - 1);
    constexpr int HALO = W + K - 1;
    constexpr bool NARROW = 2 * K <= 32;
    using XT = typename std::conditional<NARROW, u32, u64>::type;
    BaseReader rd; rd.init(pack, nmask, word_base);
    MinWindow<W, XT, u32> win; win.init();          // y = pos << 1 | strand; the read id is constant inside a read
    XT kf = 0, kr = 0;
    // narrow values -> the (x, y) pair the callers expect
    auto emit_xy = [&](XT x, u32 y) {
        if (NARROW) emit((u64)x << 8 | (u64)K, (u64)rid << 32 | (u64)y);      // (the span is always k for a valid k-mer)
        else emit((u64)x, (u64)rid << 32 | (u64)y);
    };
    int l = 0;
    i32 h = s - HALO; if (h < 0) h = 0;
    i32 i = h;
    while (i < len) {
        if (i >= e) break;  // steps starting at or beyond e belong to later chunks
        const i32 step_start = i;
        const u32 c = rd.get(i);
        XT ix = (XT)~(XT)0; u32 iy = ~0u;
        if (c < 4) {
            const int kmer_span = l + 1 < K ? l + 1 : K;
            kf = (XT)((kf << 2 | c) & (XT)mask);
            kr = (XT)((kr >> 2) | (XT)(3 ^ c) << shift1);
            // K is odd for both presets, so kf == kr (strand-symmetric k-mer) cannot happen
            const u32 z = kf < kr ? 0 : 1;
            ++l; if (l > W + K) l = W + K;  // only thresholds up to W+K are ever tested
            if (l >= K) {
                if (NARROW) ix = (XT)mm_hash32((u32)(z ? kr : kf), (u32)mask);   // (kmer_span == K here)
                else ix = (XT)(mm_hash64(z ? kr : kf, mask) << 8 | (u64)kmer_span);
                iy = (u32)i << 1 | z;
            }
        } else l = 0;
        const bool owned = step_start >= s;
        win.template step<K>(ix, iy, l, [&](XT x, u32 y) { if (owned) emit_xy(x, y); });
        ++i;
    }
    if (e == len && win.minx != win.NONE) emit_xy(win.minx, win.miny);  // final flush by the last chunk
}

// HPC clamps run lengths to 255 in the queue; a clamped run makes kmer_span >= 255+... only when
// the true span is >= 256 too, except the single case span == 255 exactly built from one 255-run and
// K-1 ... (impossible: K-1 >= 14 further runs add >= 14).  So `kmer_span < 256` is decided identically.

struct ChunkMap {  // chunk id -> (read, first base)
    const u32 *chunk_start;  // [n_reads+1] prefix of ceil(len/SK_CHUNK)
    u32 n_reads;
    __device__ __forceinline__ u32 find(u32 c) const {
        u32 lo = 0, hi = n_reads;
        while (hi - lo > 1) { u32 mid = (lo + hi) >> 1; if (chunk_start[mid] <= c) lo = mid; else hi = mid; }
        return lo;
    }
};

template <int K, int W, bool HPC>
__global__ __launch_bounds__(SK_THREADS) void k_sketch_count(const u64 *__restrict__ pack, const u32 *__restrict__ nmask,
                                                            const u64 *__restrict__ woff, const u32 *__restrict__ lens,
                                                            ChunkMap cm, u32 n_chunks, u32 *__restrict__ counts) {
    u32 c = blockIdx.x * SK_THREADS + threadIdx.x;
    if (c >= n_chunks) return;
    u32 r = cm.find(c);
    i32 len = (i32)lens[r];
    i32 s = (i32)(c - cm.chunk_start[r]) * SK_CHUNK;
    i32 e = s + SK_CHUNK < len ? s + SK_CHUNK : len;
    u32 n = 0;
    sketch_chunk<K, W, HPC>(pack, nmask, woff[r], len, r, s, e, [&](u64, u64) { ++n; });
    counts[c] = n;
}

// PK (index only): one packed u64 per minimizer, hash << (pk_rid_bits + pk_pos1) | rid << pk_pos1 | (pos << 1 | strand),
// written to out_x; out_y is not touched.  8 bytes per index entry instead of 16 through the sort and the lookups.
//
// Writes chunk c's minimizers to out[o0 ...) and returns how many it found; only the first `cap` are stored.
// A lane's minimizers are consecutive in the output, but it produces one every ~3 steps: written one by
// one, every 32-byte sector reaches HBM as four partial writes (measured 4x the bytes).  The last <= 4 are
// kept in registers and leave as one sector-aligned 32-byte run per array whenever the output index
// reaches a multiple of 4; only the head and the tail of the lane's range are written singly.
// PK == 2 (index only, "SEGW": round 5): the entry as the segment-packed index sort wants it (k_prims.h, index_sort_segw) -- out_x gets
// one word, [the low 2k - 16 bits of the hash's significance string | rid | pos << 1 | strand] (pk_ybits = the width of the y field),
// and out_y, read as a u32 ARRAY, the top 16 bits of that string: the two digits the sort's first two passes order by.  12 bytes per
// entry instead of the pair's 16 through the slots, the compaction and those two passes.
template <int K, int W, bool HPC, bool INDEX_KEYS, int PK, bool POS_OWN = false>
__device__ __forceinline__ u32 sketch_write_chunk(const u64 *__restrict__ pack, const u32 *__restrict__ nmask, u64 word_base, i32 len, u32 r,
                                                  i32 s, i32 e, u32 o0, u32 cap, u64 *__restrict__ out_x, u64 *__restrict__ out_y,
                                                  u32 pk_pos1, u32 pk_ybits) {
    static_assert(PK != 2 || (INDEX_KEYS && 2 * K > 16), "SEGW entries are index entries of a hash wider than its two sort digits");
    u32 o = o0, found = 0;
    u64 bx0 = 0, bx1 = 0, bx2 = 0, bx3 = 0, by0 = 0, by1 = 0, by2 = 0, by3 = 0;
    u32 *const out_d = (u32 *)out_y;                   // (PK == 2)
    u32 nb = 0;   // buffered entries: output indices [o - nb, o), newest in b?3
    auto flush_tail = [&]() {
        if (nb >= 3) { out_x[o - 3] = bx1; if (PK == 0) out_y[o - 3] = by1; else if (PK == 2) out_d[o - 3] = (u32)by1; }
        if (nb >= 2) { out_x[o - 2] = bx2; if (PK == 0) out_y[o - 2] = by2; else if (PK == 2) out_d[o - 2] = (u32)by2; }
        if (nb >= 1) { out_x[o - 1] = bx3; if (PK == 0) out_y[o - 1] = by3; else if (PK == 2) out_d[o - 1] = (u32)by3; }
        nb = 0;
    };
    sketch_chunk<K, W, HPC, POS_OWN>(pack, nmask, word_base, len, r, s, e, [&](u64 x, u64 y) {
        if (found++ >= cap) return;                     // counted, not stored (the caller notices found > cap)
        bx0 = bx1; bx1 = bx2; bx2 = bx3;
        by0 = by1; by1 = by2; by2 = by3;
        if (PK == 1) bx3 = (x >> 8) << pk_ybits | (y >> 32) << pk_pos1 | (u64)(u32)y;
        else if (PK == 2) {
            const u64 S = hash_to_sig(x >> 8, 2 * K);
            bx3 = (S & ((1ULL << (2 * K - 16)) - 1)) << pk_ybits | (y >> 32) << pk_pos1 | (u64)(u32)y;
            by3 = S >> (2 * K - 16);
        } else bx3 = INDEX_KEYS ? (x >> 8) : x;  // the index keeps only the hash, queries keep hash<<8|span
        if (PK != 2) by3 = y;
        ++o; ++nb;
        if ((o & 3u) == 0) {
            if (nb == 4) {
                ulonglong2 a, b;
                a.x = bx0; a.y = bx1; b.x = bx2; b.y = bx3;
                *(ulonglong2 *)(out_x + o - 4) = a; *(ulonglong2 *)(out_x + o - 2) = b;
                if (PK == 0) {
                    a.x = by0; a.y = by1; b.x = by2; b.y = by3;
                    *(ulonglong2 *)(out_y + o - 4) = a; *(ulonglong2 *)(out_y + o - 2) = b;
                } else if (PK == 2) {
                    uint4 d4; d4.x = (u32)by0; d4.y = (u32)by1; d4.z = (u32)by2; d4.w = (u32)by3;
                    *(uint4 *)(out_d + o - 4) = d4;
                }
                nb = 0;
            } else flush_tail();
        }
    });
    flush_tail();
    return found;
}

// second pass of the two-pass form: offsets from the scanned counts of k_sketch_count
template <int K, int W, bool HPC, bool INDEX_KEYS, int PK>
__global__ __launch_bounds__(SK_THREADS) void k_sketch_write(const u64 *__restrict__ pack, const u32 *__restrict__ nmask,
                                                            const u64 *__restrict__ woff, const u32 *__restrict__ lens,
                                                            ChunkMap cm, u32 n_chunks, const u32 *__restrict__ offs,
                                                            u64 *__restrict__ out_x, u64 *__restrict__ out_y, u32 pk_pos1, u32 pk_ybits) {
    u32 c = blockIdx.x * SK_THREADS + threadIdx.x;
    if (c >= n_chunks) return;
    u32 r = cm.find(c);
    i32 len = (i32)lens[r];
    i32 s = (i32)(c - cm.chunk_start[r]) * SK_CHUNK;
    i32 e = s + SK_CHUNK < len ? s + SK_CHUNK : len;
    (void)sketch_write_chunk<K, W, HPC, INDEX_KEYS, PK>(pack, nmask, woff[r], len, r, s, e, offs[c], 0xFFFFFFFFu, out_x, out_y, pk_pos1, pk_ybits);
}

// One-pass form: every chunk writes into its own slot of SK_CAP entries (tmp[c * SK_CAP ...)) and reports its count;
// k_sketch_compact then closes the gaps.  The state machine -- the hash, the window -- runs once per base instead of
// twice (count pass + write pass), at the price of one extra streaming copy of the minimizers.  A chunk with more than
// SK_CAP minimizers (possible in principle: a step can emit up to w of them) raises *overflow and the caller falls
// back to the two-pass form.
#define SK_CAP (SK_CHUNK + 8)
template <int K, int W, bool HPC, bool INDEX_KEYS, int PK>
__global__ __launch_bounds__(SK_THREADS) void k_sketch_direct(const u64 *__restrict__ pack, const u32 *__restrict__ nmask,
                                                             const u64 *__restrict__ woff, const u32 *__restrict__ lens,
                                                             ChunkMap cm, u32 n_chunks, u32 *__restrict__ counts, u32 *__restrict__ overflow,
                                                             u64 *__restrict__ tmp_x, u64 *__restrict__ tmp_y, u32 pk_pos1, u32 pk_ybits,
                                                             u32 cap /* <= SK_CAP; smaller only in tests */, u32 c_base = 0) {
    // chunks [c_base, n_chunks): a set whose upload is still in flight is sketched range by range, each behind the chunk of the
    // packed image it needs (host_sketch.inl, sketch_launch)
    u32 c = c_base + blockIdx.x * SK_THREADS + threadIdx.x;
    if (c >= n_chunks) return;
    u32 r = cm.find(c);
    i32 len = (i32)lens[r];
    i32 s = (i32)(c - cm.chunk_start[r]) * SK_CHUNK;
    i32 e = s + SK_CHUNK < len ? s + SK_CHUNK : len;
    const u64 base = (u64)c * SK_CAP;
    const u32 found = sketch_write_chunk<K, W, HPC, INDEX_KEYS, PK>(pack, nmask, woff[r], len, r, s, e, 0u, cap, tmp_x + base,
                                                                    PK == 1 ? tmp_y : PK == 2 ? (u64 *)((u32 *)tmp_y + base) : tmp_y + base, pk_pos1, pk_ybits);
    counts[c] = found;
    if (found > cap) *overflow = 1u;
}

// WAVE-DENSE form of the one-pass index sketch (round 6).  k_sketch_direct gives every chunk a slot of SK_CAP entries, a quarter to a third
// full, and k_sketch_compact closes the gaps: one more read and write of every entry (45 ms of an H. sapiens-scale step).  Here a
// WAVEFRONT owns a slot -- room for `cap` entries behind its 64 chunks -- and fills it densely: whenever some of its lanes reach an
// emission site of the state machine together, one ballot ranks them, the wavefront's counter (LDS, wavefront-scope atomics: lanes that sit at other
// program points must see every update) moves on by their number, and they write consecutive entries -- 8-byte words (and, PK == 2, the
// 16-bit digit pair) in runs that the L2 combines into whole lines.  No gaps inside a slot, `wave_cnt[wave]` entries in it: the index
// sort's first pass reads the slots through its slot-source form (k_prims.h: SlotSrc, rs_for_slot_items) and no compaction runs.
// The entries of a slot come in the order the lanes EMIT them, not in read order -- inside 8 192 bases.  The index does not care: the
// sort is by hash, a key's position list is only ever expanded into anchors that are sorted by position again (k_seed.h), and
// lrge_hip_index_dump orders its output by (hash, y) itself.  A wavefront that finds more than `cap` entries (never on real reads:
// cap is 1.2-1.25x the mean of 2 040 / 2 780 per 8 192 bases with a deviation of ~50; low-complexity input can) raises *overflow and
// the caller takes the slot-per-chunk path.
template <int K, int W, bool HPC, int PK>
__global__ __launch_bounds__(SK_THREADS) void k_sketch_wave(const u64 *__restrict__ pack, const u32 *__restrict__ nmask,
                                                           const u64 *__restrict__ woff, const u32 *__restrict__ lens,
                                                           ChunkMap cm, u32 n_chunks, u32 *__restrict__ wave_cnt, u32 *__restrict__ overflow,
                                                           u64 *__restrict__ out_x, wdig_t *__restrict__ out_d, u32 pk_pos1, u32 pk_ybits,
                                                           u32 cap, u32 c_base /* a multiple of 64 */) {
    static_assert(PK == 1 || PK == 2, "index entries only: packed words, or SEGW words + digit pairs");
    __shared__ u32 s_cnt[SK_THREADS / 64];
    const u32 w = threadIdx.x >> 6, lane = lane_id();
    const u32 c0 = c_base + blockIdx.x * SK_THREADS + 64 * w;            // the wavefront's first chunk
    if (c0 >= n_chunks) return;                                          // (wave-uniform)
    // the counter is read and written through wavefront-scope atomics ON THE LDS ARRAY ITSELF: a volatile generic pointer to it compiles to
    // flat loads / stores with system-scope bits and a wait for every global store of the wavefront in front of each of them
    auto cnt_get = [&]() -> u32 { return __hip_atomic_load(&s_cnt[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); };
    auto cnt_set = [&](u32 v) { __hip_atomic_store(&s_cnt[w], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); };
    if (lane == 0) cnt_set(0);                                           // (same wavefront, in-order LDS: no barrier)
    const u64 sbase = (u64)(c0 >> 6) * cap;
    const u32 c = c0 + lane;
    if (c < n_chunks) {
        const u32 r = cm.find(c);
        const i32 len = (i32)lens[r];
        const i32 s = (i32)(c - cm.chunk_start[r]) * SK_CHUNK;
        const i32 e = s + SK_CHUNK < len ? s + SK_CHUNK : len;
        sketch_chunk<K, W, HPC>(pack, nmask, woff[r], len, r, s, e, [&](u64 x, u64 y) {
            const u64 m = __ballot(1);                                   // the lanes at this emission site right now
            const u32 b = cnt_get();                                     // (one broadcast LDS read)
            if (lane == (u32)__builtin_ctzll(m)) cnt_set(b + (u32)__popcll(m));
            const u32 idx = b + (u32)__popcll(m & lanemask_lt());
            if (idx < cap) {
                if (PK == 1) out_x[sbase + idx] = (x >> 8) << pk_ybits | (y >> 32) << pk_pos1 | (u64)(u32)y;
                else {
                    const u64 S = hash_to_sig(x >> 8, 2 * K);
                    out_x[sbase + idx] = (S & ((1ULL << (2 * K - 16)) - 1)) << pk_ybits | (y >> 32) << pk_pos1 | (u64)(u32)y;
                    out_d[sbase + idx] = (wdig_t)(S >> (2 * K - 16));
                }
            }
        });
    }
    if (lane == 0) { const u32 t = cnt_get(); wave_cnt[c0 >> 6] = t; if (t > cap) *overflow = 1u; }
}

// Behind k_sketch_tile (HPC): the chunks it marked ST_REDO, the sequential way, with the tile form's attribution (POS_OWN above).
template <int K, int W, bool INDEX_KEYS, int PK>
__global__ __launch_bounds__(SK_THREADS) void k_sketch_redo(const u64 *__restrict__ pack, const u32 *__restrict__ nmask,
                                                           const u64 *__restrict__ woff, const u32 *__restrict__ lens,
                                                           ChunkMap cm, u32 n_chunks, u32 *__restrict__ counts, u32 *__restrict__ overflow,
                                                           u64 *__restrict__ tmp_x, u64 *__restrict__ tmp_y, u32 pk_pos1, u32 pk_ybits,
                                                           u32 cap, u32 c_base) {
    const u32 c = c_base + blockIdx.x * SK_THREADS + threadIdx.x;
    if (c >= n_chunks) return;
    if (counts[c] != ST_REDO) return;
    const u32 r = cm.find(c);
    const i32 len = (i32)lens[r];
    const i32 s = (i32)(c - cm.chunk_start[r]) * SK_CHUNK;
    const i32 e = s + SK_CHUNK < len ? s + SK_CHUNK : len;
    const u64 base = (u64)c * SK_CAP;
    const u32 found = sketch_write_chunk<K, W, true, INDEX_KEYS, PK, true>(pack, nmask, woff[r], len, r, s, e, 0u, cap, tmp_x + base,
                                                                          PK == 1 ? tmp_y : PK == 2 ? (u64 *)((u32 *)tmp_y + base) : tmp_y + base, pk_pos1, pk_ybits);
    counts[c] = found;
    if (found > cap) *overflow = 1u;
}

// One wavefront per 64 consecutive chunks: their minimizers form one contiguous output range, which the lanes walk 64
// entries at a time (the chunk of an entry by a 6-step search through the 64 scanned counts), so the writes are
// consecutive across the wave and the reads touch the fronts of two or three slots.
// c_base / out_cap / ovf: the ranged form (host_sketch.inl: a set whose slots do not fit at once is sketched range by range into the
// same slots): the launch covers chunks [c_base, n_chunks), `*d_total` is the output offset behind the range, and an entry that
// would land at or beyond out_cap raises *ovf instead (the caller's estimate of the output size was too small: it starts over).
// PAIRS: 0 = x only, 1 = (x, y) of 8 bytes each, 2 = x + a u32 per entry (tmp_y / out_y read as u32 arrays: the SEGW entries)
template <int PAIRS>
__global__ __launch_bounds__(256) void k_sketch_compact(const u64 *__restrict__ tmp_x, const u64 *__restrict__ tmp_y,
                                                        const u32 *__restrict__ offs, const u32 *__restrict__ d_total, u32 n_chunks,
                                                        u64 *__restrict__ out_x, u64 *__restrict__ out_y, u32 c_base = 0, u32 out_cap = 0xFFFFFFFFu,
                                                        u32 *__restrict__ ovf = nullptr) {
    __shared__ u32 so[4][65];
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 c0 = c_base + (blockIdx.x * 4 + w) * 64;
    const u32 c = c0 + lane;
    if (c0 < n_chunks) {
        so[w][lane] = c < n_chunks ? offs[c] : *d_total;
        if (lane == 63) so[w][64] = c0 + 64 < n_chunks ? offs[c0 + 64] : *d_total;
    }
    __syncthreads();
    if (c0 >= n_chunks) return;
    const u32 lo = so[w][0], hi = so[w][64];
    for (u32 o = lo + lane; o < hi; o += 64) {
        u32 j = 0;                                       // last chunk with offs <= o
#pragma unroll
        for (int st = 32; st > 0; st >>= 1) if (so[w][j + st] <= o) j += st;
        const u32 within = o - so[w][j];
        if (within >= SK_CAP) continue;                  // a chunk that overflowed its slot (the caller discards this output)
        if (o >= out_cap) { if (ovf) *ovf = 1u; continue; }
        const u64 src = (u64)(c0 + j) * SK_CAP + within;
        out_x[o] = tmp_x[src];
        if (PAIRS == 1) out_y[o] = tmp_y[src];
        else if (PAIRS == 2) ((u32 *)out_y)[o] = ((const u32 *)tmp_y)[src];
    }
}

// ranged sketch: offs[i] += *base for the range's chunks ...
__global__ __launch_bounds__(256) void k_add_base_u32(u32 *__restrict__ a, u32 n, const u32 *__restrict__ base) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += *base;
}
// ... and then *run += *add (one thread, behind it on the stream); the sum is clamped so that a 32-bit wrap shows as an overflow
__global__ void k_bump_u32(u32 *__restrict__ run, const u32 *__restrict__ add, u32 *__restrict__ ovf) {
    const u64 s = (u64)*run + *add;
    if (s > 0xFFFFFFFFull) { *ovf = 1u; *run = 0xFFFFFFFFu; } else *run = (u32)s;
}

// per-read minimizer offsets from per-chunk offsets: mz_off[r] = offs[chunk_start[r]]
__global__ void k_read_mz_offsets(const u32 *__restrict__ chunk_start, const u32 *__restrict__ offs, u32 n_reads,
                                  u32 n_chunks, const u32 *__restrict__ d_total, u32 *__restrict__ mz_off) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_reads) return;
    u32 c = chunk_start[r];
    mz_off[r] = (r == n_reads || c >= n_chunks) ? *d_total : offs[c];
}
