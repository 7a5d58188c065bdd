// sort_bench.hip -- the index sort of csrc/k_prims.h alone, in its three forms: the LSD passes (radix_sort_keys), the hybrid form
// (index_sort_hybrid: two most-significant-digit passes, the rest inside LDS) and the hybrid form with tiny LDS classes (every
// sub-bucket takes the segmented global passes).  Exact against std::stable_sort on the small inputs, order + permutation
// checks and timing at index scale.  Development harness, not part of the product or the tests.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/micro/_bin/sort_bench tools/micro/sort_bench.hip && tools/micro/_bin/sort_bench [n]
#include "sort_forms.h"      // (includes ../../lrge_amd/csrc/k_prims.h and the retired forms it was split from in round 4)

#include <algorithm>
#include <chrono>
#include <random>

static int g_db = 8;                                          // digit bits of the form under test
static u64 order_key(u64 k, int begin_bit, int nbits) {      // the key the reversed-digit LSD sort orders by
    const int passes = (nbits + g_db - 1) / g_db;
    u64 f = 0;
    for (int d = 0; d < passes; ++d) {
        const int w = nbits - d * g_db >= g_db ? g_db : nbits - d * g_db;
        f = (f << w) | ((k >> (begin_bit + d * g_db)) & ((1ULL << w) - 1));
    }
    return f;
}

__global__ void k_fill(u64 *k, u64 n, int ybits, int hbits, u64 seed) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 x = (i + seed) * 0x9E3779B97F4A7C15ULL; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
    // a third of the keys repeat (real minimizers), so that stability matters
    if ((x & 3) == 0) x = (x >> 8) % 1000003;
    k[i] = ((x & ((1ULL << hbits) - 1)) << ybits) | i;        // payload = original position: stable <=> ascending inside a key
}
__global__ void k_check(const u64 *k, u64 n, int ybits, int hbits, unsigned long long *bad, unsigned long long *sum, int db) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    atomicAdd(sum, (unsigned long long)(k[i] * 0x9E3779B97F4A7C15ULL >> 40));
    if (i == 0) return;
    const int passes = (hbits + db - 1) / db;
    u64 fa = 0, fb = 0;
    for (int d = 0; d < passes; ++d) {
        const int w = hbits - d * db >= db ? db : hbits - d * db;
        fa = (fa << w) | ((k[i - 1] >> (ybits + d * db)) & ((1ULL << w) - 1));
        fb = (fb << w) | ((k[i] >> (ybits + d * db)) & ((1ULL << w) - 1));
    }
    const u64 ya = k[i - 1] & ((1ULL << ybits) - 1), yb = k[i] & ((1ULL << ybits) - 1);
    if (fa > fb || (fa == fb && ya >= yb)) atomicAdd(bad, 1ULL);
}

int main(int argc, char **argv) {
    const u64 n_big = argc > 1 ? strtoull(argv[1], nullptr, 10) : 242000000ULL;
    const int ybits = 34, hbits = argc > 2 ? atoi(argv[2]) : 30;
    lrge_hip_ctx ctx; memset(ctx.ms, 0, sizeof ctx.ms); memset(ctx.counters, 0, sizeof ctx.counters);
    ctx.timer_level = 0;
    if (hipStreamCreate(&ctx.stream) != hipSuccess) { fprintf(stderr, "no device\n"); return 2; }
    int rc_all = 0;
    ctx.lsort_ok[0] = true;
    ctx.lsort_ok[1] = hipFuncSetAttribute((const void *)k_seg_sort_keys<512, 16, LSORT_DB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSORT_BYTES(512, 16, LSORT_DB)) == hipSuccess;
    ctx.lsort_ok[2] = hipFuncSetAttribute((const void *)k_seg_sort_keys<1024, 16, LSORT_DB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSORT_BYTES(1024, 16, LSORT_DB)) == hipSuccess;
    ctx.opts["HYBRID_SORT_MIN"] = "2"; ctx.opts["VERBOSE"] = "1";
    for (int form = 0; form < 5; ++form)
    for (u64 n : {(u64)1, (u64)4095, (u64)4096, (u64)4097, (u64)1000003, (u64)1048576 + 77, n_big}) {
        if (form == 2 && n > 2000000) continue;
        g_db = form == 3 ? 10 : 8;
        if (form == 2) { ctx.opts["DEBUG_LSORT_CAP0"] = "3"; ctx.opts["DEBUG_LSORT_CAP1"] = "5"; ctx.opts["DEBUG_LSORT_CAP2"] = "9"; }
        Scratch sc(&ctx);
        u64 *k0 = sc.get<u64>(n), *k1 = sc.get<u64>(n);
        unsigned long long *d_chk = sc.get<unsigned long long>(4);
        if (!k0 || !k1 || !d_chk) { fprintf(stderr, "alloc failed\n"); return 2; }
        hipLaunchKernelGGL(k_fill, dim3((u32)div_up(n, 256)), dim3(256), 0, ctx.stream, k0, n, ybits, hbits, 12345ULL);
        (void)hipMemsetAsync(d_chk, 0, 32, ctx.stream);
        hipLaunchKernelGGL(k_check, dim3((u32)div_up(n, 256)), dim3(256), 0, ctx.stream, k0, n, ybits, hbits, d_chk, d_chk + 1, g_db);
        std::vector<u64> h_in;
        if (n <= 2000000) { h_in.resize(n); (void)hipMemcpy(h_in.data(), k0, n * 8, hipMemcpyDeviceToHost); }
        u64 *res = nullptr;
        float best = 1e9f;
        const int reps = n > 2000000 ? 4 : 1;
        for (int r = 0; r < reps; ++r) {
            if (r) hipLaunchKernelGGL(k_fill, dim3((u32)div_up(n, 256)), dim3(256), 0, ctx.stream, k0, n, ybits, hbits, 12345ULL);
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            (void)hipEventRecord(a, ctx.stream);
            int rc; bool hyb = false;
            if (form == 0) rc = radix_sort_keys(&ctx, sc, k0, k1, n, ybits, hbits, &res, /*reverse_digits=*/true);
            else if (form == 3) rc = radix_sort_keys_db10(&ctx, sc, k0, k1, n, ybits, hbits, &res, /*reverse_digits=*/true);
            else if (form == 4) {
                (void)hipMemsetAsync(d_chk + 3, 0, 8, ctx.stream);       // (d_chk[3] is rewritten by the check below)
                rc = radix_sort_keys_onesweep(&ctx, sc, k0, k1, n, ybits, hbits, &res, true, (u32 *)(d_chk + 3), &hyb);
                if (!rc && !hyb) rc = radix_sort_keys(&ctx, sc, k0, k1, n, ybits, hbits, &res, true);
                if (!rc) { unsigned long long e = 0; (void)hipMemcpyAsync(&e, d_chk + 3, 8, hipMemcpyDeviceToHost, ctx.stream); (void)hipStreamSynchronize(ctx.stream); if (e) { fprintf(stderr, "one-sweep look-back gave up\n"); return 1; } (void)hipMemsetAsync(d_chk + 3, 0, 8, ctx.stream); }
            }
            else { rc = index_sort_hybrid(&ctx, sc, k0, k1, n, ybits, hbits, &res, &hyb); if (!rc && !hyb) rc = radix_sort_keys(&ctx, sc, k0, k1, n, ybits, hbits, &res, true); }
            (void)hipEventRecord(b, ctx.stream);
            if (rc || hipStreamSynchronize(ctx.stream) != hipSuccess) { fprintf(stderr, "sort failed: %s / %s\n", ctx.err.c_str(), hipGetErrorString(hipGetLastError())); return 1; }
            float ms = 0; (void)hipEventElapsedTime(&ms, a, b); best = std::min(best, ms);
            if (res != k0 && r + 1 < reps) {}      // (refilled into k0 every round)
        }
        hipLaunchKernelGGL(k_check, dim3((u32)div_up(n, 256)), dim3(256), 0, ctx.stream, res, n, ybits, hbits, d_chk + 2, d_chk + 3, g_db);
        unsigned long long h[4];
        (void)hipMemcpy(h, d_chk, 32, hipMemcpyDeviceToHost);
        bool ok = h[2] == 0 && h[1] == h[3];
        if (!h_in.empty()) {
            std::stable_sort(h_in.begin(), h_in.end(), [&](u64 a, u64 b) { return order_key(a, ybits, hbits) < order_key(b, ybits, hbits); });
            std::vector<u64> h_out(n); (void)hipMemcpy(h_out.data(), res, n * 8, hipMemcpyDeviceToHost);
            ok = ok && h_out == h_in;
        }
        printf("%s n = %llu: %s  (order violations %llu, checksum %s)  %.3f ms  %.2f G keys/s\n", form == 0 ? "LSD   " : form == 1 ? "hybrid" : form == 2 ? "hybrid, tiny LDS classes" : form == 3 ? "LSD, 10-bit digits" : "one-sweep", (unsigned long long)n, ok ? "ok" : "WRONG", h[2],
               h[1] == h[3] ? "equal" : "DIFFERENT", best, n / best * 1e-6);
        (void)0;
        if (!ok) rc_all = 1;
    }
    {   // one pass of the two digit widths taken apart: histogram, scan, scatter
        Scratch sc(&ctx);
        const u64 n = n_big; const u32 nb = (u32)div_up(n, RS_TILE);
        u64 *k0 = sc.get<u64>(n), *k1 = sc.get<u64>(n); u32 *hist = sc.get<u32>((u64)1024 * nb);
        hipLaunchKernelGGL(k_fill, dim3((u32)div_up(n, 256)), dim3(256), 0, ctx.stream, k0, n, ybits, hbits, 12345ULL);
        hipEvent_t e[4]; for (auto &x : e) (void)hipEventCreate(&x);
        for (int db : {8, 9, 10, 8, 9, 10}) {
            const u32 dm = (1u << db) - 1;
            (void)hipEventRecord(e[0], ctx.stream);
            if (db == 8) hipLaunchKernelGGL((k_rs_hist<false, 8>), dim3(nb), dim3(RS_THREADS), 0, ctx.stream, k0, n, ybits, nb, hist, (const SegTile *)nullptr, dm);
            else if (db == 9) hipLaunchKernelGGL((k_rs_hist<false, 9>), dim3(nb), dim3(RS_THREADS), 0, ctx.stream, k0, n, ybits, nb, hist, (const SegTile *)nullptr, dm);
            else hipLaunchKernelGGL((k_rs_hist<false, 10>), dim3(nb), dim3(RS_THREADS), 0, ctx.stream, k0, n, ybits, nb, hist, (const SegTile *)nullptr, dm);
            (void)hipEventRecord(e[1], ctx.stream);
            (void)scan_exclusive_u32(&ctx, sc, hist, hist, ((u64)1 << db) * nb, nullptr);
            (void)hipEventRecord(e[2], ctx.stream);
            UnpackParams up{0, 0, 0, dm};
            if (db == 8) hipLaunchKernelGGL((k_rs_scatter<false, RS_MODE_KEYS, 8>), dim3(nb), dim3(RS_THREADS), 0, ctx.stream, k0, (const u64 *)nullptr, k1, (u64 *)nullptr, n, ybits, nb, hist, (const SegTile *)nullptr, up);
            else if (db == 9) hipLaunchKernelGGL((k_rs_scatter<false, RS_MODE_KEYS, 9>), dim3(nb), dim3(RS_THREADS), 0, ctx.stream, k0, (const u64 *)nullptr, k1, (u64 *)nullptr, n, ybits, nb, hist, (const SegTile *)nullptr, up);
            else hipLaunchKernelGGL((k_rs_scatter<false, RS_MODE_KEYS, 10>), dim3(nb), dim3(RS_THREADS), 0, ctx.stream, k0, (const u64 *)nullptr, k1, (u64 *)nullptr, n, ybits, nb, hist, (const SegTile *)nullptr, up);
            (void)hipEventRecord(e[3], ctx.stream);
            (void)hipStreamSynchronize(ctx.stream);
            float a, b, c; (void)hipEventElapsedTime(&a, e[0], e[1]); (void)hipEventElapsedTime(&b, e[1], e[2]); (void)hipEventElapsedTime(&c, e[2], e[3]);
            printf("one pass, %2d-bit digits: hist %.3f ms  scan %.3f ms  scatter %.3f ms\n", db, a, b, c);
        }
    }
    {   // the one-sweep form taken apart: all histograms, one pass
        Scratch sc(&ctx);
        const u64 n = n_big; const u32 nb = (u32)div_up(n, RS_TILE);
        u64 *k0 = sc.get<u64>(n), *k1 = sc.get<u64>(n); u32 *gh = sc.get<u32>(4 * 256 + 16), *state = sc.get<u32>((u64)256 * nb);
        hipLaunchKernelGGL(k_fill, dim3((u32)div_up(n, 256)), dim3(256), 0, ctx.stream, k0, n, ybits, hbits, 12345ULL);
        OsPasses P; P.n = 4; for (int p = 0; p < 4; ++p) { P.shift[p] = ybits + (3 - p) * 8; P.dmask[p] = p == 0 ? 63u : 255u; }
        hipEvent_t e[5]; for (auto &x : e) (void)hipEventCreate(&x);
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipMemsetAsync(gh, 0, (4 * 256 + 16) * 4, ctx.stream);
            (void)hipEventRecord(e[0], ctx.stream);
            hipLaunchKernelGGL(k_rs_hist_all, dim3(std::min<u32>(nb, 2048)), dim3(RS_THREADS), 0, ctx.stream, k0, n, P, gh);
            (void)hipEventRecord(e[1], ctx.stream);
            hipLaunchKernelGGL(k_rs_gscan, dim3(1), dim3(256), 0, ctx.stream, gh, 4);
            (void)hipEventRecord(e[2], ctx.stream);
            (void)hipMemsetAsync(state, 0, (u64)256 * nb * 4, ctx.stream);
            (void)hipEventRecord(e[3], ctx.stream);
            hipLaunchKernelGGL(k_rs_onesweep, dim3(nb), dim3(RS_THREADS), 0, ctx.stream, k0, k1, n, P.shift[1], P.dmask[1], gh + 256, state, nb, (u32)rep + 3, gh + 1024, gh + 1032);
            (void)hipEventRecord(e[4], ctx.stream);
            (void)hipStreamSynchronize(ctx.stream);
            float a, b, c, d; (void)hipEventElapsedTime(&a, e[0], e[1]); (void)hipEventElapsedTime(&b, e[1], e[2]); (void)hipEventElapsedTime(&c, e[2], e[3]); (void)hipEventElapsedTime(&d, e[3], e[4]);
            printf("one-sweep parts (groups of %d tiles): all histograms %.3f ms  digit scan %.3f ms  state memset %.3f ms  one pass %.3f ms\n", 8 << rep, a, b, c, d);
        }
    }
    ctx.pool.destroy();
    return rc_all;
}
