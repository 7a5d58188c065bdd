// scan_bench.hip -- scan_exclusive_u32 (csrc/k_prims.h) alone: exact against a host scan, timed at the sizes the overlap step uses.
// Measured (round 2): the three-kernel form moves 121 M counters in 0.283 ms (5.1 TB/s over its 12 bytes per item); a single-pass
// decoupled look-back scan (ticketed 4096-item tiles, one 64-bit state word per tile, agent-scope atomics) was exact but took
// 0.441 ms -- the cross-XCD round trips of the look-back cost more than the second read saves -- and was not kept.
#include "../../lrge_amd/csrc/k_prims.h"
#include <numeric>

__global__ void k_fill32(u32 *d, u64 n, u64 seed) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 x = (i + seed) * 0x9E3779B97F4A7C15ULL; x ^= x >> 31;
    d[i] = (u32)(x & 15);
}

int main() {
    lrge_hip_ctx ctx; memset(ctx.ms, 0, sizeof ctx.ms); memset(ctx.counters, 0, sizeof ctx.counters);
    ctx.timer_level = 0;
    if (hipStreamCreate(&ctx.stream) != hipSuccess) { fprintf(stderr, "no device\n"); return 2; }
    int bad = 0;
    for (u64 n : {(u64)1, (u64)63, (u64)4095, (u64)4096, (u64)4097, (u64)1000003, (u64)15100000, (u64)121419222}) {
        Scratch sc(&ctx);
        u32 *in = sc.get<u32>(n), *out = sc.get<u32>(n + 1), *d_tot = sc.get<u32>(1);
        hipLaunchKernelGGL(k_fill32, dim3((u32)div_up(n, 256)), dim3(256), 0, ctx.stream, in, n, 777ULL);
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            (void)hipEventRecord(a, ctx.stream);
            if (scan_exclusive_u32(&ctx, sc, in, out, n, d_tot)) { fprintf(stderr, "scan failed: %s\n", ctx.err.c_str()); return 1; }
            (void)hipEventRecord(b, ctx.stream);
            if (hipStreamSynchronize(ctx.stream) != hipSuccess) { fprintf(stderr, "device error: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
            float ms = 0; (void)hipEventElapsedTime(&ms, a, b); best = std::min(best, ms);
        }
        std::vector<u32> h(n), o(n); u32 tot = 0;
        (void)hipMemcpy(h.data(), in, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(o.data(), out, n * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&tot, d_tot, 4, hipMemcpyDeviceToHost);
        u32 run = 0; u64 wrong = 0;
        for (u64 i = 0; i < n; ++i) { if (o[i] != run) ++wrong; run += h[i]; }
        // in place
        if (scan_exclusive_u32(&ctx, sc, in, in, n, nullptr) || hipStreamSynchronize(ctx.stream) != hipSuccess) return 1;
        (void)hipMemcpy(h.data(), in, n * 4, hipMemcpyDeviceToHost);
        const bool same = h == o;
        printf("n = %10llu: %s (wrong %llu, total %s, in place %s)  %.3f ms  %.1f GB/s (read + write)\n", (unsigned long long)n,
               wrong == 0 && tot == run && same ? "ok" : "WRONG", (unsigned long long)wrong, tot == run ? "ok" : "WRONG", same ? "ok" : "WRONG", best, n * 8 / best * 1e-6);
        if (wrong || tot != run || !same) bad = 1;
    }
    ctx.pool.destroy();
    return bad;
}
