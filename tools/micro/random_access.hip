// random_access.hip -- how many random 128-byte lines per second does the chip deliver, as a function of the working set?
// (one 16-byte load per thread, one line per load: k_lookup's pattern).  Development harness.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k_probe(const ulonglong2 *__restrict__ tab, uint64_t n_lines, uint64_t n, uint64_t seed, unsigned long long *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = (i + seed) * 0x9E3779B97F4A7C15ULL; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
    const uint64_t line = __umul64hi(x, n_lines);
    const ulonglong2 v = tab[line * 8 + (x & 7)];           // 16-byte slot inside a 128-byte line
    if (v.x == 0x1234567ULL) atomicAdd(out, v.y);             // (never true: keeps the load alive)
}

// k_lookup's shape: read a key stream, probe, continue linearly with probability ~1/3 per step (dependent loads, mostly in
// the same line), fetch the second half of the slot on a hit, write three result streams
template <int MODE>
__global__ __launch_bounds__(256) void k_probe2(const unsigned long long *__restrict__ qx, const unsigned long long *__restrict__ tab, uint64_t n_slots,
                                                uint64_t n, unsigned *__restrict__ o1, unsigned *__restrict__ o2, unsigned *__restrict__ o3) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = MODE & 1 ? qx[i] : (i + 7) * 0x9E3779B97F4A7C15ULL;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
    uint64_t slot = __umul64hi(x, n_slots - 64);
    unsigned long long k = tab[2 * slot], v = 0;
    if (MODE & 2) {
        unsigned steps = 0;
        uint64_t h = x;
        while ((h & 3) == 0 && steps < 8) {            // continue with probability 1/4 per step
            ++slot; ++steps; k ^= tab[2 * slot]; h = (h >> 2) ^ k;
        }
        if ((x >> 40) % 3 == 0) v = tab[2 * slot + 1];  // a third of the keys are present: fetch start | count
    }
    if (MODE & 4) { o1[i] = (unsigned)k; o2[i] = (unsigned)v; o3[i] = (unsigned)(k >> 32); }
    else if (k == 0x1234567ULL) o1[0] = (unsigned)v;
}

template <int MODE> static void run2(const char *name, uint64_t mb) {
    const uint64_t n = 121ULL << 20, bytes = mb << 20;
    unsigned long long *tab, *qx; unsigned *o1, *o2, *o3;
    (void)hipMalloc(&tab, bytes); (void)hipMemset(tab, 0xAB, bytes);
    (void)hipMalloc(&qx, n * 8); (void)hipMemset(qx, 0x5C, n * 8);
    (void)hipMalloc(&o1, n * 4); (void)hipMalloc(&o2, n * 4); (void)hipMalloc(&o3, n * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k_probe2<MODE>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, qx, tab, bytes / 16, n, o1, o2, o3);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("%-52s %6llu MB: %7.3f ms = %6.1f G probes/s\n", name, (unsigned long long)mb, best, n / best * 1e-6);
    (void)hipFree(tab); (void)hipFree(qx); (void)hipFree(o1); (void)hipFree(o2); (void)hipFree(o3);
}

// k_expand's gather: consecutive lanes read consecutive 8-byte entries of short lists (G entries each) at random places
template <int G>
__global__ __launch_bounds__(256) void k_lists(const unsigned long long *__restrict__ tab, uint64_t n_entries, uint64_t n, unsigned long long *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t list = i / G, j = i % G;
    uint64_t x = (list + 7) * 0x9E3779B97F4A7C15ULL; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
    const unsigned long long v = tab[__umul64hi(x, n_entries - G) + j];
    if (v == 0x1234567ULL) atomicAdd(out, v);
}
template <int G> static void run_lists() {
    const uint64_t n = 211ULL << 20, bytes = 1940ULL << 20;
    unsigned long long *tab, *d_out; (void)hipMalloc(&tab, bytes); (void)hipMemset(tab, 0xAB, bytes); (void)hipMalloc(&d_out, 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k_lists<G>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, tab, bytes / 8, n, d_out);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("lists of %d entries, 211 M entries gathered from 1.9 GB: %7.3f ms = %5.1f G lists/s\n", G, best, (n / G) / best * 1e-6);
    (void)hipFree(tab); (void)hipFree(d_out);
}

int main() {
    run_lists<1>(); run_lists<2>(); run_lists<5>(); run_lists<13>();
    run2<0>("one 8-byte load per thread", 6144);
    run2<1>("+ key stream read", 6144);
    run2<2>("+ linear continuation + value fetch", 6144);
    run2<4>("+ three result streams", 6144);
    run2<7>("all (k_lookup's shape)", 6144);
    const uint64_t n = 128ULL << 20;
    unsigned long long *d_out; (void)hipMalloc(&d_out, 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (uint64_t mb : {2ULL, 16ULL, 64ULL, 128ULL, 256ULL, 512ULL, 1024ULL, 2048ULL, 6144ULL, 16384ULL}) {
        const uint64_t bytes = mb << 20;
        ulonglong2 *tab;
        if (hipMalloc(&tab, bytes) != hipSuccess) { printf("%llu MB: alloc failed\n", (unsigned long long)mb); continue; }
        (void)hipMemset(tab, 0xAB, bytes);
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(k_probe, dim3((unsigned)(n / 256)), dim3(256), 0, 0, tab, bytes / 128, n, 99ULL + r, d_out);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("working set %6llu MB: %7.3f ms for %llu M probes = %6.1f G lines/s (%5.2f TB/s of line traffic)\n", (unsigned long long)mb, best,
               (unsigned long long)(n >> 20), n / best * 1e-6, n * 128.0 / best * 1e-9);
        (void)hipFree(tab);
    }
    return 0;
}
