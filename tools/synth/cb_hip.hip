// cb_hip.hip -- device twin of the counter-based read generator (cb_core.h): writes the ASCII reads of a whole
// configuration straight into HBM (31 GB for the H. sapiens-scale set) in well under a second, bit-identical to the
// host twin (cb_host.c).  TEST / BENCH INFRASTRUCTURE, a library of its own (libcbgen_hip.so): the product library does
// not contain it.  gfx950: one 256-lane workgroup per read, 16 consecutive source bases per lane and step, block scan of
// the emitted counts (wave scan by DPP-lowered shuffles + 4 wave totals in LDS), byte stores of contiguous runs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cb_core.h"

#define CB_THREADS 256
#define CB_PER 16

struct ReadMeta { uint64_t rk, start; uint32_t len, strand; };

__device__ static inline ReadMeta read_meta(const cb_params &p, const uint32_t *lentab, uint64_t i) {
    ReadMeta m;
    m.rk = cb_read_key(&p, i);
    m.len = cb_read_len(&p, m.rk, lentab);
    m.start = cb_read_start(&p, m.rk, m.len);
    m.strand = cb_read_strand(m.rk);
    return m;
}

// emitted length of every read (and its length / start / strand if asked for)
__global__ __launch_bounds__(CB_THREADS) void k_cb_count(cb_params p, const uint32_t *__restrict__ lentab, uint64_t first, uint64_t n,
                                                          uint32_t *__restrict__ out_len) {
    __shared__ uint32_t wsum[CB_THREADS / 64];
    for (uint64_t r = blockIdx.x; r < n; r += gridDim.x) {
        const ReadMeta m = read_meta(p, lentab, first + r);
        uint32_t o = 0, c0, c1;
        for (uint32_t j = threadIdx.x; j < m.len; j += CB_THREADS) o += cb_emit(&p, m.rk, m.start, m.len, m.strand, j, &c0, &c1);
        for (int d = 32; d; d >>= 1) o += __shfl_xor(o, d);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = o;
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t t = wsum[0] + wsum[1] + wsum[2] + wsum[3]; out_len[r] = t ? t : 1; }
        __syncthreads();
    }
}

__global__ __launch_bounds__(CB_THREADS) void k_cb_write(cb_params p, const uint32_t *__restrict__ lentab, uint64_t first, uint64_t n,
                                                          const uint64_t *__restrict__ offsets, char *__restrict__ bases) {
    __shared__ uint32_t wsum[CB_THREADS / 64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint64_t r = blockIdx.x; r < n; r += gridDim.x) {
        const ReadMeta m = read_meta(p, lentab, first + r);
        char *o = bases + offsets[r];
        uint32_t done = 0;          // output bases written by earlier steps
        for (uint32_t j0 = 0; j0 < m.len; j0 += CB_THREADS * CB_PER) {
            const uint32_t jb = j0 + threadIdx.x * CB_PER;
            uint8_t buf[2 * CB_PER];
            uint32_t cnt = 0;
#pragma unroll
            for (int t = 0; t < CB_PER; ++t) {
                const uint32_t j = jb + t;
                if (j < m.len) {
                    uint32_t c0, c1;
                    const uint32_t e = cb_emit(&p, m.rk, m.start, m.len, m.strand, j, &c0, &c1);
                    if (e >= 1) buf[cnt++] = (uint8_t)c0;
                    if (e == 2) buf[cnt++] = (uint8_t)c1;
                }
            }
            // exclusive scan of cnt over the workgroup
            uint32_t inc = cnt;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(inc, d); if ((int)lane >= d) inc += v; }
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
            uint32_t base = done;
            for (uint32_t w = 0; w < wave; ++w) base += wsum[w];
            const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            __syncthreads();
            char *dst = o + base + (inc - cnt);
            for (uint32_t t = 0; t < cnt; ++t) dst[t] = "ACGT"[buf[t]];
            done += total;
        }
        if (done == 0 && threadIdx.x == 0) {
            uint32_t code = cb_genome_base(&p, m.strand ? m.start + m.len - 1 : m.start);
            o[0] = "ACGT"[m.strand ? 3u - code : code];
        }
    }
}

#define CBCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return (int)e_; } while (0)

extern "C" {
// Emitted lengths of reads [first, first + n) into the HOST array out_len (the caller scans them into offsets).
int cb_hip_count(int device, const cb_params *p, const uint32_t *lentab_host, uint64_t first, uint64_t n, uint32_t *out_len_host) {
    CBCHK(hipSetDevice(device));
    uint32_t *d_tab = nullptr, *d_len = nullptr;
    CBCHK(hipMalloc(&d_tab, 65536 * 4));
    CBCHK(hipMalloc(&d_len, (n ? n : 1) * 4));
    CBCHK(hipMemcpy(d_tab, lentab_host, 65536 * 4, hipMemcpyHostToDevice));
    if (n) hipLaunchKernelGGL(k_cb_count, dim3((unsigned)(n < (1u << 20) ? n : (1u << 20))), dim3(CB_THREADS), 0, 0, *p, d_tab, first, n, d_len);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out_len_host, d_len, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_tab); (void)hipFree(d_len);
    return (int)e;
}
// ASCII bases of reads [first, first + n) into DEVICE memory d_bases (offsets_host[n + 1], from the emitted lengths).
int cb_hip_write(int device, const cb_params *p, const uint32_t *lentab_host, uint64_t first, uint64_t n, const uint64_t *offsets_host,
                 void *d_bases) {
    CBCHK(hipSetDevice(device));
    uint32_t *d_tab = nullptr; uint64_t *d_off = nullptr;
    CBCHK(hipMalloc(&d_tab, 65536 * 4));
    CBCHK(hipMalloc(&d_off, (n + 1) * 8));
    CBCHK(hipMemcpy(d_tab, lentab_host, 65536 * 4, hipMemcpyHostToDevice));
    CBCHK(hipMemcpy(d_off, offsets_host, (n + 1) * 8, hipMemcpyHostToDevice));
    if (n) hipLaunchKernelGGL(k_cb_write, dim3((unsigned)(n < (1u << 20) ? n : (1u << 20))), dim3(CB_THREADS), 0, 0, *p, d_tab, first, n, d_off, (char *)d_bases);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(d_tab); (void)hipFree(d_off);
    return (int)e;
}
int cb_hip_malloc(int device, uint64_t bytes, void **out) { CBCHK(hipSetDevice(device)); return (int)hipMalloc(out, bytes ? bytes : 1); }
int cb_hip_free(int device, void *p) { CBCHK(hipSetDevice(device)); return (int)hipFree(p); }
int cb_hip_to_host(int device, void *dst, const void *src, uint64_t bytes) { CBCHK(hipSetDevice(device)); return (int)hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost); }
int cb_hip_mem_info(int device, uint64_t *free_b, uint64_t *total_b) {
    CBCHK(hipSetDevice(device));
    size_t f = 0, t = 0; hipError_t e = hipMemGetInfo(&f, &t); *free_b = f; *total_b = t; return (int)e;
}
}
