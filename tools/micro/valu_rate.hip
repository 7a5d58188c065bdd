// valu_rate.hip -- how fast does one gfx950 SIMD issue wave64 VALU instructions, dependent and independent, at 1..8 waves/SIMD?
// Settles the issue model DESIGN.md section 3 (K6) prices the chain kernels with.  One workgroup per CU (100 KB of LDS keeps a
// second one out), 4 * W wavefronts in it, i.e. W per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o valu_rate tools/micro/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 4096
template <int MODE> __global__ void k(unsigned *out, unsigned seed) {
    extern __shared__ unsigned lds[];
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    unsigned b = seed | 1;
    for (int r = 0; r < REP; ++r) {
        if (MODE == 0) {        // 16 dependent integer adds
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(b));
        } else if (MODE == 1) { // 16 adds on 8 independent registers
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a1) : "v"(b));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a3) : "v"(b));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a4) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a5) : "v"(b));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a6) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a7) : "v"(b));
            }
        } else if (MODE == 2) { // 8 dependent compare + select pairs (16 instructions), the k_chain_lpg pattern
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %1, %0, vcc" : "+v"(a0) : "v"(b) : "vcc");
        } else if (MODE == 3) { // 16 dependent f32 fma
            float f = __uint_as_float((a0 & 0xffff) | 0x3f800000u), g = 1.0000001f, h = 1e-9f;
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(g), "v"(h));
            a0 = __float_as_uint(f);
        } else if (MODE == 4) { // 16 dependent max3 / min (integer, VOP3)
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a0) : "v"(b));
        } else if (MODE == 5) { // DPP: 16 dependent row_shr moves
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + lds[threadIdx.x & 15];
}

template <int MODE> static void run(const char *name, unsigned *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount; const double ghz = p.clockRate * 1e-6;
    printf("%-34s", name);
    for (int W : {1, 2, 3, 4, 8}) {
        const int threads = 256 * W;
        if (threads > 1024) {   // W = 8: two workgroups of 16 wavefronts per CU (70 KB of LDS each)
            hipLaunchKernelGGL(k<MODE>, dim3(cus * 2), dim3(1024), 70 * 1024, 0, d, 1u);
            hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(cus * 2), dim3(1024), 70 * 1024, 0, d, 2u); hipEventRecord(e1);
        } else {
            hipLaunchKernelGGL(k<MODE>, dim3(cus), dim3(threads), 100 * 1024, 0, d, 1u);
            hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(cus), dim3(threads), 100 * 1024, 0, d, 2u); hipEventRecord(e1);
        }
        hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_wave = 16.0 * REP;
        const double cyc = ms * 1e-3 * ghz * 1e9;
        // per-wave cadence (cycles between two instructions of one wave) and per-SIMD issue interval
        printf("  W=%d: %5.2f cyc/instr/wave, %5.2f cyc/instr/SIMD |", W, cyc / instr_per_wave, cyc / (instr_per_wave * W));
    }
    printf("\n");
}

int main() {
    unsigned *d; hipMalloc(&d, 256 * 2048 * 4 * 4);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, %.0f MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate * 1e-3);
    run<0>("dependent v_add_u32", d);
    run<1>("8 independent v_add_u32", d);
    run<2>("dependent v_cmp + v_cndmask", d);
    run<3>("dependent v_fma_f32", d);
    run<4>("dependent v_max_i32", d);
    run<5>("dependent v_add_u32 dpp row_shr", d);
    return 0;
}
