// Instruction-rate micro-benchmarks on gfx950 (issue cost per wave64 instruction; MFMA/VALU co-issue) behind DESIGN.md section 8.
// Build + run: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_gfx950.hip -o gpurun_scratch/ubench && gpurun -- ./gpurun_scratch/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define ITER 4096
#define REP8(X) X X X X X X X X

__global__ void __launch_bounds__(512) k_exp(float* o, float s) {
    float a0 = s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < ITER; ++i) {
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void __launch_bounds__(512) k_rcp(float* o, float s) {
    float a0 = s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < ITER; ++i) {
        asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void __launch_bounds__(512) k_fma(float* o, float s) {
    float a0 = s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < ITER; ++i) {
        asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void __launch_bounds__(512) k_pkfma(float* o, float s) {
    f32x2 a0 = {s + threadIdx.x, s}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    for (int i = 0; i < ITER; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    f32x2 r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    o[blockIdx.x * blockDim.x + threadIdx.x] = r[0] + r[1];
}
__global__ void __launch_bounds__(512) k_cvt(float* o, float s) {
    float a0 = s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < ITER; ++i) {
        asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
// 8 independent MFMAs per iteration (+ optionally V interleaved v_exp per MFMA)
template <int V>
__global__ void __launch_bounds__(512) k_mfma(float* o, float s) {
    f32x16 c[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) c[t][r] = s + t + r;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3F80 + ((threadIdx.x + j) & 63)); b[j] = (short)(0x3C00 + ((threadIdx.x * 3 + j) & 63)); }
    float e0 = s, e1 = s + 1, e2 = s + 2, e3 = s + 3;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            c[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[t], 0, 0, 0);
            if (V >= 1) asm volatile("v_exp_f32 %0, %0" : "+v"(e0));
            if (V >= 2) asm volatile("v_exp_f32 %0, %0" : "+v"(e1));
            if (V >= 3) asm volatile("v_exp_f32 %0, %0" : "+v"(e2));
            if (V >= 4) asm volatile("v_exp_f32 %0, %0" : "+v"(e3));
            if (V >= 5) asm volatile("v_exp_f32 %0, %0" : "+v"(e0));
            if (V >= 6) asm volatile("v_exp_f32 %0, %0" : "+v"(e1));
            if (V >= 7) asm volatile("v_exp_f32 %0, %0" : "+v"(e2));
            if (V >= 8) asm volatile("v_exp_f32 %0, %0" : "+v"(e3));
        }
    }
    float r = e0 + e1 + e2 + e3;
    for (int t = 0; t < 8; ++t) for (int q = 0; q < 16; ++q) r += c[t][q];
    o[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
// MFMA + V plain FMAs per MFMA
template <int V>
__global__ void __launch_bounds__(512) k_mfma_fma(float* o, float s) {
    f32x16 c[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) c[t][r] = s + t + r;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3F80 + ((threadIdx.x + j) & 63)); b[j] = (short)(0x3C00 + ((threadIdx.x * 3 + j) & 63)); }
    float e0 = s, e1 = s + 1, e2 = s + 2, e3 = s + 3;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            c[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[t], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                if ((v & 3) == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e0));
                if ((v & 3) == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e1));
                if ((v & 3) == 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e2));
                if ((v & 3) == 3) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e3));
            }
        }
    }
    float r = e0 + e1 + e2 + e3;
    for (int t = 0; t < 8; ++t) for (int q = 0; q < 16; ++q) r += c[t][q];
    o[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void __launch_bounds__(512) k_accread(float* o, float s) {
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = s + r;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %0\n v_accvgpr_write_b32 a2, %0\n v_accvgpr_write_b32 a3, %0" :: "v"(s) : "a0", "a1", "a2", "a3");
    for (int i = 0; i < ITER; ++i) {
        asm volatile("v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3\n v_accvgpr_read_b32 %4, a0\n v_accvgpr_read_b32 %5, a1\n v_accvgpr_read_b32 %6, a2\n v_accvgpr_read_b32 %7, a3\n"
                     : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) :: "a0", "a1", "a2", "a3");
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <typename K>
static double run(K k, const char* name, int per_iter, int threads, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double ns_per_instr = ms * 1e6 / 5 / ITER / per_iter;
    printf("%-28s threads/block=%4d  %.2f ns per instruction per wave  (= %.1f clk at 2.4 GHz)\n", name, threads, ns_per_instr, ns_per_instr * 2.4);
    return ns_per_instr;
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 1024 * 4);
    for (int threads : {256, 512}) {
        run(k_exp, "v_exp_f32", 8, threads, d);
        run(k_rcp, "v_rcp_f32", 8, threads, d);
        run(k_fma, "v_fma_f32", 8, threads, d);
        run(k_pkfma, "v_pk_fma_f32", 8, threads, d);
        run(k_cvt, "v_cvt_pk_bf16_f32", 8, threads, d);
        run(k_accread, "v_accvgpr_read_b32", 8, threads, d);
        run(k_mfma<0>, "mfma 32x32x16 bf16", 8, threads, d);
        run(k_mfma<1>, "mfma + 1 exp", 8, threads, d);
        run(k_mfma<2>, "mfma + 2 exp", 8, threads, d);
        run(k_mfma<3>, "mfma + 3 exp", 8, threads, d);
        run(k_mfma<4>, "mfma + 4 exp", 8, threads, d);
        run(k_mfma<6>, "mfma + 6 exp", 8, threads, d);
        run(k_mfma<8>, "mfma + 8 exp", 8, threads, d);
        run(k_mfma_fma<4>, "mfma + 4 fma", 8, threads, d);
        run(k_mfma_fma<7>, "mfma + 7 fma", 8, threads, d);
        run(k_mfma_fma<8>, "mfma + 8 fma", 8, threads, d);
        run(k_mfma_fma<12>, "mfma + 12 fma", 8, threads, d);
    }
    return 0;
}
