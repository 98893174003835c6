// Issue cost of the VALU instruction classes the env tick is made of (float64 arithmetic, float64 transcendental seeds, converts,
// compares / selects, 32-bit integer multiplies of Philox), per wave64 instruction, at 1 / 2 / 4 waves per SIMD, with the shader
// clock measured in the same launch (s_memtime ticks over wall_clock64's 100 MHz) - the inputs of the VALU-issue roofline of the
// register-resident kernels (tools/valu_model.py).  Eight independent dependency chains per wave, so the
// figure is ISSUE cost, not latency.
// Build + run: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_f64.hip -o gpurun_scratch/ubench_f64 && ./gpurun_scratch/ubench_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 2048

#define CHAIN8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

// clk[0] += s_memtime ticks of wave 0 of block 0, clk[1] += wall_clock64 ticks (100 MHz)
#define TIMED_BEGIN  const uint64_t c0 = __builtin_readcyclecounter(); const uint64_t w0 = wall_clock64();
#define TIMED_END    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - w0; }

#define KERNEL_D(NAME, ASM3)                                                                                          \
    __global__ void __launch_bounds__(1024) NAME(double* o, double s, uint64_t* clk) {                                \
        double a0 = s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        double b = 1.0000001;                                                                                         \
        TIMED_BEGIN                                                                                                   \
        for (int i = 0; i < ITER; ++i) {                                                                              \
            asm volatile(ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%4") ASM3("%5") ASM3("%6") ASM3("%7")      \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));  \
        }                                                                                                             \
        TIMED_END                                                                                                     \
        o[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                             \
    }
#define A_FMA64(R)   "v_fma_f64 " R ", " R ", %8, " R "\n"
#define A_MUL64(R)   "v_mul_f64 " R ", " R ", %8\n"
#define A_ADD64(R)   "v_add_f64 " R ", " R ", %8\n"
#define A_RCP64(R)   "v_rcp_f64 " R ", " R "\n"
#define A_RSQ64(R)   "v_rsq_f64 " R ", " R "\n"
#define A_SQRT64(R)  "v_sqrt_f64 " R ", " R "\n"
#define A_TRUNC64(R) "v_trunc_f64 " R ", " R "\n"
#define A_RNDNE64(R) "v_rndne_f64 " R ", " R "\n"
#define A_MAX64(R)   "v_max_f64 " R ", " R ", %8\n"
#define A_LDEXP64(R) "v_ldexp_f64 " R ", " R ", 1\n"
#define A_DIVFIX(R)  "v_div_fixup_f64 " R ", " R ", %8, " R "\n"
KERNEL_D(k_fma64, A_FMA64)
KERNEL_D(k_mul64, A_MUL64)
KERNEL_D(k_add64, A_ADD64)
KERNEL_D(k_rcp64, A_RCP64)
KERNEL_D(k_rsq64, A_RSQ64)
KERNEL_D(k_sqrt64, A_SQRT64)
KERNEL_D(k_trunc64, A_TRUNC64)
KERNEL_D(k_rndne64, A_RNDNE64)
KERNEL_D(k_max64, A_MAX64)
KERNEL_D(k_ldexp64, A_LDEXP64)
KERNEL_D(k_divfix64, A_DIVFIX)

#define KERNEL_F(NAME, ASM3)                                                                                          \
    __global__ void __launch_bounds__(1024) NAME(double* o, double s, uint64_t* clk) {                                \
        float a0 = (float)s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b = 1.0000001f;                                                                                         \
        TIMED_BEGIN                                                                                                   \
        for (int i = 0; i < ITER; ++i) {                                                                              \
            asm volatile(ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%4") ASM3("%5") ASM3("%6") ASM3("%7")      \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));  \
        }                                                                                                             \
        TIMED_END                                                                                                     \
        o[blockIdx.x * blockDim.x + threadIdx.x] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                   \
    }
#define A_FMA32(R)   "v_fma_f32 " R ", " R ", %8, " R "\n"
#define A_MUL32(R)   "v_mul_f32 " R ", " R ", %8\n"
#define A_SQRT32(R)  "v_sqrt_f32 " R ", " R "\n"
#define A_MULLO(R)   "v_mul_lo_u32 " R ", " R ", %8\n"
#define A_MULHI(R)   "v_mul_hi_u32 " R ", " R ", %8\n"
#define A_XOR(R)     "v_xor_b32 " R ", " R ", %8\n"
#define A_CNDMASK(R) "v_cndmask_b32 " R ", " R ", %8, vcc\n"
#define A_MOV(R)     "v_mov_b32 " R ", %8\n"
#define A_LSHLADD(R) "v_lshl_add_u32 " R ", " R ", 1, %8\n"
KERNEL_F(k_fma32, A_FMA32)
KERNEL_F(k_mul32, A_MUL32)
KERNEL_F(k_sqrt32, A_SQRT32)
KERNEL_F(k_mullo, A_MULLO)
KERNEL_F(k_mulhi, A_MULHI)
KERNEL_F(k_xor, A_XOR)
KERNEL_F(k_cndmask, A_CNDMASK)
KERNEL_F(k_mov, A_MOV)
KERNEL_F(k_lshladd, A_LSHLADD)

// conversions and compares have other register shapes
__global__ void __launch_bounds__(1024) k_cvt_f64_f32(double* o, double s, uint64_t* clk) {
    float f0 = (float)s + threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    TIMED_BEGIN
    for (int i = 0; i < ITER; ++i)
        asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n v_cvt_f64_f32 %4, %8\n v_cvt_f64_f32 %5, %9\n v_cvt_f64_f32 %6, %10\n v_cvt_f64_f32 %7, %11\n"
                     : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
    TIMED_END
    o[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void __launch_bounds__(1024) k_cvt_f32_f64(double* o, double s, uint64_t* clk) {
    double d0 = s + threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    TIMED_BEGIN
    for (int i = 0; i < ITER; ++i)
        asm volatile("v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n v_cvt_f32_f64 %4, %8\n v_cvt_f32_f64 %5, %9\n v_cvt_f32_f64 %6, %10\n v_cvt_f32_f64 %7, %11\n"
                     : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
    TIMED_END
    o[blockIdx.x * blockDim.x + threadIdx.x] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void __launch_bounds__(1024) k_cmp64(double* o, double s, uint64_t* clk) {
    double d0 = s + threadIdx.x, d1 = d0 + 1;
    TIMED_BEGIN
    for (int i = 0; i < ITER; ++i)
        asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_gt_f64 vcc, %0, %1\n v_cmp_le_f64 vcc, %0, %1\n v_cmp_ge_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %0\n v_cmp_gt_f64 vcc, %1, %0\n v_cmp_le_f64 vcc, %1, %0\n v_cmp_ge_f64 vcc, %1, %0\n"
                     :: "v"(d0), "v"(d1) : "vcc");
    TIMED_END
    o[blockIdx.x * blockDim.x + threadIdx.x] = d0;
}

template <typename K>
static void run(K k, const char* name, double* d, uint64_t* clk_dev, FILE* js, bool first) {
    printf("%-18s", name);
    if (js) fprintf(js, "%s\n  \"%s\": {", first ? "" : ",", name);
    int col = 0;
    for (int threads : {256, 512, 1024}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, 1.0, clk_dev);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, 1.0, clk_dev);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        uint64_t clk[2];
        hipMemcpy(clk, clk_dev, sizeof(clk), hipMemcpyDeviceToHost);
        const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);          // s_memtime ticks per ns (wall_clock64 = 100 MHz)
        const int waves_per_simd = threads / 256;
        // in-kernel: cycles of wave 0 per instruction it issued (issue interval as that wave sees it)
        const double cyc_wave = (double)clk[0] / ((double)ITER * 8.0);
        // SIMD's cost per instruction = that / waves sharing the SIMD
        const double cyc_simd = cyc_wave / waves_per_simd;
        printf("  %dw/SIMD: %5.2f cyc/instr/wave = %5.2f cyc per SIMD-instr @ %.2f GHz (ev %.2f ns)", waves_per_simd, cyc_wave, cyc_simd, ghz,
               ms * 1e6 / 5 / ITER / 8);
        if (js) fprintf(js, "%s\"w%d\": {\"cyc_per_instr_wave\": %.3f, \"cyc_per_instr_simd\": %.3f, \"ghz\": %.3f}", col ? ", " : "", waves_per_simd, cyc_wave, cyc_simd, ghz);
        ++col;
    }
    printf("\n");
    if (js) fprintf(js, "}");
}

int main(int argc, char** argv) {
    double* d;
    uint64_t* clk;
    hipMalloc(&d, 256 * 1024 * 8);
    hipMalloc(&clk, 16);
    FILE* js = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (js) fprintf(js, "{");
    bool first = true;
#define RUN(K, NAME) run(K, NAME, d, clk, js, first); first = false;
    RUN(k_fma64, "v_fma_f64") RUN(k_mul64, "v_mul_f64") RUN(k_add64, "v_add_f64") RUN(k_max64, "v_max_f64") RUN(k_trunc64, "v_trunc_f64")
    RUN(k_rndne64, "v_rndne_f64") RUN(k_ldexp64, "v_ldexp_f64") RUN(k_divfix64, "v_div_fixup_f64") RUN(k_cmp64, "v_cmp_f64")
    RUN(k_rcp64, "v_rcp_f64") RUN(k_rsq64, "v_rsq_f64") RUN(k_sqrt64, "v_sqrt_f64")
    RUN(k_cvt_f64_f32, "v_cvt_f64_f32") RUN(k_cvt_f32_f64, "v_cvt_f32_f64")
    RUN(k_fma32, "v_fma_f32") RUN(k_mul32, "v_mul_f32") RUN(k_sqrt32, "v_sqrt_f32") RUN(k_mullo, "v_mul_lo_u32") RUN(k_mulhi, "v_mul_hi_u32")
    RUN(k_xor, "v_xor_b32") RUN(k_cndmask, "v_cndmask_b32") RUN(k_mov, "v_mov_b32") RUN(k_lshladd, "v_lshl_add_u32")
    if (js) { fprintf(js, "\n}\n"); fclose(js); }
    return 0;
}
