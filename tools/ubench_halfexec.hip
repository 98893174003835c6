// VERDICT r4 item 6's experiment: does gfx950 skip the 16-lane passes of a wave64 VALU instruction whose EXEC half is empty?
// If it did, 65 536 envs could run as 2 048 half-full waves (two per SIMD) at no issue cost and gain the second wave's latency hiding.
// Eight independent dependency chains per wave (issue cost, not latency), at 1 and 2 waves per SIMD, for float64 FMA / MUL / ADD and
// float32 FMA, with EXEC = all 64 lanes, the LOW 32 lanes, the EVEN lanes (control: as many active lanes, spread over every pass),
// and the low 16 lanes.  Prints shader cycles per instruction per wave (s_memtime over wall_clock64, as tools/ubench_f64.hip).
// Build + run: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_halfexec.hip -o tools/_bin/ubench_halfexec && tools/_bin/ubench_halfexec
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 4096

#define A_FMA64(R) "v_fma_f64 " R ", " R ", %8, " R "\n"
#define A_MUL64(R) "v_mul_f64 " R ", " R ", %8\n"
#define A_ADD64(R) "v_add_f64 " R ", " R ", %8\n"
#define A_FMA32(R) "v_fma_f32 " R ", " R ", %8, " R "\n"

// MODE 0: all lanes, 1: lanes 0..31, 2: even lanes, 3: lanes 0..15
template <int MODE>
__device__ __forceinline__ bool active(uint32_t lane) {
    return MODE == 0 ? true : MODE == 1 ? lane < 32u : MODE == 2 ? (lane & 1u) == 0u : lane < 16u;
}

#define KERNEL(NAME, T, ASM3)                                                                                          \
    template <int MODE>                                                                                                \
    __global__ void __launch_bounds__(512) NAME(double* o, double s, uint64_t* clk) {                                  \
        T a0 = (T)s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        T b = (T)1.0000001;                                                                                            \
        const uint64_t c0 = __builtin_readcyclecounter(); const uint64_t w0 = wall_clock64();                          \
        if (active<MODE>(threadIdx.x & 63u)) {                                                                         \
            for (int i = 0; i < ITER; ++i) {                                                                           \
                asm volatile(ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%4") ASM3("%5") ASM3("%6") ASM3("%7")   \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); \
            }                                                                                                          \
        }                                                                                                              \
        if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - w0; } \
        o[blockIdx.x * blockDim.x + threadIdx.x] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                    \
    }
KERNEL(k_fma64, double, A_FMA64)
KERNEL(k_mul64, double, A_MUL64)
KERNEL(k_add64, double, A_ADD64)
KERNEL(k_fma32, float, A_FMA32)

template <typename K>
static void run(K k, const char* name, const char* mode, double* d, uint64_t* clk_dev) {
    printf("%-10s %-12s", name, mode);
    for (int threads : {256, 512}) {
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, 1.0, clk_dev);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, 1.0, clk_dev);
        hipDeviceSynchronize();
        uint64_t clk[2];
        hipMemcpy(clk, clk_dev, sizeof(clk), hipMemcpyDeviceToHost);
        const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);
        printf("  %dw/SIMD: %6.2f cyc/instr/wave @ %.2f GHz", threads / 256, (double)clk[0] / ((double)ITER * 8.0), ghz);
    }
    printf("\n");
}

int main() {
    double* d;
    uint64_t* clk;
    hipMalloc(&d, 256 * 512 * 8);
    hipMalloc(&clk, 16);
#define RUN4(K, NAME) run(K<0>, NAME, "all 64", d, clk); run(K<1>, NAME, "low 32", d, clk); run(K<2>, NAME, "even lanes", d, clk); run(K<3>, NAME, "low 16", d, clk);
    RUN4(k_fma64, "v_fma_f64") RUN4(k_mul64, "v_mul_f64") RUN4(k_add64, "v_add_f64") RUN4(k_fma32, "v_fma_f32")
    return 0;
}
