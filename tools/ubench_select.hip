// What does a per-lane select cost on gfx950?  tools/ubench_f64.hip measured 16.9 cycles per v_cndmask_b32 on a lone wave (9.4 ns of
// the SIMD per instruction at ANY wave count) against 4.8-5.2 for every other 32-bit / float64 VALU instruction.  The env tick holds
// ~45 of them.  This probe separates the causes: mask in VCC vs an SGPR pair, e32 vs e64 encoding, interleaving with other VALU work,
// a v_cmp in front, and the alternatives a select can be rewritten into (v_bfi_b32 / v_and_b32 with a lane mask in a VGPR, v_med3,
// v_max / v_min, multiply by 0 / 1).  Eight independent chains per wave: issue cost, not latency; plus single-chain latencies.
// Build + run: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_select.hip -o gpurun_scratch/ubench_select && ./gpurun_scratch/ubench_select
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 2048

#define KERNEL(NAME, BODY, NINSTR)                                                                                    \
    __global__ void __launch_bounds__(1024) NAME(float* o, float s, int* ninstr) {                                    \
        float a0 = s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b = 1.0000001f;                                                                                         \
        uint32_t m = (threadIdx.x & 1) ? 0xffffffffu : 0u;                                                            \
        double d0 = a0, d1 = a1, d2 = a2, d3 = a3;                                                                    \
        asm volatile("v_cmp_gt_f32 vcc, %0, %1\n s_mov_b64 s[10:11], vcc\n" :: "v"(a0), "v"(a3) : "vcc", "s10", "s11"); \
        for (int i = 0; i < ITER; ++i) {                                                                              \
            asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),      \
                                "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(m) : "vcc", "scc", "s12", "s13", "s14", "s15");       \
        }                                                                                                             \
        if (ninstr && threadIdx.x == 0 && blockIdx.x == 0) *ninstr = NINSTR;                                          \
        o[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3); \
    }
#define R8(OP) OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7")

#define I_MOV(R)      "v_mov_b32 " R ", %12\n"
#define I_CND_VCC(R)  "v_cndmask_b32 " R ", " R ", %12, vcc\n"
#define I_CND_SG(R)   "v_cndmask_b32_e64 " R ", " R ", %12, s[10:11]\n"
#define I_CND_CONST(R) "v_cndmask_b32_e64 " R ", 0, 1.0, s[10:11]\n"
#define I_BFI(R)      "v_bfi_b32 " R ", %13, %12, " R "\n"
#define I_AND(R)      "v_and_b32 " R ", %13, " R "\n"
#define I_MAX(R)      "v_max_f32 " R ", " R ", %12\n"
#define I_MED3(R)     "v_med3_f32 " R ", " R ", %12, 1.0\n"
#define I_FMA(R)      "v_fma_f32 " R ", " R ", %12, " R "\n"
#define I_CND_FMA(R)  "v_cndmask_b32 " R ", " R ", %12, vcc\n v_fma_f32 " R ", " R ", %12, " R "\n"
#define I_CMP_CND(R)  "v_cmp_gt_f32 vcc, " R ", %12\n v_cndmask_b32 " R ", " R ", %12, vcc\n"
#define I_CMP_SG_CND(R) "v_cmp_gt_f32 s[12:13], " R ", %12\n s_nop 1\n v_cndmask_b32_e64 " R ", " R ", %12, s[12:13]\n"
#define I_CMP(R)      "v_cmp_gt_f32 vcc, " R ", %12\n"
KERNEL(k_mov, R8(I_MOV), 8)
KERNEL(k_cnd_vcc, R8(I_CND_VCC), 8)
KERNEL(k_cnd_sg, R8(I_CND_SG), 8)
KERNEL(k_cnd_const, R8(I_CND_CONST), 8)
KERNEL(k_bfi, R8(I_BFI), 8)
KERNEL(k_and, R8(I_AND), 8)
KERNEL(k_max, R8(I_MAX), 8)
KERNEL(k_med3, R8(I_MED3), 8)
KERNEL(k_fma, R8(I_FMA), 8)
KERNEL(k_cnd_fma, R8(I_CND_FMA), 16)
KERNEL(k_cmp_cnd, R8(I_CMP_CND), 16)
KERNEL(k_cmp_sg_cnd, R8(I_CMP_SG_CND), 16)
KERNEL(k_cmp, R8(I_CMP), 8)
// float64 selects: two v_cndmask_b32 per value; the same through v_bfi pairs
#define D4(OP) OP("%8") OP("%9") OP("%10") OP("%11") OP("%8") OP("%9") OP("%10") OP("%11")
#define I_FMA64(R)    "v_fma_f64 " R ", " R ", " R ", " R "\n"
#define I_MAX64(R)    "v_max_f64 " R ", " R ", " R "\n"
KERNEL(k_fma64_dep, "v_fma_f64 %8, %8, %8, %8\n v_fma_f64 %8, %8, %8, %8\n v_fma_f64 %8, %8, %8, %8\n v_fma_f64 %8, %8, %8, %8\n v_fma_f64 %8, %8, %8, %8\n v_fma_f64 %8, %8, %8, %8\n v_fma_f64 %8, %8, %8, %8\n v_fma_f64 %8, %8, %8, %8\n", 8)
KERNEL(k_fma64_2ch, "v_fma_f64 %8, %8, %8, %8\n v_fma_f64 %9, %9, %9, %9\n v_fma_f64 %8, %8, %8, %8\n v_fma_f64 %9, %9, %9, %9\n v_fma_f64 %8, %8, %8, %8\n v_fma_f64 %9, %9, %9, %9\n v_fma_f64 %8, %8, %8, %8\n v_fma_f64 %9, %9, %9, %9\n", 8)
KERNEL(k_fma64_4ch, D4(I_FMA64), 8)
KERNEL(k_fma32_dep, "v_fma_f32 %0, %0, %12, %0\n v_fma_f32 %0, %0, %12, %0\n v_fma_f32 %0, %0, %12, %0\n v_fma_f32 %0, %0, %12, %0\n v_fma_f32 %0, %0, %12, %0\n v_fma_f32 %0, %0, %12, %0\n v_fma_f32 %0, %0, %12, %0\n v_fma_f32 %0, %0, %12, %0\n", 8)
KERNEL(k_cnd_dep, "v_cndmask_b32 %0, %0, %12, vcc\n v_cndmask_b32 %0, %0, %12, vcc\n v_cndmask_b32 %0, %0, %12, vcc\n v_cndmask_b32 %0, %0, %12, vcc\n v_cndmask_b32 %0, %0, %12, vcc\n v_cndmask_b32 %0, %0, %12, vcc\n v_cndmask_b32 %0, %0, %12, vcc\n v_cndmask_b32 %0, %0, %12, vcc\n", 8)

#define I_CND2_FMA2 "v_cndmask_b32 %0, %0, %12, vcc\n v_cndmask_b32 %1, %1, %12, vcc\n v_fma_f32 %2, %2, %12, %2\n v_fma_f32 %3, %3, %12, %3\n v_cndmask_b32 %4, %4, %12, vcc\n v_cndmask_b32 %5, %5, %12, vcc\n v_fma_f32 %6, %6, %12, %6\n v_fma_f32 %7, %7, %12, %7\n"
#define I_CND4_FMA4 "v_cndmask_b32 %0, %0, %12, vcc\n v_cndmask_b32 %1, %1, %12, vcc\n v_cndmask_b32 %2, %2, %12, vcc\n v_cndmask_b32 %3, %3, %12, vcc\n v_fma_f32 %4, %4, %12, %4\n v_fma_f32 %5, %5, %12, %5\n v_fma_f32 %6, %6, %12, %6\n v_fma_f32 %7, %7, %12, %7\n"
#define I_CND2SG_FMA2 "v_cndmask_b32_e64 %0, %0, %12, s[10:11]\n v_cndmask_b32_e64 %1, %1, %12, s[10:11]\n v_fma_f32 %2, %2, %12, %2\n v_fma_f32 %3, %3, %12, %3\n v_cndmask_b32_e64 %4, %4, %12, s[10:11]\n v_cndmask_b32_e64 %5, %5, %12, s[10:11]\n v_fma_f32 %6, %6, %12, %6\n v_fma_f32 %7, %7, %12, %7\n"
#define I_CMP_CND2 "v_cmp_gt_f32 vcc, %0, %12\n v_cndmask_b32 %0, %0, %12, vcc\n v_cndmask_b32 %1, %1, %12, vcc\n v_cmp_gt_f32 vcc, %2, %12\n v_cndmask_b32 %2, %2, %12, vcc\n v_cndmask_b32 %3, %3, %12, vcc\n"
#define I_CMP_CND4 "v_cmp_gt_f32 vcc, %0, %12\n v_cndmask_b32 %0, %0, %12, vcc\n v_cndmask_b32 %1, %1, %12, vcc\n v_cndmask_b32 %2, %2, %12, vcc\n v_cndmask_b32 %3, %3, %12, vcc\n v_fma_f32 %4, %4, %12, %4\n"
#define I_SALU8 "s_add_u32 s12, s12, 1\n s_and_b32 s13, s13, s12\n s_add_u32 s12, s12, 1\n s_and_b32 s13, s13, s12\n s_add_u32 s12, s12, 1\n s_and_b32 s13, s13, s12\n s_add_u32 s12, s12, 1\n s_and_b32 s13, s13, s12\n"
#define I_SALU_VALU "s_add_u32 s12, s12, 1\n v_fma_f32 %0, %0, %12, %0\n s_and_b32 s13, s13, s12\n v_fma_f32 %1, %1, %12, %1\n s_add_u32 s12, s12, 1\n v_fma_f32 %2, %2, %12, %2\n s_and_b32 s13, s13, s12\n v_fma_f32 %3, %3, %12, %3\n"
#define I_BR_VALU "v_fma_f32 %0, %0, %12, %0\n s_cbranch_vccz 1f\n 1: v_fma_f32 %1, %1, %12, %1\n s_cbranch_vccz 2f\n 2: v_fma_f32 %2, %2, %12, %2\n s_cbranch_vccz 3f\n 3: v_fma_f32 %3, %3, %12, %3\n s_cbranch_vccz 4f\n 4:\n"
#define I_BR_TAKEN "s_mov_b64 vcc, 0\n v_fma_f32 %0, %0, %12, %0\n s_cbranch_vccz 1f\n v_fma_f32 %4, %4, %12, %4\n 1: v_fma_f32 %1, %1, %12, %1\n s_cbranch_vccz 2f\n v_fma_f32 %4, %4, %12, %4\n 2: v_fma_f32 %2, %2, %12, %2\n s_cbranch_vccz 3f\n v_fma_f32 %4, %4, %12, %4\n 3: v_fma_f32 %3, %3, %12, %3\n s_cbranch_vccz 4f\n v_fma_f32 %4, %4, %12, %4\n 4:\n"
#define I_SAVEEXEC "v_cmp_gt_f32 vcc, %0, %12\n s_and_saveexec_b64 s[12:13], vcc\n v_fma_f32 %0, %0, %12, %0\n s_or_b64 exec, exec, s[12:13]\n v_cmp_gt_f32 vcc, %1, %12\n s_and_saveexec_b64 s[12:13], vcc\n v_fma_f32 %1, %1, %12, %1\n s_or_b64 exec, exec, s[12:13]\n"
#define I_NOP8 "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
#define I_READLANE "v_readlane_b32 s12, %0, 3\n v_fma_f32 %1, %1, %12, %1\n v_readlane_b32 s12, %2, 3\n v_fma_f32 %3, %3, %12, %3\n"
#define SEL_V1(A,B) "v_cmp_gt_f32 vcc, " A ", %12\n v_cndmask_b32 " A ", " A ", %12, vcc\n v_cndmask_b32 " B ", " B ", %12, vcc\n"
#define SEL_V2(A,B) "v_cmp_gt_f32 s[12:13], " A ", %12\n s_nop 1\n v_cndmask_b32_e64 " A ", " A ", %12, s[12:13]\n v_cndmask_b32_e64 " B ", " B ", %12, s[12:13]\n"
#define SEL_V3(A,B) "v_cmp_gt_f32 s[12:13], " A ", %12\n v_cndmask_b32_e64 " A ", " A ", %12, s[12:13]\n v_cndmask_b32_e64 " B ", " B ", %12, s[12:13]\n"
#define SEL_V4(A,B) "v_cmp_gt_f32 vcc, " A ", %12\n v_cndmask_b32 " A ", " A ", %12, vcc\n v_fma_f32 %7, %7, %12, %7\n v_cndmask_b32 " B ", " B ", %12, vcc\n"
#define SEL_V5(A,B) "v_cmp_gt_f32 vcc, " A ", %12\n v_cndmask_b32_e64 " A ", " A ", %12, vcc\n v_cndmask_b32_e64 " B ", " B ", %12, vcc\n"
#define SEL_V6(A,B) "v_cmp_gt_f32 vcc, " A ", %12\n v_cndmask_b32 " A ", " A ", %12, vcc\n v_cmp_gt_f32 s[12:13], " B ", %12\n v_cndmask_b32 " B ", " B ", %12, vcc\n"
#define SEL_V7(A,B) "v_cmp_gt_f32 vcc, " A ", %12\n s_nop 3\n v_cndmask_b32 " A ", " A ", %12, vcc\n v_cndmask_b32 " B ", " B ", %12, vcc\n"
#define SEL_V8(A,B) "v_cmp_gt_f32 vcc, " A ", %12\n v_cndmask_b32 " A ", " A ", %12, vcc\n s_nop 0\n v_cndmask_b32 " B ", " B ", %12, vcc\n"
#define SEL_V9(A,B) "v_cmp_gt_f32 s[12:13], " A ", %12\n v_fma_f32 %6, %6, %12, %6\n v_fma_f32 %7, %7, %12, %7\n v_cndmask_b32_e64 " A ", " A ", %12, s[12:13]\n v_cndmask_b32_e64 " B ", " B ", %12, s[12:13]\n"
#define SEL3(V) V("%0","%1") V("%2","%3") V("%4","%5")
KERNEL(k_sel_v1, SEL3(SEL_V1), 9)
KERNEL(k_sel_v2, SEL3(SEL_V2), 9)
KERNEL(k_sel_v3, SEL3(SEL_V3), 9)
KERNEL(k_sel_v4, SEL3(SEL_V4), 12)
KERNEL(k_sel_v5, SEL3(SEL_V5), 9)
KERNEL(k_sel_v6, SEL3(SEL_V6), 12)
KERNEL(k_sel_v7, SEL3(SEL_V7), 9)
KERNEL(k_sel_v8, SEL3(SEL_V8), 9)
KERNEL(k_sel_v9, SEL3(SEL_V9), 15)
KERNEL(k_cnd2_fma2, I_CND2_FMA2, 8)
KERNEL(k_cnd4_fma4, I_CND4_FMA4, 8)
KERNEL(k_cnd2sg_fma2, I_CND2SG_FMA2, 8)
KERNEL(k_cmp_cnd2, I_CMP_CND2, 6)
KERNEL(k_cmp_cnd4, I_CMP_CND4, 6)
KERNEL(k_salu8, I_SALU8, 8)
KERNEL(k_salu_valu, I_SALU_VALU, 8)
KERNEL(k_br_valu, I_BR_VALU, 8)
KERNEL(k_br_taken, I_BR_TAKEN, 9)
KERNEL(k_saveexec, I_SAVEEXEC, 8)
KERNEL(k_nop8, I_NOP8, 8)
KERNEL(k_readlane, I_READLANE, 4)

template <typename K>
static void run(K k, const char* name, float* d, int* nd) {
    printf("%-34s", name);
    for (int threads : {64, 256, 512, 1024}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, 1.0f, nd);
        hipDeviceSynchronize();
        int n = 8;
        hipMemcpy(&n, nd, sizeof(int), hipMemcpyDeviceToHost);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, 1.0f, nd);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double waves_per_simd = threads >= 256 ? threads / 256.0 : 1.0;
        printf("  %4d thr: %6.2f ns/instr/wave (%5.2f ns of the SIMD)", threads, ms * 1e6 / 5 / ITER / n, ms * 1e6 / 5 / ITER / n / waves_per_simd);
    }
    printf("\n");
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const bool extra = argc > 1;
    float* d;
    int* nd;
    hipMalloc(&d, 256 * 1024 * 4);
    hipMalloc(&nd, 4);
#define RUN(K, NAME) run(K, NAME, d, nd);
    RUN(k_mov, "v_mov_b32") RUN(k_fma, "v_fma_f32") RUN(k_cnd_vcc, "v_cndmask_b32 (vcc)") RUN(k_cnd_sg, "v_cndmask_b32_e64 (sgpr pair)")
    RUN(k_cnd_const, "v_cndmask_b32_e64 0, 1.0 (sgpr)") RUN(k_bfi, "v_bfi_b32") RUN(k_and, "v_and_b32") RUN(k_max, "v_max_f32") RUN(k_med3, "v_med3_f32")
    RUN(k_cmp, "v_cmp_gt_f32 vcc") RUN(k_cnd_fma, "cndmask + fma_f32 (per instr)") RUN(k_cmp_cnd, "cmp vcc + cndmask (per instr)")
    RUN(k_cmp_sg_cnd, "cmp sgpr + nop + cndmask (per instr)")
    RUN(k_cnd_dep, "v_cndmask_b32 dependent chain") RUN(k_fma32_dep, "v_fma_f32 dependent chain")
    RUN(k_fma64_dep, "v_fma_f64 dependent chain") RUN(k_fma64_2ch, "v_fma_f64 2 chains") RUN(k_fma64_4ch, "v_fma_f64 4 chains")
    if (!extra) return 0;
    RUN(k_sel_v1, "sel64: cmp vcc, cnd, cnd        /9") RUN(k_sel_v2, "sel64: cmp sgpr, nop1, cnd64 x2 /9") RUN(k_sel_v3, "sel64: cmp sgpr, cnd64 x2 (!)   /9")
    RUN(k_sel_v4, "sel64: cmp vcc, cnd, fma, cnd  /12") RUN(k_sel_v5, "sel64: cmp vcc, cnd_e64(vcc) x2 /9") RUN(k_sel_v6, "sel64: cmp,cnd,cmp2,cnd       /12")
    RUN(k_sel_v7, "sel64: cmp vcc, nop3, cnd, cnd  /9") RUN(k_sel_v8, "sel64: cmp vcc, cnd, nop0, cnd  /9") RUN(k_sel_v9, "sel64: cmp sgpr, 2 fma, cnd64x2 /15")
    RUN(k_cnd2_fma2, "2 cndmask(vcc) + 2 fma") RUN(k_cnd4_fma4, "4 cndmask(vcc) + 4 fma") RUN(k_cnd2sg_fma2, "2 cndmask(sgpr) + 2 fma")
    RUN(k_cmp_cnd2, "cmp + 2 cndmask(vcc)") RUN(k_cmp_cnd4, "cmp + 4 cndmask(vcc) + fma")
    RUN(k_salu8, "8 SALU") RUN(k_salu_valu, "SALU, VALU alternating") RUN(k_br_valu, "VALU + untaken branch") RUN(k_br_taken, "VALU + taken branch (9 instr)")
    RUN(k_saveexec, "cmp, saveexec, fma, or exec") RUN(k_nop8, "s_nop 0") RUN(k_readlane, "readlane + fma")
    return 0;
}
