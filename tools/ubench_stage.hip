// ubench_stage.hip - how long does it take every CU to copy the SAME 152 KB weight image from L2 into its LDS (the prologue of the
// policy / learner kernels), and does the ORDER in which the workgroups walk the image matter?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_stage.hip -o /tmp/ubench_stage && /tmp/ubench_stage
// mode 0: every workgroup copies chunk k = 0, 1, 2, ... (what stage_image / stage_copy did until round 4)
// mode 1: workgroup b starts at chunk (b * 5) mod PER and wraps (the workgroups of an XCD address different 4 KB chunks at any moment)
// mode 2: as 0, but only ONE workgroup per XCD... (grid of 8): the no-contention reference
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr uint32_t IMG_BYTES = 152064, NVEC = IMG_BYTES / 16;

template <uint32_t THREADS, int MODE>
__global__ void __launch_bounds__(THREADS, 1) stage_kernel(const uint4* __restrict__ src, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint4* d = reinterpret_cast<uint4*>(lds);
    constexpr uint32_t PER = (NVEC + THREADS - 1u) / THREADS;
    const uint32_t tid = threadIdx.x;
    const uint32_t rot = MODE == 1 ? (blockIdx.x * 5u) % PER : 0u;
    uint4 v[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        uint32_t kk = k + rot; if (kk >= PER) kk -= PER;
        const uint32_t c = kk * THREADS + tid;
        v[k] = c < NVEC ? src[c] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        uint32_t kk = k + rot; if (kk >= PER) kk -= PER;
        const uint32_t c = kk * THREADS + tid;
        if (c < NVEC) d[c] = v[k];
    }
    __syncthreads();
    if (out && tid == 0) out[blockIdx.x] = reinterpret_cast<uint32_t*>(lds)[(blockIdx.x * 64u) % (IMG_BYTES / 4u)];
}

__global__ void empty_kernel(uint32_t* out) { if (out && threadIdx.x == 999) out[0] = 1; }

// the learner's case: the image is REWRITTEN (by the optimizer kernel, from CUs of every XCD) between two stagings, so no XCD's L2
// holds it when the next kernel's workgroups ask for it.  how: 0 = plain 2-byte stores (what learner_adam_kernel does), 1 = plain
// 16-byte stores, 2 = 16-byte stores with the nt policy, 3 = 16-byte sc1 (agent-scope write-through) stores
template <int HOW>
__global__ void __launch_bounds__(256) rewrite_kernel(uint4* img, uint32_t salt) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (HOW == 0) {
        uint16_t* h = reinterpret_cast<uint16_t*>(img);
        if (i < IMG_BYTES / 2u) h[i] = (uint16_t)(i + salt);
    } else if (i < NVEC) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = {i, salt, i ^ salt, 7u};
        u32x4* q = reinterpret_cast<u32x4*>(img) + i;
        if (HOW == 1) *q = v;
        else if (HOW == 2) __builtin_nontemporal_store(v, q);
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(q), "v"(v) : "memory");
    }
}

template <typename F> static float time_us(F launch, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    uint4* src; uint32_t* out;
    hipMalloc(&src, IMG_BYTES); hipMalloc(&out, 4096 * 4);
    hipMemset(src, 1, IMG_BYTES);
    const int reps = 400;
#define ATTR(K) hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, (int)IMG_BYTES + 1024)
    ATTR((stage_kernel<256, 0>)); ATTR((stage_kernel<256, 1>)); ATTR((stage_kernel<512, 0>)); ATTR((stage_kernel<512, 1>));
    const float t_empty = time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, out); }, reps);
    printf("empty kernel, 256 x 256 threads, back to back: %.2f us per launch\n", t_empty);
    for (int grid : {8, 32, 256}) {
        const float a = time_us([&] { hipLaunchKernelGGL((stage_kernel<256, 0>), dim3(grid), dim3(256), IMG_BYTES + 1024, 0, src, out); }, reps);
        const float b = time_us([&] { hipLaunchKernelGGL((stage_kernel<256, 1>), dim3(grid), dim3(256), IMG_BYTES + 1024, 0, src, out); }, reps);
        const float c = time_us([&] { hipLaunchKernelGGL((stage_kernel<512, 0>), dim3(grid), dim3(512), IMG_BYTES + 1024, 0, src, out); }, reps);
        const float e = time_us([&] { hipLaunchKernelGGL((stage_kernel<512, 1>), dim3(grid), dim3(512), IMG_BYTES + 1024, 0, src, out); }, reps);
        printf("grid %3d: 256 threads same order %.2f us, rotated %.2f us; 512 threads same order %.2f us, rotated %.2f us (per launch, back to back)\n",
               grid, a, b, c, e);
    }
    // cold case: rewrite + stage back to back, minus rewrite alone
    uint32_t salt = 0;
    auto cold = [&](int how, int threads, int mode) {
        auto rw = [&] {
            ++salt;
            if (how == 0) hipLaunchKernelGGL((rewrite_kernel<0>), dim3((IMG_BYTES / 2 + 255) / 256), dim3(256), 0, 0, src, salt);
            else if (how == 1) hipLaunchKernelGGL((rewrite_kernel<1>), dim3((NVEC + 255) / 256), dim3(256), 0, 0, src, salt);
            else if (how == 2) hipLaunchKernelGGL((rewrite_kernel<2>), dim3((NVEC + 255) / 256), dim3(256), 0, 0, src, salt);
            else hipLaunchKernelGGL((rewrite_kernel<3>), dim3((NVEC + 255) / 256), dim3(256), 0, 0, src, salt);
        };
        const float t_rw = time_us(rw, reps);
        const float t_both = time_us([&] {
            rw();
            if (threads == 256 && mode == 0) hipLaunchKernelGGL((stage_kernel<256, 0>), dim3(256), dim3(256), IMG_BYTES + 1024, 0, src, out);
            else if (threads == 256) hipLaunchKernelGGL((stage_kernel<256, 1>), dim3(256), dim3(256), IMG_BYTES + 1024, 0, src, out);
            else if (mode == 0) hipLaunchKernelGGL((stage_kernel<512, 0>), dim3(256), dim3(512), IMG_BYTES + 1024, 0, src, out);
            else hipLaunchKernelGGL((stage_kernel<512, 1>), dim3(256), dim3(512), IMG_BYTES + 1024, 0, src, out);
        }, reps);
        printf("image rewritten (how %d) before every staging, grid 256, %d threads, %s order: rewrite %.2f us, rewrite + stage %.2f us -> stage %.2f us\n", how,
               threads, mode ? "rotated" : "same", t_rw, t_both, t_both - t_rw);
    };
    for (int how = 0; how < 4; ++how) { cold(how, 256, 0); cold(how, 256, 1); cold(how, 512, 0); }
    return 0;
}
