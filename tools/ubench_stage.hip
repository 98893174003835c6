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
    return 0;
}
