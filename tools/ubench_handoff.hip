// ubench_handoff.hip - round-trip time of a tagged-granule ping-pong between wave pairs of ONE dispatch on gfx950, per store flavour.
// Blocks [0, B) are "servers", blocks [B, 2B) "drivers"; lane l of server block b and lane l of driver block (b + shift) % B play
// ROUNDS rounds of: driver stores ping[tag] -> server sees it, stores pong[tag] -> driver sees it.  Every spin is bounded.
//   mode 0: sc1 stores, sc1 loads                      (agent scope: what the tick server uses; placement-independent)
//   mode 1: plain stores, sc1 loads                    (visible only through a shared L2: same-XCD pairs)
//   mode 2: both stores (two buffers), both polled     (placement-independent, but every poll waits for the slow load)
//   mode 3: both stores; the L2 copy is polled `m` times for every poll of the sc1 copy (placement-independent AND L2-fast when
//           the pair shares an XCD; a pair that does not pays m wasted L2 polls per hop: slower, not wrong)
// shift = 0 pairs block b with block B + b (same XCD when B % 8 == 0: blocks are observed to go to XCD id % 8); shift = 1 pairs
// across XCDs.  Build: hipcc --offload-arch=gfx950 -O3 -o ubench_handoff tools/ubench_handoff.hip ; run: ./ubench_handoff
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint64_t ld_sc1(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_plain(uint64_t* p, uint64_t v) { asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }

// wait until *a (or *b, if given) carries `tag` in its upper 24 bits; returns the granule or 0 after max_polls
__device__ __forceinline__ uint64_t wait_tag(const uint64_t* a, const uint64_t* b, uint64_t tag, uint32_t max_polls, uint32_t* local_wins, uint32_t m = 0) {
    bool ok = false;
    uint64_t g = 0;
    for (uint32_t polls = 0; polls < max_polls; ++polls) {
        if (!ok) {
            const bool far = m == 0 || (polls % (m + 1u)) == m;         // (wave-uniform)
            const uint64_t ga = far ? ld_sc1(a) : 0;
            const uint64_t gb = b ? ld_sc1(b) : 0;
            if ((gb >> 40) == tag) { g = gb; ok = true; if (local_wins) ++*local_wins; }
            else if ((ga >> 40) == tag) { g = ga; ok = true; }
        }
        if (__all(ok)) return g;
    }
    return ok ? g : 0;
}

__global__ void __launch_bounds__(64)
pingpong(int mode, int shift, int rounds, uint64_t* ping, uint64_t* pong, uint64_t* ping_l, uint64_t* pong_l, uint32_t* fails, uint32_t* wins,
         uint32_t* xcc, uint32_t max_polls, uint32_t m) {
    const uint32_t B = gridDim.x >> 1, lane = threadIdx.x;
    const bool server = blockIdx.x < B;
    const uint32_t b = server ? blockIdx.x : (blockIdx.x - B + (uint32_t)shift) % B;      // the server block of this pair
    const uint32_t i = b * 64u + lane;
    if (lane == 0) xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 0xF;     // HW_REG_XCC_ID[3:0]
    uint32_t local_wins = 0;
    uint64_t acc = 0;
    for (int r = 1; r <= rounds; ++r) {
        const uint64_t tag = (uint64_t)r;
        if (!server) {
            const uint64_t v = (tag << 40) | (acc & 0xFFFFFFFFull);
            if (mode != 0) st_plain(ping_l + i, v);
            if (mode != 1) st_sc1(ping + i, v);
            const uint64_t g = wait_tag(mode == 1 ? pong_l + i : pong + i, mode >= 2 ? pong_l + i : nullptr, tag, max_polls, &local_wins, mode == 3 ? m : 0u);
            if (g == 0) { if (lane == 0) atomicAdd(&fails[1], 1u); break; }
            acc = g + 1;
        } else {
            const uint64_t g = wait_tag(mode == 1 ? ping_l + i : ping + i, mode >= 2 ? ping_l + i : nullptr, tag, max_polls, &local_wins, mode == 3 ? m : 0u);
            if (g == 0) { if (lane == 0) atomicAdd(&fails[0], 1u); break; }
            const uint64_t v = (tag << 40) | ((g + 3) & 0xFFFFFFFFull);
            if (mode != 0) st_plain(pong_l + i, v);
            if (mode != 1) st_sc1(pong + i, v);
        }
    }
    if (lane == 0) atomicAdd(&wins[server ? 0 : 1], local_wins);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int B : {64, 1024}) {
        const size_t n = (size_t)B * 64;
        uint64_t* buf; uint32_t *fails, *wins, *xcc;
        CHECK(hipMalloc(&buf, 4 * n * 8)); CHECK(hipMalloc(&fails, 8)); CHECK(hipMalloc(&wins, 8)); CHECK(hipMalloc(&xcc, 2 * B * 4));
        for (int shift : {0, 1})
            for (int mm : {0, 1, 2, 3, 4, 6, 10}) {
                const int mode = mm <= 2 ? mm : 3;
                const uint32_t m = mm <= 2 ? 0u : (uint32_t)(mm == 3 ? 2 : mm == 4 ? 4 : mm == 6 ? 8 : 16);
                CHECK(hipMemset(buf, 0, 4 * n * 8)); CHECK(hipMemset(fails, 0, 8)); CHECK(hipMemset(wins, 0, 8));
                const uint32_t max_polls = (mode == 1 && shift == 1) ? 20000u : 2000000u;      // (expected to fail: keep it short)
                const int r = (mode == 1 && shift == 1) ? 3 : rounds;
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(pingpong, dim3(2 * B), dim3(64), 0, 0, mode, shift, r, buf, buf + n, buf + 2 * n, buf + 3 * n, fails, wins, xcc, max_polls, m);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
                uint32_t f[2], w[2]; std::vector<uint32_t> x(2 * B);
                CHECK(hipMemcpy(f, fails, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(w, wins, 8, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(x.data(), xcc, 2 * B * 4, hipMemcpyDeviceToHost));
                int same = 0;
                for (int b = 0; b < B; ++b) same += x[b] == x[B + (b + B - shift) % B];
                printf("B=%4d shift=%d mode=%d m=%2u rounds=%5d: %8.3f us/round  fails server/driver %u/%u  same-XCD pairs %d/%d  local-copy wins per lane-round: server %.3f driver %.3f\n",
                       B, shift, mode, m, r, ms * 1e3 / r, f[0], f[1], same, B, w[0] / (double)(n * r), w[1] / (double)(n * r));
                fflush(stdout);
            }
        CHECK(hipFree(buf)); CHECK(hipFree(fails)); CHECK(hipFree(wins)); CHECK(hipFree(xcc));
    }
    return 0;
}
