// gather_rate.hip -- what a DIVERGENT 16-B gather costs the vector-memory pipeline as a function of the number of ACTIVE lanes:
// does the address unit charge per wave-instruction or per active lane?  (render_bvh_kernel runs its node step at ~45 % of
// lanes active and is co-limited by this pipeline: DESIGN.md section 5.)
// Every wave issues ROUNDS x 8 independent random gathers (uint4) from a table; only lanes < ACTIVE take part (the others
// skip the loads through EXEC).  Tables: 2 MiB (L2-resident), 32 MiB (the BVH's node array: L2 + Infinity Cache), 160 MiB.
// Prints ns per wave-instruction per CU and lane-gathers per ns chip-wide.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

template <int PIECES>
__global__ __launch_bounds__(256) void gather(const uint4 *__restrict__ tab, uint32_t mask, int active, int rounds, uint32_t *sink) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    if ((int)lane < active) {
        for (int r = 0; r < rounds; ++r) {
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s = s * 1664525u + 1013904223u;
                const uint32_t rec = (s >> 7) & mask;  // a 32-B record
#pragma unroll
                for (int p = 0; p < PIECES; ++p) {
                    const uint4 w = tab[(size_t)rec * 2 + p];
                    if (p == 0) v[k] = w; else v[k].x ^= w.y;
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k].x ^ v[k].w;
        }
    }
    if (acc == 0x1234567u) *sink = acc;
}

int main() {
    const size_t big = (size_t)160 << 20;
    uint4 *tab = nullptr;
    uint32_t *sink = nullptr;
    if (hipMalloc((void **)&tab, big) != hipSuccess || hipMalloc((void **)&sink, 4) != hipSuccess) return 1;
    (void)hipMemset(tab, 7, big);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int blocks = 256 * 8, rounds = 64;  // 8 blocks of 4 waves per CU = 8 waves / SIMD
    const size_t sizes[3] = {(size_t)2 << 20, (size_t)32 << 20, (size_t)128 << 20};
    for (int pieces = 1; pieces <= 2; ++pieces)
        for (size_t sz : sizes)
            for (int active : {64, 48, 32, 16, 8, 4}) {
                const uint32_t mask = (uint32_t)(sz / 32 - 1);
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    (void)hipEventRecord(e0);
                    if (pieces == 1) gather<1><<<blocks, 256>>>(tab, mask, active, rounds, sink);
                    else gather<2><<<blocks, 256>>>(tab, mask, active, rounds, sink);
                    (void)hipEventRecord(e1);
                    (void)hipEventSynchronize(e1);
                    float ms = 0;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                const double insts = (double)blocks * 4 * rounds * 8 * pieces;  // wave-level gather instructions
                const double per_cu_ns = best * 1e6 / (insts / 256.0);
                std::printf("pieces %d table %4zu MiB active %2d: %8.3f ms  %7.2f ns per wave-gather per CU (%6.1f cycles at 2.4 GHz)  %7.2f lane-gathers/ns chip\n",
                            pieces, sz >> 20, active, best, per_cu_ns, per_cu_ns * 2.4, insts * active / (best * 1e6));
            }
    return 0;
}
