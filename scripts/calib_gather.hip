// calib_gather.hip -- what rocprofv3's FETCH_SIZE (and any DRAM-side counter the box offers) reports for the access
// pattern of render_bvh_kernel: DIVERGENT per-lane 16-B gathers (two per 32-B node) and per-lane reads of a 128-B
// record (the triangle test reads 80 B of one), NOT the wave-coalesced streams scripts/calib_fetch.hip calibrates
// (the streaming x2 must not be reused for gathers, and FETCH_SIZE sits on the fabric side of
// the L2, so Infinity-Cache hits may be in it).
//
// Every kernel touches each 128-B line of its buffer EXACTLY ONCE, in a scattered order (line = i * odd mod 2^k: a
// bijection), so the bytes that must cross the L2's memory side are known: lines x (the fetch granule).  Buffers:
//   small = 128 MiB (2^20 lines): fits the 256 MiB Infinity Cache, like config 5's 176 MB working set
//   large =   1 GiB (2^23 lines): does not
// Each pattern runs COLD (right after 2 GiB of other traffic) and WARM (the same launch repeated): if FETCH_SIZE is the
// same warm as cold on the small buffer, the counter includes Infinity-Cache hits and says nothing about HBM.
// Kernel names carry pattern, buffer and temperature so the counter CSV is self-describing.
// Build + run: scripts/calib_gather.sh (hipcc --offload-arch=gfx950 -O3).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr uint32_t kMult = 2654435761u;  // odd: i -> i * kMult mod 2^k is a bijection

template <int TAG>
__global__ __launch_bounds__(256) void flush_read(const uint4 *__restrict__ p, size_t n, uint32_t *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i].x;
    if (acc == 0x12345u) *sink = acc;
}

// distinct kernel NAMES per (pattern, buffer, temperature)
#define DEF(NAME, PIECES, TAG) \
    __global__ __launch_bounds__(256) void NAME(const uint4 *__restrict__ b, uint32_t m, uint32_t *s) { \
        const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;                                        \
        const uint32_t line = (i * kMult) & m;                                                           \
        const uint4 *p = b + (size_t)line * 8;                                                           \
        uint32_t acc = 0;                                                                                \
        _Pragma("unroll") for (int k = 0; k < PIECES; ++k) {                                             \
            const uint4 v = p[(k + (line & 7)) & 7];                                                     \
            acc += v.x ^ v.y ^ v.z ^ v.w;                                                                \
        }                                                                                                \
        if (acc == 0x12345u) *s = acc;                                                                   \
    }
DEF(g16_small_cold, 1, 0)
DEF(g16_small_warm, 1, 1)
DEF(g32_small_cold, 2, 0)
DEF(g32_small_warm, 2, 1)
DEF(g80_small_cold, 5, 0)
DEF(g80_small_warm, 5, 1)
DEF(g128_small_cold, 8, 0)
DEF(g128_small_warm, 8, 1)
DEF(g16_large_cold, 1, 0)
DEF(g16_large_warm, 1, 1)
DEF(g128_large_cold, 8, 0)
DEF(g128_large_warm, 8, 1)

int main() {
    const size_t small_b = (size_t)128 << 20, large_b = (size_t)1 << 30, flush_b = (size_t)2 << 30;
    uint4 *small = nullptr, *large = nullptr, *fl = nullptr;
    uint32_t *sink = nullptr;
    if (hipMalloc((void **)&small, small_b) != hipSuccess || hipMalloc((void **)&large, large_b) != hipSuccess ||
        hipMalloc((void **)&fl, flush_b) != hipSuccess || hipMalloc((void **)&sink, 4) != hipSuccess) {
        std::printf("alloc failed\n");
        return 1;
    }
    (void)hipMemset(small, 1, small_b);
    (void)hipMemset(large, 2, large_b);
    (void)hipMemset(fl, 3, flush_b);
    const uint32_t ms = (uint32_t)(small_b / 128 - 1), ml = (uint32_t)(large_b / 128 - 1);
    const dim3 gs((unsigned)(small_b / 128 / 256)), gl((unsigned)(large_b / 128 / 256)), b(256);
    auto flush = [&]() { flush_read<0><<<8192, 256>>>(fl, flush_b / 16, sink); };
#define PAIR(COLD, WARM, BUF, MASK, GRID) \
    flush();                              \
    COLD<<<GRID, b>>>(BUF, MASK, sink);   \
    WARM<<<GRID, b>>>(BUF, MASK, sink);
    for (int rep = 0; rep < 2; ++rep) {
        PAIR(g16_small_cold, g16_small_warm, small, ms, gs)
        PAIR(g32_small_cold, g32_small_warm, small, ms, gs)
        PAIR(g80_small_cold, g80_small_warm, small, ms, gs)
        PAIR(g128_small_cold, g128_small_warm, small, ms, gs)
        PAIR(g16_large_cold, g16_large_warm, large, ml, gl)
        PAIR(g128_large_cold, g128_large_warm, large, ml, gl)
    }
    const hipError_t e = hipDeviceSynchronize();
    std::printf("calib_gather: %s; small = %zu lines of 128 B (%zu B), large = %zu lines (%zu B); every kernel touches each "
                "line of its buffer once\n", hipGetErrorString(e), small_b / 128, small_b, large_b / 128, large_b);
    return e == hipSuccess ? 0 : 1;
}
