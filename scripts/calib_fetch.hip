// calib_fetch.hip -- calibrates rocprofv3's FETCH_SIZE on gfx950 for THIS repo's access widths
// (MI355X_MICROARCH.md: FETCH_SIZE is exact only for some widths; "calibrate on a known byte count in
// your own access pattern").  Streams a 1 GiB buffer once per kernel with (a) 16 B per lane (the
// pixel/disc table reads, global_load_dwordx4) and (b) 8 B per lane (the hemisphere-plane reads,
// global_load_dwordx2); profiles/ records FETCH_SIZE per kernel next to the known 1 GiB.
// Build + run: scripts/profile_gpu.sh (hipcc --offload-arch=gfx950 -O3 scripts/calib_fetch.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void read16(const double2 *p, size_t n, double *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (; i < n; i += stride) { double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 1.2345e300) *sink = acc;
}
__global__ void read8(const double *p, size_t n, double *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (; i < n; i += stride) acc += p[i];
    if (acc == 1.2345e300) *sink = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    void *buf = nullptr, *flush = nullptr;
    double *sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&flush, bytes) != hipSuccess ||
        hipMalloc((void **)&sink, 8) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, bytes);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(flush, rep, bytes);  // 1 GiB of other traffic: evicts the 256 MiB Infinity Cache
        read16<<<4096, 256>>>((const double2 *)buf, bytes / 16, sink);
        (void)hipMemset(flush, rep + 2, bytes);
        read8<<<4096, 256>>>((const double *)buf, bytes / 8, sink);
    }
    hipError_t e = hipDeviceSynchronize();
    std::printf("calib_fetch: %s, each read kernel streams %zu bytes\n", hipGetErrorString(e), bytes);
    return e == hipSuccess ? 0 : 1;
}
