// cold_start.hip -- what a process's FIRST HIP calls cost on this box, one by one: the part of flux_ctx_create's cold time that is
// the runtime's lazy initialisation rather than work of the context (DESIGN.md "Context creation").
//   hipcc --offload-arch=gfx950 -O2 cold_start.hip -o cold_start -L../../flux_amd -lflux_hip -Wl,-rpath,$PWD/../../flux_amd
//   ./cold_start [order]     order: 0 = copy first, then kernels; 1 = own kernel first; 2 = the big module's kernel first;
//                                   3 = copy on a second thread beside the big module's first kernel
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../include/flux_abi.h"

__global__ void tiny(int *p) { p[threadIdx.x] = threadIdx.x; }

static std::chrono::steady_clock::time_point t_last;
static void lap(const char *what) {
    const auto now = std::chrono::steady_clock::now();
    std::printf("  %-58s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = std::chrono::steady_clock::now();
}

int main(int argc, char **argv) {
    const int order = argc > 1 ? std::atoi(argv[1]) : 0;
    std::printf("order %d\n", order);
    const auto t0 = std::chrono::steady_clock::now();
    t_last = t0;
    int n = 0;
    (void)hipGetDeviceCount(&n);
    lap("hipGetDeviceCount (hipInit)");
    (void)hipSetDevice(0);
    lap("hipSetDevice(0)");
    (void)hipFree(nullptr);
    lap("hipFree(nullptr)");
    int *d = nullptr;
    (void)hipMalloc((void **)&d, 1 << 20);
    lap("hipMalloc 1 MiB");
    void *big = nullptr;
    (void)hipMalloc(&big, (size_t)3 << 30);
    lap("hipMalloc 3 GiB");
    std::vector<int> h(1024, 7);
    auto copy = [&] {
        (void)hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    };
    auto own_kernel = [&] {
        tiny<<<1, 64>>>(d);
        (void)hipDeviceSynchronize();
    };
    std::vector<double> xy(2 * 16);
    auto big_kernel = [&] { (void)flux_sampler_grid(0, 0, 4, 1, xy.data(), nullptr); };  // a kernel of libflux_hip.so's code object
    if (order == 0) {
        copy(); lap("first hipMemcpy H2D 4 KiB (pageable)");
        copy(); lap("second hipMemcpy H2D");
        own_kernel(); lap("first launch + sync: this file's kernel");
        own_kernel(); lap("second launch + sync");
        big_kernel(); lap("first launch: libflux_hip.so's module (flux_sampler_grid)");
        big_kernel(); lap("second flux_sampler_grid");
    } else if (order == 1) {
        own_kernel(); lap("first launch + sync: this file's kernel");
        copy(); lap("first hipMemcpy H2D 4 KiB (pageable)");
        big_kernel(); lap("first launch: libflux_hip.so's module (flux_sampler_grid)");
    } else if (order == 2) {
        big_kernel(); lap("first launch: libflux_hip.so's module (flux_sampler_grid)");
        copy(); lap("first hipMemcpy H2D 4 KiB (pageable)");
        own_kernel(); lap("first launch + sync: this file's kernel");
    } else {
        std::thread t([&] { (void)hipSetDevice(0); copy(); });
        big_kernel();
        t.join();
        lap("first copy on a thread beside the big module's first launch");
        copy(); lap("second hipMemcpy H2D");
        big_kernel(); lap("second flux_sampler_grid");
    }
    void *pinned = nullptr;
    (void)hipHostMalloc(&pinned, 1 << 20, 0);
    lap("hipHostMalloc 1 MiB");
    (void)hipMemcpy(d, pinned, 4096, hipMemcpyHostToDevice);
    lap("hipMemcpy H2D from pinned");
    hipStream_t s;
    (void)hipStreamCreate(&s);
    lap("hipStreamCreate");
    hipEvent_t e;
    (void)hipEventCreate(&e);
    lap("hipEventCreate");
    (void)hipFree(big);
    lap("hipFree 3 GiB");
    std::printf("  total %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return 0;
}
