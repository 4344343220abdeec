// flux_ctx.h -- the context behind include/flux_abi.h's opaque flux_ctx, shared by abi.hip (the single-device entry
// points) and multi.hip (the multi-GPU frame).  Internal: nothing here crosses the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/flux_abi.h"
#include "flux_device.h"
#include "flux_tables.h"

namespace flux {

// message of the calling thread's last failed call (flux_last_error); fail() sets it and returns `code`
extern thread_local std::string g_last_error;
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace flux

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return flux::fail(e_ == hipErrorOutOfMemory ? FLUX_E_NOMEM : FLUX_E_DEVICE, "%s: %s", \
                              #expr, hipGetErrorString(e_));                                      \
    } while (0)

struct flux_ctx {
    int device = 0;
    flux::RenderParams rp{};  // camera + table pointers; work fields set per launch
    uint64_t seed = 0;
    uint32_t n = 0, N = 0, D = 0, S = 0, W = 0, H = 0;
    flux::SetRange sets{0, 1, 0};  // sets with tables in this context (all S unless created by flux_ctx_create_sets)
    flux::DevShape *d_shapes = nullptr;
    flux::DevMaterial *d_mats = nullptr;
    unsigned char *d_fscene = nullptr;  // FAST path: scan spheres | scan planes | hit records | f32 filter spheres
    double2 *d_pix = nullptr, *d_disc = nullptr;
    double *d_hemi = nullptr;
    double *d_gloss = nullptr;  // FAST glossy-lobe factors of pixel_sets
    flux::DevSetRows *d_setrows = nullptr;  // per table slot: where the set's rows of the four tables start
    int32_t *d_rowperm = nullptr, *d_invperm = nullptr;
    unsigned long long *d_stats = nullptr;
    bool stats_on = false;
    // extension: triangle meshes
    flux::DevTri *d_tris = nullptr;
    flux::DevNode *d_nodes = nullptr;
    flux::DevNode4Q *d_nodes4 = nullptr;
    flux::DevLeafRec *d_leaves = nullptr;
    flux::DevNodeQ *d_nodesq = nullptr;
    flux::BvhInfo bvh{};
    int traversal = FLUX_TRAVERSE_BVH;
    int variant = FLUX_KERNEL_DEFAULT;
    int math = FLUX_MATH_FAST;
    // scratch framebuffer for the host-output path
    double *d_out = nullptr;
    size_t d_out_doubles = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    uint64_t device_bytes = 0;
    double U[3], V[3], Wv[3];
    // where flux_ctx_create's wall time went (flux_ctx_create_timing), milliseconds
    double create_ms[FLUX_CREATE_TIMING_WORDS] = {};
};
