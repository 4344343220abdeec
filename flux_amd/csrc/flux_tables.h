// flux_tables.h -- host entry points of tables.hip / render.hip used by abi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flux_device.h"

namespace flux {

// MasterSampleSets::new (sampling.rs:13-33) for the sets of `sets` (slot m = global set first + m * stride; all S of
// them for {0, 1, S}) + shuffle_indices for all H rows and all S sets (sampling.rs:35-40), generated on the device.
// Synchronises `stream`.  phase_ms (optional) += {scratch allocation, launches + wait, scratch release} in ms.
hipError_t generate_tables(uint64_t seed, uint32_t S, SetRange sets, uint32_t D, uint32_t n, uint32_t H,
                           double2 *pix, double2 *disc, double *hemi, int32_t *rowperm, int32_t *invperm,
                           hipStream_t stream, double *phase_ms = nullptr);
// One set of a samplers-crate generator (0 regular, 1 jittered, 2 multi-jittered, 3 correlated MJ) and,
// if d_hemi != nullptr, its to_hemisphere(.., 0.0) image [N][3].  Synchronises `stream`.
hipError_t generate_sampler_grid(int kind, uint64_t seed, uint32_t n, double *d_xy, double *d_hemi,
                                 hipStream_t stream);
// FAST glossy lobe factors of the pixel samples: gloss[s][i] = (cos 2 pi x, sin 2 pi x, log2(1 - y), 0) with
// flux_math.h's functions (render.hip, FAST arithmetic).  Asynchronous on `stream`.
hipError_t generate_gloss_table(const double2 *pix, size_t count, double *gloss, hipStream_t stream);
hipError_t hemi_to_aos(size_t SD, size_t N, const double *in, double *out, hipStream_t stream);

// Camera::render (trace.rs:53-97).  variant: FLUX_KERNEL_STATIC / FLUX_KERNEL_REFILL;
// math: FLUX_MATH_FAST / FLUX_MATH_STRICT (render_body.inc).
hipError_t launch_render(const RenderParams &p, int variant, int math, hipStream_t stream);
// Which kernel launch_render runs for these parameters, with what block size, grid and dynamic LDS: decided in ONE place
// (render_body.inc plan_render_impl) and asked from there by the host's LDS budget check, by flux_ctx_launch_plan /
// flux_ctx_bvh_info and by the launch itself, so that they cannot drift apart.
struct LaunchPlan {
    int kernel;  // FLUX_PLAN_* (include/flux_abi.h): 0 static, 1 refill, 2 split, 3 BVH state machine over the binary tree,
                 // 4 the same over the 4-wide tree; -1 nothing to do
    unsigned block;
    uint64_t blocks;
    size_t lds;  // dynamic LDS per block (the refill / split kernels add 96 B of static LDS)
    unsigned waves_per_pixel;  // K: waves that share one pixel's samples (1 in the static and BVH kernels)
    int lds_scene;             // kernel 4: 1 = the instantiation that keeps the analytic set's records and the materials in LDS (`lds` includes them)
};
LaunchPlan plan_render(const RenderParams &p, int variant, int math);

// Scene::shade for caller-supplied rays (device pointers; rays = n x (origin, direction)).
hipError_t launch_shade_rays(const RenderParams &p, int math, const double *d_rays, int n, int depth, uint32_t set,
                             uint32_t index, double *d_rgb, int *d_hit, double *d_t, hipStream_t stream);

// flux_math.h under test: out[i] = fn(a[i], b[i]) on the device (b may be null).
hipError_t launch_fastmath_probe(int fn, const double *a, const double *b, double *out, size_t n,
                                 hipStream_t stream);

}  // namespace flux
