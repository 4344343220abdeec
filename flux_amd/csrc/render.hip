// render.hip -- the per-pixel render loop on gfx950 (MI355X), FP64.
//
// Replaces Camera::render (fluxcore/src/trace.rs:53-97) and everything it
// calls: Camera::ray_direction (trace.rs:44-51), Scene::shade / Scene::hit
// (scene.rs:156-172), BoundingBox::hit / Sphere::hit / Plane::hit
// (shapes.rs:98-217), the three path_shade impls (materials.rs:18-72), the
// three BRDFs (brdf.rs:19-79), to_unit_hemi (samplers/src/lib.rs:133-142) and
// Color::max_to_one (color.rs:35-44).
//
// Mapping to the machine (not a translation of the reference's recursion):
//  * one lane = one camera path; one BLOCK owns one pixel: its 1..4 waves take contiguous slices of the
//    pixel's samples and each walks its slice 64 at a time, so pixel/lens table reads are contiguous
//    across the wave (16 B per lane);
//  * blocks are ordered by sample set and dealt to the 8 XCDs so that the tables of the set being worked on
//    stay in that XCD's L2 (render_body.inc map_wave);
//  * the scene (<= a few dozen shapes) is read with a wave-uniform index -> scalar loads, operands sit in
//    SGPRs; only the nearest hit's record is gathered per lane;
//  * the reference's recursion  L = (f1 (*) ((f2 (*) (...)) * s2)) * s1  is run iteratively -- STRICT pushes
//    (f, s = n.wi/pdf) on a per-lane LDS stack and folds it from the deepest bounce outward (the reference's
//    multiplication order), FAST multiplies a register throughput;
//  * SPLIT kernel (FAST, analytic scenes, the default from 256 spp): PRIMARY segments -- all lanes of a wave sample
//    one pixel, so their rays are coherent -- are traced 64 at a time against the pixel's own candidate spheres
//    (a wave-uniform mask, scalar operands, uniform control flow) and shaded together; the continuing paths go
//    through a 64-entry LDS queue to the SECONDARY loop, where a lane whose path ended pops the next queued path
//    (ballot + mbcnt prefix), so lanes stay busy although path lengths differ (1..D segments);
//  * REFILL kernel (STRICT; 64..255 spp): the same refill discipline with primaries and secondaries mixed in one loop;
//    STATIC kernel (< 64 spp): lane l traces samples l, l+64, ...;
//  * BVH kernels (FAST mesh scenes): persistent lanes with a per-lane traversal state machine -- over a 4-wide tree of 64-B
//    nodes (boxes on a 16-bit grid, planes built by v_perm_b32, tested through v_pk_fma_f32) with 128-B leaf records that
//    hold both halves of a quad (render_bvh4_kernel), or over the binary tree of 32-B nodes (render_bvh_kernel, fallback);
//  * per-lane partial sums are combined in lane order, wave totals in wave order (a fixed tree => the image is
//    bit-reproducible run to run and independent of how the frame is split), then * 1/n^2, max_to_one, and the
//    3 doubles are written once.  No atomics.
//
// The loop body (render_body.inc) is compiled twice: STRICT under `#pragma clang fp contract(off)` with the
// reference's operation order, FAST under contract(fast) with flux_math.h (see render_body.inc's header).
#include "flux_device.h"
#include "flux_tables.h"
#include "flux_math.h"
#include "../../include/flux_abi.h"

// Tunables: NUMBERS only (overridable with -D, scripts/sweep_variants.py).  Every either/or of rounds 1-5 -- 45 boolean FLUX_*
// switches with one shipped value and a measured verdict -- was folded into the code in round 6 (same ISA before and after,
// profiles/r06_experiments/prune_flags/); the experiments' patches and logs stay under profiles/r0*_experiments/.  What is left
// here, in flux_device.h (FLUX_UNI_SPHERES, FLUX_BVH_WIDE_MAX_STACK, FLUX_MAX_WAVES_PER_PIXEL, FLUX_MIN_SAMPLES_PER_WAVE) and in
// bvh.cpp / flux_bvh.h (FLUX_BVH_BINS, FLUX_BVH_COLLAPSE_MODE, FLUX_BVH_LEAF), plus the six -DFLUX_DEBUG_* instrumentation hooks of
// render_body.inc (never in the product build), is the whole list.
// Block size: measured on demo2 at 1024 spp (refill kernel): 256 threads x 2 waves/SIMD 88.6 ms; 256 x 3 71.7 ms; 64 x 3 70.7 ms;
// 256 x 4 83.0 ms (spills).  Waves never cooperate, so one wave per block lets the LDS stack of a finished wave be reused at once.
#ifndef FLUX_BLOCK_THREADS
#define FLUX_BLOCK_THREADS 64
#endif
#ifndef FLUX_WAVES_PER_EU
#define FLUX_WAVES_PER_EU 3
#endif
#ifndef FLUX_WPE_BVH
#define FLUX_WPE_BVH 4            // waves/SIMD of the binary-tree BVH kernel (the fallback): 4 waves, nothing spilled -- at 5 it spills 11 VGPRs
#endif
#ifndef FLUX_WPE_BVH4
#define FLUX_WPE_BVH4 5           // waves/SIMD of render_bvh4_kernel: 96 VGPRs, nothing spilled, since round 4's register diet (before: 122 at 4)
#endif
#ifndef FLUX_BVH4_EARLY_AT
#define FLUX_BVH4_EARLY_AT 48     // render_bvh4_kernel leaves its node loop for the shading step as soon as this many walks have ended
#endif
#ifndef FLUX_BVH_REFILL_AT
#define FLUX_BVH_REFILL_AT 40     // lanes that must be waiting for shading before the wave leaves traversal (swept 16..64 with the leaf vote)
#endif
// the early exit leaves the node loop when EARLY_AT walks have ended and expects the shading step to follow: with
// EARLY_AT < REFILL_AT the wave would re-enter the loop on the same counts and never advance
static_assert(FLUX_BVH4_EARLY_AT >= FLUX_BVH_REFILL_AT,
              "FLUX_BVH4_EARLY_AT must not be below FLUX_BVH_REFILL_AT (render_bvh4_kernel would livelock)");
#ifndef FLUX_WPE_SPLIT
#define FLUX_WPE_SPLIT 5          // waves/SIMD of the split kernel: 96 VGPRs, nothing spilled since round 4 (4 until then: 128 VGPRs); demo2 @16384 spp 250.0 -> 225.4 ms
#endif
#ifndef FLUX_BVH_LEAF_NUM
#define FLUX_BVH_LEAF_NUM 2       // binary-tree kernel: the inner-node loop is left once the lanes holding a leaf outweigh the descending ones,
#define FLUX_BVH_LEAF_DEN 3       //     n_leaf * NUM > n_inner * DEN (swept: 1:1 178.4, 2:3 175.9, 1:2 176.4, 1:3 179.7, 3:2 178.8 ms at 1024 spp)
#endif
#ifndef FLUX_STRICT_SCAN_UNROLL
#define FLUX_STRICT_SCAN_UNROLL 4  // STRICT shape scan: records fetched this many at a time (scalar loads issued together); demo2 @16384 spp 1035 -> 1026 ms
#endif
#ifndef FLUX_WAVES_PER_EU_FAST
#define FLUX_WAVES_PER_EU_FAST 5  // FAST render_refill_kernel without meshes: 94 VGPRs, nothing spilled
#endif
#ifndef FLUX_WAVES_PER_EU_FAST_WIDE
#define FLUX_WAVES_PER_EU_FAST_WIDE 4  // FAST render_static_kernel (12 VGPRs spilled under the 5-wave cap) and the mesh instantiations of
#endif                                 // render_refill_kernel (15 spilled): 4 waves/SIMD, nothing spilled (round 5; they shared the cap tuned
                                       // for the refill kernel in round 1)


// The loop itself lives in render_body.inc and is compiled twice (see its header): the STRICT
// arithmetic (reference operation order, no contraction) and the FAST arithmetic (FMA + flux_math.h).
#define FLUX_FAST 0
#define FLUX_WPE FLUX_WAVES_PER_EU
#define FLUX_WPE_WIDE FLUX_WAVES_PER_EU
#pragma clang fp contract(off)
namespace flux {
namespace strict {
#include "render_body.inc"
}  // namespace strict
}  // namespace flux
#undef FLUX_FAST
#undef FLUX_WPE
#undef FLUX_WPE_WIDE

#define FLUX_FAST 1
#define FLUX_WPE FLUX_WAVES_PER_EU_FAST
#define FLUX_WPE_WIDE FLUX_WAVES_PER_EU_FAST_WIDE
#pragma clang fp contract(fast)
namespace flux {
namespace fast {
#include "render_body.inc"
}  // namespace fast
}  // namespace flux
// FAST glossy-lobe factors of every pixel sample (RenderParams::gloss), compiled with the FAST arithmetic
// so that the table holds bit for bit what to_unit_hemi would compute inline.
namespace flux {
__global__ void gloss_fill_kernel(const double2 *__restrict__ pix, size_t count, double *__restrict__ gloss) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const double2 p = pix[t];
    double s, c;
    fastmath::fsincos2pi(p.x, s, c);
    double *o = gloss + t * 4;
    o[0] = c;
    o[1] = s;
    o[2] = fastmath::flog2(1.0 - p.y);
    o[3] = 0.0;
}
hipError_t generate_gloss_table(const double2 *pix, size_t count, double *gloss, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    gloss_fill_kernel<<<dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream>>>(pix, count, gloss);
    return hipGetLastError();
}
}  // namespace flux
#undef FLUX_FAST
#undef FLUX_WPE
#undef FLUX_WPE_WIDE
#pragma clang fp contract(off)

namespace flux {

hipError_t launch_render(const RenderParams &p, int variant, int math, hipStream_t stream) {
    if (math == FLUX_MATH_STRICT) return strict::launch_render_impl(p, variant, stream);
    return fast::launch_render_impl(p, variant, stream);
}

// the kernel, grid and LDS launch_render would pick (host-side budget check, flux_ctx_launch_plan)
LaunchPlan plan_render(const RenderParams &p, int variant, int math) {
    return math == FLUX_MATH_STRICT ? strict::plan_render_impl(p, variant) : fast::plan_render_impl(p, variant);
}

hipError_t launch_shade_rays(const RenderParams &p, int math, const double *d_rays, int n, int depth, uint32_t set,
                             uint32_t index, double *d_rgb, int *d_hit, double *d_t, hipStream_t stream) {
    if (math == FLUX_MATH_STRICT) return strict::launch_shade_rays_impl(p, d_rays, n, depth, set, index, d_rgb, d_hit, d_t, stream);
    return fast::launch_shade_rays_impl(p, d_rays, n, depth, set, index, d_rgb, d_hit, d_t, stream);
}

// ---- flux_math.h under test: out[i] = fn(a[i], b[i]) computed on the device ----------------------
__global__ void fastmath_probe_kernel(int fn, const double *a, const double *b, double *out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i], y = b ? b[i] : 0.0;
    double r = 0.0, s, c;
    switch (fn) {
        case 0: r = fastmath::frsqrt(x); break;
        case 1: r = fastmath::fsqrt(x); break;
        case 2: r = fastmath::fdiv(x, y); break;
        case 3: r = fastmath::flog2(x); break;
        case 4: r = fastmath::fexp2(x); break;
        case 5: r = fastmath::fpow_pos(x, y); break;
        case 6: fastmath::fsincos2pi(x, s, c); r = s; break;
        case 7: fastmath::fsincos2pi(x, s, c); r = c; break;
        case 8: r = __builtin_amdgcn_rsq(x); break;   // raw hardware seeds, for the record
        case 9: r = __builtin_amdgcn_rcp(x); break;
        default: break;
    }
    out[i] = r;
}

hipError_t launch_fastmath_probe(int fn, const double *a, const double *b, double *out, size_t n,
                                 hipStream_t stream) {
    if (n == 0) return hipSuccess;
    fastmath_probe_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(fn, a, b, out, n);
    return hipGetLastError();
}

}  // namespace flux
