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

// Tunables (overridable with -D for experiments, scripts/sweep_variants.py).  Measured on demo2 at
// 1024 spp (refill kernel): 256 threads x 2 waves/SIMD 88.6 ms; 256 x 3 71.7 ms; 64 x 3 70.7 ms;
// 256 x 4 83.0 ms (spills).  Waves never cooperate, so one wave per block lets the LDS stack of a
// finished wave be reused at once.
#ifndef FLUX_BLOCK_THREADS
#define FLUX_BLOCK_THREADS 64
#endif
#ifndef FLUX_WAVES_PER_EU
#define FLUX_WAVES_PER_EU 3
#endif
#ifndef FLUX_WPE_BVH
#define FLUX_WPE_BVH 4            // waves/SIMD of the binary-tree BVH kernel (the fallback): 4 waves, nothing spilled -- at 5 it spills 11 VGPRs
#endif
#ifndef FLUX_BVH4_PERM
#define FLUX_BVH4_PERM 1          // render_bvh4_kernel's slab test: v_perm + magic-number planes + v_pk_fma_f32 (0: rotate + convert)
#endif
#ifndef FLUX_BVH4_SORT
#define FLUX_BVH4_SORT 0          // render_bvh4_kernel: 1 = hit children fully sorted by entry distance; 0 = only the nearest is singled
#endif                            //   out and the others stacked as they come (1024 spp: 165.7 -> 159.6 ms; 0.6 % more node visits)
#ifndef FLUX_WPE_BVH4
#define FLUX_WPE_BVH4 5           // waves/SIMD of render_bvh4_kernel: 96 VGPRs, nothing spilled, since round 4's register diet (before: 122 at 4)
#endif
#ifndef FLUX_BVH4_WAVE_TOTAL
#define FLUX_BVH4_WAVE_TOTAL 1    // render_bvh4_kernel: 1 = a pixel's sum is kept per wave in scalar registers (6 VGPRs less, ~75 VALU instructions per shading pass more)
#endif
#ifndef FLUX_SHADE_TWO_PHASE
#define FLUX_SHADE_TWO_PHASE 1    // FAST shade_hit: the bounce (and the only update of the loop-carried path state) under ONE `if` after the join of the
#endif                            //   miss / emitter exits (render_body.inc shade_hit_fast), instead of early returns
#ifndef FLUX_RELOAD_PARAMS
#define FLUX_RELOAD_PARAMS 1      // render_refill_kernel, render_bvh_kernel: the same re-read of the kernel arguments per pass
#endif
#ifndef FLUX_SCALAR_VOTES
#define FLUX_SCALAR_VOTES 1  // wave votes on boolean expressions written as scalar arithmetic on the lane masks of their compares (render_body.inc)
#endif
#ifndef FLUX_BVH4_TYP
#define FLUX_BVH4_TYP 1  // render_bvh4_kernel: an instantiation with the usual analytic set's flags as compile-time constants
#endif
#ifndef FLUX_SPLIT_TYP
#define FLUX_SPLIT_TYP 1  // render_split_kernel: a third instantiation with the usual scene's flags as compile-time constants
#endif
#ifndef FLUX_SPLIT_MAX32
#define FLUX_SPLIT_MAX32 1  // render_split_kernel: a second instantiation for scenes of at most 32 spheres (one filter group, no group loop)
#endif
#ifndef FLUX_EXP2_ARGS
#define FLUX_EXP2_ARGS 1  // FAST glossy lobe: the 2^x polynomial's coefficients from the kernel arguments (scalar loads) instead of literals
#endif
#ifndef FLUX_SET_ROWS
#define FLUX_SET_ROWS 1  // FAST bounce / split kernel: a set's table rows from the context's DevSetRows record (one scalar load) instead of pointer + set * stride
#endif
#ifndef FLUX_SPLIT_UNIFORM_SUB
#define FLUX_SPLIT_UNIFORM_SUB 1  // render_split_kernel: the wave's index in its block read into a scalar register (cursor / queue count in SGPRs)
#endif
#ifndef FLUX_SCALAR_LIVE
#define FLUX_SCALAR_LIVE 0  // render_split_kernel: the lane mask of `live` kept in scalar registers; bit 0: the pop, 1: "a lane is free", 2: "no lane is live" use it
#endif
#ifndef FLUX_SPLIT_OPAQUE_UNIFORMS
#define FLUX_SPLIT_OPAQUE_UNIFORMS 3  // render_split_kernel: bit 0 = the sample set, bit 1 = the sphere masks opaque per pass (no loop-invariant scalar pairs derived from them)
#endif
#ifndef FLUX_SPLIT_PIXEL_CONSTS
#define FLUX_SPLIT_PIXEL_CONSTS 2  // render_split_kernel: primary_ray's per-pixel constants: 1 = kept in scalar registers, 2 = scalar loads from tables per pass
#endif
#ifndef FLUX_SPLIT_RELOAD_PARAMS
#define FLUX_SPLIT_RELOAD_PARAMS 1 // render_split_kernel: kernel arguments re-read (scalar loads) in every pass instead of kept alive across the loop (38 SGPRs spilled to VGPR lanes)
#endif
#ifndef FLUX_BVH4_MAT_LIST
#define FLUX_BVH4_MAT_LIST 0      // render_bvh4_kernel: 1 = a path's throughput as the list of its bounces' materials (1 VGPR, 8 gathers when the path
                                  // ends: the vector-memory pipeline is what this kernel is short of), 0 = the running product (6 VGPRs).
                                  // 1 M triangles @4096 spp, one box: list + per-lane sums 572.1, list + wave totals 583.5, product + wave totals 560.5 ms
#endif
#ifndef FLUX_BVH4_RELOAD_PARAMS
#define FLUX_BVH4_RELOAD_PARAMS 1 // render_bvh4_kernel: kernel arguments re-read (scalar loads) in every pass instead of ~64 of them spilled to VGPR lanes
#endif
#ifndef FLUX_BVH4_EARLY_REFILL
#define FLUX_BVH4_EARLY_REFILL 1  // render_bvh4_kernel: the node loop is left for the shading step as soon as FLUX_BVH_REFILL_AT walks have ended
#endif
#ifndef FLUX_BVH4_EARLY_AT
#define FLUX_BVH4_EARLY_AT 48     //   ... that many
#endif
#ifndef FLUX_BVH4_ENTRY
#define FLUX_BVH4_ENTRY 1         // render_bvh4_kernel: a pixel's camera rays enter the tree where its ray bundle first reaches two children
#endif
#ifndef FLUX_BVH_REFILL_AT
#define FLUX_BVH_REFILL_AT 40     // lanes that must be waiting for shading before the wave leaves traversal (swept 16..64 with the leaf vote)
#endif
// the early exit leaves the node loop when EARLY_AT walks have ended and expects the shading step to follow: with
// EARLY_AT < REFILL_AT the wave would re-enter the loop on the same counts and never advance
static_assert(!FLUX_BVH4_EARLY_REFILL || FLUX_BVH4_EARLY_AT >= FLUX_BVH_REFILL_AT,
              "FLUX_BVH4_EARLY_AT must not be below FLUX_BVH_REFILL_AT (render_bvh4_kernel would livelock)");
#ifndef FLUX_WPE_SPLIT
#define FLUX_WPE_SPLIT 5          // waves/SIMD of the split kernel: 96 VGPRs, nothing spilled since round 4 (4 until then: 128 VGPRs); demo2 @16384 spp 250.0 -> 225.4 ms
#endif
#ifndef FLUX_BVH_LEAF_VOTE
#define FLUX_BVH_LEAF_VOTE 1      // leave the inner-node loop once the lanes holding a leaf outweigh the descending ones
#endif
#ifndef FLUX_BVH_LEAF_NUM
#define FLUX_BVH_LEAF_NUM 2       // ... i.e. when n_leaf * NUM > n_inner * DEN (re-swept after the node step got cheaper: 1:1 178.4,
#define FLUX_BVH_LEAF_DEN 3       //     2:3 175.9, 1:2 176.4, 1:3 179.7, 3:2 178.8 ms at 1024 spp)
#endif
#ifndef FLUX_STRICT_BOX_HWMINMAX
#define FLUX_STRICT_BOX_HWMINMAX 1 // STRICT BoundingBox::hit: the reference's min / max forms through v_min_f64 / v_max_f64 + one unordered compare
#endif                             //   of the z slab (the same verdict bit for bit, render_body.inc scene_hit)
#ifndef FLUX_BVH4_LDS_SCENE
#define FLUX_BVH4_LDS_SCENE 1      // render_bvh4_kernel: the analytic set's hit records, the materials and the scan spheres in the block's LDS while they are small
                                   //   (stack + records <= 7 680 B, the 6 granules of 5 waves per SIMD: its own instantiation): the shading step's dependent gathers lose an L2 round trip; 1 M triangles 542.0 -> 530.9 ms
#endif
#ifndef FLUX_TRI_FDIV
#define FLUX_TRI_FDIV 0            // FAST triangle test: 1 / det by fastmath::fdiv (<= 2 ulp) instead of the IEEE division: measured SLOWER (548.3 against 542.3 ms), off
#endif
#ifndef FLUX_TRAV_RCP32
#define FLUX_TRAV_RCP32 1          // BVH kernels: the slab test's 1 / d from v_rcp_f32 instead of three IEEE f64 divisions per ray segment
#endif
#ifndef FLUX_SPLIT_EARLY_SAMPLES
#define FLUX_SPLIT_EARLY_SAMPLES 0 // render_split_kernel: phase A's pixel / lens samples requested before the queue pop (experiment, round 5)
#endif
#ifndef FLUX_SPLIT_LDS_SCENE
#define FLUX_SPLIT_LDS_SCENE 2     // render_split_kernel: hit records + scan spheres copied into the block's LDS: the per-lane gathers in the middle of a
                                   //   pass become LDS reads (round 5: the chip runs the kernel at 2.37 GHz and its VALU idles ~14 % of the cycles -- all
                                   //   resident waves waiting on memory at once); demo2 @16384 spp 227.3 -> 222.6 ms.  2 = the records FIRST in the
                                   //   dynamic LDS (an address the compiler knows: no scalar register holds it), the queues behind them: 215.4 -> 214.9 ms
#endif
#ifndef FLUX_STRICT_FILTER
#define FLUX_STRICT_FILTER 1       // STRICT Scene::hit: BoundingBox::hit + Sphere::hit only for the spheres FAST's conservative f32 filter passes
#endif                             //   (a rejected sphere's quadratic says miss whatever its box says): the same frames bit for bit
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
// FAST glossy-lobe factors of every pixel sample (flux_device.h FLUX_GLOSS_TABLE), compiled with the FAST arithmetic
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
