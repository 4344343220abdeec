// flux_device.h -- device-side data layout shared by the table generator, the
// render kernels and the C-ABI implementation.  See DESIGN.md "Data layout in HBM".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flux_bvh.h"

// Numeric tunables of the data layout / launch mapping (the boolean either/ors of rounds 1-5 are folded into the code: render.hip).
// The render path as shipped: hemi_sets as [S][D][N][4] (one aligned 32-B sector per sample); waves ordered by sample set, one set
// per XCD at a time (render_body.inc map_wave); FAST: the conservative f32 sphere filter, unit directions where the scene
// guarantees them (RenderParams::unit_dirs), the self-skip of convex spheres (self_skip), the environment shortcut (env_short),
// BoundingBox::hit's z-slab NaN miss reproduced by rule (box_z_nan_miss), the glossy lobe's sample-only factors tabulated
// (RenderParams::gloss); mesh scenes on the 4-wide tree where its stack fits.

// FAST scan: up to this many `invert` spheres are tested wave-uniformly instead of per lane (RenderParams::n_uni)
#ifndef FLUX_UNI_SPHERES
#define FLUX_UNI_SPHERES 2
#endif
// FAST mesh scenes: render_bvh4_kernel over the 4-wide tree (flux_bvh.h) as long as its per-lane stack needs at most this many
// entries (48 x 256 B = 12 KiB of LDS per wave: 3 waves/SIMD); deeper trees fall back to render_bvh_kernel over the binary one
#ifndef FLUX_BVH_WIDE_MAX_STACK
#define FLUX_BVH_WIDE_MAX_STACK 48
#endif
// refill kernel: most waves that share one pixel's samples (launch_render picks K <= this, a power of two)
#ifndef FLUX_MAX_WAVES_PER_PIXEL
#define FLUX_MAX_WAVES_PER_PIXEL 4
#endif
// ... each wave of a pixel's block gets at least this many samples (a multiple of 64): a wave's slice ends with a
// drain of a few passes at falling occupancy, so short slices cost throughput; long ones cost balance at the end of a
// launch.  K depends on the sample count only, never on how a frame is split.
#ifndef FLUX_MIN_SAMPLES_PER_WAVE
#define FLUX_MIN_SAMPLES_PER_WAVE 4096
#endif

namespace flux {

constexpr int kHemiDoubles = 4;  // doubles of hemi table per (set, depth, sample): x, y, z, pad -- one aligned 32-B sector
constexpr double kTMin = 0.0005;                       // constants.rs:4
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double kInvPi = 1.0 / kPi;                   // constants.rs:5

constexpr int kShapeSphere = 0;
constexpr int kShapePlane = 1;
constexpr int kMatMatte = 0;
constexpr int kMatEmissive = 1;
constexpr int kMatReflective = 2;
constexpr int kMatGlossy = 3;

// One shape, 128 B.  The shape loop index is wave-uniform, so these are
// fetched with scalar loads (s_load_dwordx8/x16) and live in SGPRs.
struct DevShape {
    double px, py, pz;     // sphere centre | plane point
    double rr;             // radius*radius (shapes.rs:179 recomputes it per ray); first 32 B = sphere quadratic
    double c0x, c0y, c0z;  // sphere AABB corner0 (Sphere::new, shapes.rs:154-169) | plane normal
    double radius;         // sphere radius
    double c1x, c1y, c1z;  // sphere AABB corner1
    double inv;            // invert_val: -1 if `invert` else +1 (shapes.rs:181)
    int32_t kind;
    int32_t pad0;
    double inv_rad;        // inv / radius (FAST path: the sphere normal's scale in one rounding)
    double pad1[2];
};
static_assert(sizeof(DevShape) == 128, "DevShape layout");

// One material per shape (same index), 64 B; gathered per lane after the
// nearest hit is known.
struct DevMaterial {
    // Matte: diffuse_color*kd*INV_PI (brdf.rs:30); Emissive: color*power
    // (materials.rs:45); Reflective/Glossy: reflect_color*reflect_amount
    // (brdf.rs:45,76) -- per-material constants the reference recomputes per hit.
    double fr, fg, fb;
    double exponent;  // Glossy reflect_exponent
    double inv_e1;    // 1/(exponent+1) (samplers/src/lib.rs:136)
    int32_t kind;
    int32_t exp_parity;  // Glossy exponent: 1 even integer, 2 odd integer, 0 not integral (powf of a negative base)
    double pad1[2];
};
static_assert(sizeof(DevMaterial) == 64, "DevMaterial layout");

// ---- FAST path scene: the same shapes, laid out for the scan / for the shade gather -------------
// Scan records are walked with a wave-uniform index (scalar loads, software-pipelined one record
// ahead); spheres and planes are separated so the sphere loop has no branch on the shape kind.
struct DevScanSphere {  // 32 B: one s_load_dwordx8
    double px, py, pz, rr;
};
static_assert(sizeof(DevScanSphere) == 32, "DevScanSphere layout");
// The same spheres for the CONSERVATIVE candidate filter, which runs in f32 (half the issue cost of f64; the exact f64
// test decides every hit afterwards).  In expanded form hb = o.u - p.u and c = o.o + p.(-2 o) + (p.p - r^2), with the
// rounding of every term (< 2.1e-6 (o.o + p.p + r^2) in all, render_body.inc) covered by a bias of 8e-6 of the same
// magnitudes folded into the constants: ppr = (p.p - r^2) - 8e-6 (p.p + r^2) - 1e-30 rounded DOWN here, the ray
// side scaling o.o by (1 - 8e-6) -- so c is only ever under- and dq = hb^2 - c over-estimated: a superset.
// Stored as PAIRS (sphere 2j in element 0, 2j+1 in element 1) so that one packed-f32 instruction (v_pk_fma_f32: two
// floats per lane at the issue cost of one f64 instruction) tests two spheres; a missing partner is all zeros.
typedef float flux_f2 __attribute__((ext_vector_type(2)));
struct DevScanSphere32 {  // 32 B per pair: four pairs (8 spheres) per 2 x s_load_dwordx16
    flux_f2 px, py, pz, ppr;  // (px, py, pz) = MINUS the centre: hb = o.u + (-p).u, c = o.o + (-p).(2 o) + ppr, the ray's side un-negated
};
static_assert(sizeof(DevScanSphere32) == 32, "DevScanSphere32 layout");
struct DevScanPlane {   // 64 B
    double px, py, pz;  // point
    double nx, ny, nz;  // normal as stored (never flipped / normalised, shapes.rs:135-152)
    int32_t id;         // index in YAML order (tie-break, scene.rs:156-160)
    int32_t pad0;
    double pad1;
};
static_assert(sizeof(DevScanPlane) == 64, "DevScanPlane layout");
// Everything Scene::shade needs about the winning shape, fetched in ONE per-lane batch (6 x 16 B) after
// the scan instead of a chain of dependent loads.  Indexed in scan order: spheres, then planes.
struct DevHitRec {      // 96 B
    double cx, cy, cz;  // sphere centre | plane normal
    double inv_rad;     // sphere: invert_val / radius
    // Emissive: the emitted radiance color * power (materials.rs:45).  Every other material: the FAST bounce weight f (n.wi)/pdf in
    // its closed form -- f / INV_PI for Matte (f = diffuse_color kd INV_PI, brdf.rs:30: the same two IEEE multiplications the kernels
    // performed per bounce until round 6, done once on the host), f itself for Reflective / Glossy (DevMaterial keeps the plain f)
    double fr, fg, fb;
    double inv_e1;      // Glossy: 1 / (exponent + 1); the exponent itself and its parity: DevMaterial (mats[orig_id]), long-form lobes only
    int32_t shape_kind, mat_kind;
    int32_t orig_id;      // index in YAML order (the tie rule's key; also the shape's material index)
    int32_t unit_normal;  // 1: the hit normal has length 1 to rounding (every sphere; a plane whose stored normal does)
    // x and z of the helper vector a = (ax, 1, az) the lobe's frame is built around: (0.0034, 1, 0.0071) for Matte (brdf.rs:22),
    // (0.00424, 1, 0.00764) otherwise (brdf.rs:58) -- per-lane constants that cost a dozen register moves per bounce when selected
    // by material kind in the kernel
    double ax, az;
};
static_assert(sizeof(DevHitRec) == 96, "DevHitRec layout");

// The sample sets a context holds tables for: local slot m = global set first + m * stride, m < count.
struct SetRange {
    uint32_t first, stride, count;
};

// Where one held sample set's rows of the four sample tables start (abi.hip fills one record per table slot).
struct DevSetRows {
    const double2 *pix, *disc;
    const double *hemi, *gloss;
};
static_assert(sizeof(DevSetRows) == 32, "DevSetRows layout");

// Kernel argument block (by value -> kernarg segment -> scalar loads).
struct RenderParams {
    // camera (trace.rs:44-60, scene.rs:28-35)
    double ex, ey, ez;
    double Ux, Uy, Uz, Vx, Vy, Vz, Wx, Wy, Wz;
    double aps;          // adjusted_pixel_size = pixel_size / zoom_factor
    double half_w, half_h;
    double factor;       // focal_distance / view_plane_distance
    double focal, lens_radius;
    double bgr, bgg, bgb;
    double pixel_denom;  // 1/(n*n)
    int32_t img_w, img_h;
    int32_t n_shapes, max_depth;
    uint32_t nsamp;      // N = n*n
    uint32_t num_sets;   // S = image_width
    // tables in HBM
    const DevShape *shapes;
    const DevMaterial *mats;
    const double2 *pix;   // [S][N] (x,y)                 pixel_sets
    const double2 *disc;  // [S][N] (x,y)                 disc_sets
    const double *hemi;   // hemi_sets: [S][D][N][4] (x,y,z,pad)
    const double *gloss;  // [S][N][4] (cos 2 pi x, sin 2 pi x, log2(1 - y), pad) of pixel_sets (FAST glossy lobe)
    const int32_t *rowperm;  // [H][S] sample-set index per (row, col)
    const int32_t *invperm;  // [H][S] its inverse: the column that uses set s in a row
    // work: rows first_row + k*row_stride, k < num_rows
    double *out;          // [num_rows][W][3]
    int32_t first_row, row_stride, num_rows;
    int32_t n_mats;       // records of `mats`; behind them n_mats bounce-weight records of 32 B (abi.hip; render_bvh4_kernel)
    unsigned long long *stats;  // FLUX_NUM_STATS counters or nullptr
    // extension: triangle meshes (0 / nullptr for reference scenes)
    const DevTri *tris;    // leaf order
    const DevNode *nodes;  // node 0 = root
    const DevNodeQ *nodesq;  // the same nodes, 32 B each, boxes on a 16-bit grid (FAST traversal state machine)
    float bvh_qmin[3], bvh_qstep[3];  // that grid: coordinate = qmin + q * qstep
    const DevNode4Q *nodes4;   // the 4-wide tree of the FAST traversal kernel (flux_bvh.h), or nullptr
    const DevLeafRec *leaves;  // its leaf records
    int32_t bvh4_stack;        // most entries its per-lane stack can hold at once
    int32_t mat_bits;          // bits that hold a material index (2^mat_bits >= n_mats)
    int32_t n_tris;
    int32_t bvh_stack;     // per-lane traversal stack entries (= BVH max depth); 0 = brute force
    // FAST path scene (same shapes as `shapes`/`mats`)
    const DevScanSphere *fsph;
    const DevScanPlane *fpln;
    const DevHitRec *frec;  // [n_sph + n_pln]
    const DevScanSphere32 *fsph32;  // [n_sph] f32 candidate-filter records, or nullptr (a coordinate beyond f32's safe range)
    const DevShape *sshapes;        // [n_sph] the spheres' STRICT records in scan order (pad0 = YAML index): STRICT's candidates
    int32_t n_sph, n_pln;
    double bvh_mag;  // largest |coordinate| of any mesh vertex (padding scale of the f32 slab test)
    // work, second axis: sample sets set_first + m*set_stride, m < set_count (default: all S sets).  When
    // out_by_set != 0 the pixel of (local row k, local set m) is written to out[(k*set_count + m)*3]
    // instead of out[(k*W + col)*3] (flux_render_sets_device).
    int32_t set_first, set_stride, set_count, out_by_set;
    // table slot of set set_first + m*set_stride: slot_first + m*slot_stride (equal to the set index itself unless the
    // context holds only a subset of the sets, flux_ctx_create_sets).  The kernels address pix/disc/hemi/gloss by slot.
    int32_t slot_first, slot_stride;
    // FAST: 1 = the scene has a plane whose stored normal is not a unit vector, so a reflected direction may not be one
    // and the Phong lobe may under/overflow: glossy bounces then use the reference's long-form weight (render_body.inc)
    int32_t glossy_long;
    // FAST: 1 = every sphere has |centre| and radius below 1e3 and the scene is not glossy_long: a ray leaving a convex
    // sphere outwards is then never tested against that sphere (render_body.inc scan_shapes_fast)
    int32_t self_skip;
    // FAST with the f32 filter: hit-record indices of up to two `invert` spheres that scan_shapes_fast tests for all
    // lanes together (their filter records never pass)
    int32_t n_uni, uni_idx[2];
    // FAST: 1 = no plane is stored with a non-unit normal (= !glossy_long) and the rays are the render loop's own: every
    // ray direction is then a unit vector to rounding and scan_shapes_fast skips its normalisation
    int32_t unit_dirs;
    // FAST: 1 = the scene's ONE `invert` sphere is Emissive (an environment): a secondary ray that starts well inside it
    // always reaches it, what it emits does not depend on where, so the split kernel decides "nearer than the best hit so
    // far?" on squared quantities and takes no square root (render_body.inc scan_shapes_fast<true>); env_radius = its radius
    int32_t env_short, pad_env;
    double env_radius;
    // kTMin (constants.rs:4) and -(4 kTMin) env_radius (the environment shortcut's "origin well inside" bound), as kernel arguments: held in a
    // scalar register pair through a pass, where the literals were two scalar moves at every use (round 6)
    double t_min, env_deep;
    // the environment sphere's scan record (fsph[uni_idx[0]] when n_uni == 1) as kernel arguments: one scalar load where the record's
    // address was a load of the index, four scalar instructions and a dependent load, every pass
    double env_px, env_py, env_pz, env_rr;
    double env_eps;  // 1e-9: the relative width of the shortcut's "too close to call" band
    // the f32 filter's walk over a scene of at most 32 spheres (sphere_filter32_laid_out), laid out on the host: the half group of one
    // or two pairs past the full groups (or nullptr), one past the last full group of four pairs, the number of full groups --
    // per pass the kernel formed all three from n_sph (two dozen scalar instructions)
    const DevScanSphere32 *f32_half, *f32_top;
    int32_t f32_groups;
    uint32_t f32_valid;  // bits 0 .. n_sph - 1
    // split kernel: the primary ray's per-frame and per-pixel constants, read with scalar loads in the ray-generation step instead of
    // living in scalar registers across the pass loop: focal * (Wx, Wy, Wz); pxc[x] = x - half_w, pxc[img_w + row] = (img_h - row) - half_h
    double fwx, fwy, fwz;
    const double *pxc;
    const DevSetRows *set_rows;  // [slots held]: the rows of pix / disc / hemi / gloss of each held set
    // flux_math_coeffs.h kExp2Poly, for the glossy lobe's 2^x: read with two scalar loads where the
    // literals cost 24 s_mov_b32 per evaluation
    double exp2c[12];
};

}  // namespace flux
