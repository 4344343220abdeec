/*
 * flux_abi.h -- C ABI of the MI355X (gfx950) render path for jtdaugherty/flux.
 *
 * This is the drop-in boundary for fluxcore's per-pixel render loop.  In the
 * reference the path sits behind exactly three calls made by the LocalWorker
 * job loop (fluxcore/src/workers.rs:46-60):
 *
 *     Scene::from_data(job.scene_data, job.config)        workers.rs:46
 *     Camera::new(..., num_sets = image_width, ...)       workers.rs:47-54
 *     camera.render(&scene, unit) -> WorkUnitResult       workers.rs:60
 *
 * A GPU worker (a third sibling of LocalWorker / NetworkWorker behind
 * `trait Worker`, fluxcore/src/manager.rs:232-236) binds the entry points
 * below instead; INTEGRATION.md shows the Rust `extern "C"` block.
 *
 * Plain pointers and sizes only; no C++/torch types.  All floating point is
 * IEEE f64 as in the reference.  Functions return 0 on success or a negative
 * FLUX_E_* code; flux_last_error() gives the message for the calling thread
 * (the reference panics instead -- workers.rs:78, job.rs:67-70 -- a C ABI
 * must not unwind).
 */
#ifndef FLUX_ABI_H
#define FLUX_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: flux_ctx_bvh_info writes FLUX_BVH_INFO_WORDS = 16 words (version 1 documented 8) and takes the caller's capacity;
 *    flux_ctx_launch_plan added.
 * 3: the multi-GPU frame (flux_multi_*, flux_render_frame_multi) and flux_ctx_create_timing added. */
#define FLUX_ABI_VERSION 3

/* error codes */
#define FLUX_OK 0
#define FLUX_E_INVALID (-1)  /* bad argument (null, out-of-range row, root<1, depth<1 ...) */
#define FLUX_E_DEVICE (-2)   /* HIP runtime error / no usable gfx950 device */
#define FLUX_E_NOMEM (-3)    /* device or host allocation failed */
#define FLUX_E_IO (-4)       /* file output failed */

/* ShapeData variants: fluxcore/src/scene.rs:71-74 */
#define FLUX_SHAPE_SPHERE 0
#define FLUX_SHAPE_PLANE 1

/* MaterialData variants: fluxcore/src/shapes.rs:42-47 */
#define FLUX_MAT_MATTE 0       /* MatteData            shapes.rs:52-56 */
#define FLUX_MAT_EMISSIVE 1    /* EmissiveData         shapes.rs:61-64 */
#define FLUX_MAT_REFLECTIVE 2  /* ReflectiveData       shapes.rs:69-72 */
#define FLUX_MAT_GLOSSY 3      /* GlossyReflectiveData shapes.rs:77-81 */

/* MaterialData (shapes.rs:42-81) flattened:
 *   Matte     : color = diffuse_color, ambient = ambient_color (parsed but
 *               unused by path_shade, materials.rs:18-34), k = diffuse_coefficient
 *   Emissive  : color, k = power
 *   Reflective: color = reflect_color, k = reflect_amount
 *   Glossy    : color = reflect_color, k = reflect_amount, exponent = reflect_exponent */
typedef struct flux_material {
    int32_t kind;
    int32_t reserved;
    double color[3];
    double ambient[3];
    double k;
    double exponent;
} flux_material;

/* ShapeData (scene.rs:71-74) = SphereData (shapes.rs:18-23) | PlaneData
 * (shapes.rs:33-37).  sphere: p = center, radius, invert.  plane: p = point,
 * n = normal (used as given: never normalised or flipped, shapes.rs:135-152). */
typedef struct flux_shape {
    int32_t kind;
    int32_t invert;
    double p[3];
    double n[3];
    double radius;
    flux_material material;
} flux_shape;

/* EXTENSION (absent in the reference: scene.rs:71-74 has Sphere | Plane only, TODO.md lists a Quad):
 * an indexed triangle mesh.  Semantics chosen to follow the reference's conventions for Plane
 * (shapes.rs:135-152): two-sided, hit iff t > T_MIN, geometric normal normalize((v1-v0) x (v2-v0))
 * stored per triangle and never flipped towards the ray.  Intersection is Moeller-Trumbore in f64
 * (DESIGN.md "Triangles and the BVH").  Hit order for the nearest-hit tie rule: all `shapes` first
 * (YAML order), then the triangles of meshes[0], meshes[1], ... in index order. */
typedef struct flux_mesh {
    uint64_t num_vertices;
    const double *vertices;  /* [num_vertices][3] */
    uint64_t num_triangles;
    const uint32_t *indices; /* [num_triangles][3] vertex indices */
    flux_material material;
} flux_mesh;

/* SceneData (scene.rs:40-49) with CameraSettings (:14-18), CameraData
 * (:53-58) and OutputSettings (:62-66) inlined.  `shapes` keeps YAML order:
 * Scene::hit resolves distance ties to the lowest index (scene.rs:156-160,
 * common.rs:17-23). */
typedef struct flux_scene_desc {
    const char *scene_name;
    uint64_t image_width;
    uint64_t image_height;
    double pixel_size;
    double background[3];
    double eye[3];
    double look_at[3];
    double up[3];
    double zoom_factor;
    double view_plane_distance;
    double focal_distance;
    double lens_radius;
    uint64_t num_shapes;
    const flux_shape *shapes;
    uint64_t num_meshes;      /* extension; 0 for reference scenes */
    const flux_mesh *meshes;
} flux_scene_desc;

/* JobConfiguration: fluxcore/src/job.rs:49-53 */
typedef struct flux_job_cfg {
    uint64_t sample_root;
    uint64_t max_trace_depth;
    uint64_t rows_per_work_unit;
} flux_job_cfg;

/* WorkUnit: fluxcore/src/job.rs:40-44 (row_end inclusive; job_id is
 * scheduler bookkeeping and stays on the caller's side) */
typedef struct flux_work_unit {
    uint64_t row_start;
    uint64_t row_end;
} flux_work_unit;

typedef struct flux_ctx flux_ctx;

uint32_t flux_abi_version(void);
const char *flux_last_error(void);

/* Number of usable devices (0 when no GPU is visible; never fails). */
int flux_device_count(void);

/* Identity of the library's sources, fixed at build time (flux_amd/build.py): "lib:<16 hex> kernels:<16 hex>" -- hashes of all
 * sources + flags, and of the device-code sources alone (render.hip, render_body.inc, tables.hip and the headers they include).
 * A profile summary under profiles/ records the id of the binary it was taken from; bench.py compares. */
const char *flux_build_id(void);

/* Brings the HIP runtime up on `device` -- device context, the first copy, the first kernel launch of this library's code object,
 * a stream -- so that the first flux_ctx_create on it costs what every later one does (7 ms for demo2 at 16384 spp instead of 30-130).
 * The reference's counterpart is LocalWorker::new building its rayon pool when the worker is made (workers.rs:27-38), before any
 * job's timer runs (manager.rs:145).  Optional; idempotent; safe from any thread. */
int flux_device_warmup(int device);

/* Replaces Scene::from_data (scene.rs:128-154) + Camera::new (trace.rs:26-42)
 * incl. MasterSampleSets::new (sampling.rs:13-33): copies the scene, uploads
 * it to HBM and generates the S = image_width sample sets of pixel / lens-disc
 * / hemisphere tables ON THE DEVICE.  `seed` is an addition: the reference
 * seeds from OS entropy (samplers/src/lib.rs:27-33); see DESIGN.md "RNG
 * contract".  The caller keeps ownership of `scene`. */
int flux_ctx_create(const flux_scene_desc *scene, const flux_job_cfg *cfg, uint64_t seed,
                    int device, flux_ctx **out);

/* flux_ctx_create for ONE RANK of a set-sharded render (flux_render_sets_device with the same first_set / set_stride):
 * the sample tables are generated and held only for the sets first_set + m*set_stride this rank owns -- 1/set_stride
 * of MasterSampleSets::new's work and memory (sampling.rs:13-33).  A set's contents depend on (seed, set index) only,
 * so they equal the full context's.  flux_render_rows* and flux_debug_shade need every set and are rejected on such a
 * context; flux_ctx_copy_table returns the held sets in slot order.  first_set < set_stride. */
int flux_ctx_create_sets(const flux_scene_desc *scene, const flux_job_cfg *cfg, uint64_t seed, int device,
                         uint64_t first_set, uint64_t set_stride, flux_ctx **out);

/* Drops Scene + Camera (workers.rs:73-74). NULL is a no-op. */
void flux_ctx_destroy(flux_ctx *ctx);

/* Replaces Camera::render (trace.rs:53-97) for one WorkUnit.  Writes
 * (row_end-row_start+1) * image_width * 3 doubles, row-major RGB, already
 * averaged over sample_root^2 samples and max_to_one-clamped -- exactly what
 * WorkUnitResult.rows holds (manager.rs:24-28).  Synchronous: returns after
 * the device->host copy.  out_rgb is caller-owned host memory. */
int flux_render_rows(flux_ctx *ctx, uint64_t row_start, uint64_t row_end, double *out_rgb);

/* Same render, device-resident: rows first_row, first_row+row_stride, ...
 * (num_rows of them) are written to d_out_rgb (device pointer, num_rows *
 * image_width * 3 doubles) asynchronously on `hip_stream` (a hipStream_t, or
 * NULL for the default stream).  This is what the multi-GPU path uses so that
 * the framebuffer gather (RCCL) never leaves HBM.  row_stride >= 1. */
int flux_render_rows_device(flux_ctx *ctx, uint64_t first_row, uint64_t row_stride,
                            uint64_t num_rows, void *d_out_rgb, void *hip_stream);

/* The same render sharded along the OTHER axis of the work: sample sets instead of rows.  Every pixel of a row
 * uses a different sample set (trace.rs:64-69: `idx` is a permutation of 0..num_sets, num_sets = image_width,
 * workers.rs:50), so "the pixels whose set is first_set + m*set_stride, m < num_sets" is exactly one pixel per
 * row per set -- a 1/G share of the image for set_stride = G, with perfect load balance, and the share whose
 * sample tables stay cache-resident (DESIGN.md "set-grouped order").  Writes, for every image row r and m <
 * num_sets, 3 doubles at d_out_rgb[(r*num_sets + m)*3]; the pixel's column is the c with
 * flux_ctx_copy_row_perm(r)[c] == first_set + m*set_stride.  Asynchronous on `hip_stream` like
 * flux_render_rows_device.  Needs sample_root^2 >= 64 (refill kernel). */
int flux_render_sets_device(flux_ctx *ctx, uint64_t first_set, uint64_t set_stride, uint64_t num_sets,
                            void *d_out_rgb, void *hip_stream);

/* Render-kernel variants (all produce the same image within rounding):
 *   0 = default (FLUX_KERNEL_SPLIT where it applies, else FLUX_KERNEL_REFILL; STATIC below 64 spp)
 *   1 = FLUX_KERNEL_STATIC: one lane per sample, lanes idle once their path ends
 *   2 = FLUX_KERNEL_REFILL: persistent lanes refilled with the pixel's next
 *       sample by ballot/prefix compaction */
#define FLUX_KERNEL_DEFAULT 0
#define FLUX_KERNEL_STATIC 1
#define FLUX_KERNEL_REFILL 2
/*   3 = FLUX_KERNEL_SPLIT: FLUX_MATH_FAST, scenes without meshes and with <= 64 spheres, >= 64 spp: primary
 *       segments (coherent: all lanes of a wave sample one pixel) and secondary segments run in separate passes
 *       of the wave, joined by an LDS queue; elsewhere it behaves as FLUX_KERNEL_REFILL */
#define FLUX_KERNEL_SPLIT 3
int flux_ctx_set_kernel(flux_ctx *ctx, int variant);

/* Arithmetic of the render kernels -- both FP64 end to end, both checked against the oracle at the
 * north-star tolerance (1e-4 per channel):
 *   FLUX_MATH_FAST (default): the reference's estimator evaluated for the machine -- FMA contraction,
 *       division/sqrt/pow/sincos from flux_math.h (<= ~2 ulp), no BoundingBox::hit pre-test (the sphere quadratic
 *       implies its answer, except that a ray with direction.z == 0 whose origin lies on a z face of a sphere's box
 *       misses that sphere -- the box's 0 * inf = NaN, shapes.rs:121-130 -- which is reproduced by an explicit rule),
 *       path throughput multiplied front to back, bounce weights in closed form (glossy: cs ks, the Phong lobe cancels).
 *       That is the reference's value to rounding wherever surface normals are unit vectors.  A plane stored with a
 *       NON-unit normal makes reflected directions non-unit, lobes under- / overflow and the reference's recursion meet
 *       zeros and infinities in its own order (NaN pixels): a scene that has one is rendered with the STRICT arithmetic
 *       whatever this setting says, so the reference's NaN pixels appear exactly (flux_ctx_launch_plan reports the
 *       arithmetic in use).  WHAT THAT COSTS: STRICT renders demo2 at 16384 spp in 0.80 s where FAST
 *       takes 0.20 s, and a mesh scene also loses both traversal kernels -- so `normal: [0, 2, 0]`, or a normal typed to four
 *       digits ([0.7071, 0, 0.7071]: |n|^2 = 0.99997), is a 4 x slower scene than the same plane normalised to double
 *       precision.  The test is |n.n - 1| > 4 eps and is not looser on purpose: FAST takes every direction as a unit vector,
 *       and a normal that is off by 3e-5 bends every reflection off it by as much, which is above the parity tolerance.
 *       One exception: where the STRICT arithmetic cannot run the job at all (its LDS recursion stack holds 31 levels of
 *       max_trace_depth), such a scene stays with FAST and the reference's long-form glossy weights, as before round 4;
 *   FLUX_MATH_STRICT: the reference's operation order, no contraction, IEEE division/sqrt, OCML
 *       pow/sincos, BoundingBox::hit before every sphere, (f,s) stack folded deepest bounce first. */
#define FLUX_MATH_FAST 0
#define FLUX_MATH_STRICT 1
int flux_ctx_set_math(flux_ctx *ctx, int mode);

/* Triangle traversal (extension): 0 = BVH with a per-lane LDS stack (default), 1 = brute force over
 * all triangles in index order (the definition the BVH must reproduce exactly; parity tests). */
#define FLUX_TRAVERSE_BVH 0
#define FLUX_TRAVERSE_BRUTE 1
/* the BVH, but the FAST state-machine kernel walks the BINARY tree (32-B nodes, one record per triangle: round 2's kernel,
 * otherwise only the fallback for meshes whose 4-wide tree needs too deep a stack) -- a test hook that keeps it exercised */
#define FLUX_TRAVERSE_BVH_BINARY 2
int flux_ctx_set_traversal(flux_ctx *ctx, int mode);

/* Device time (ms, HIP events on the launch stream) of the most recent render
 * kernel launched through this context; synchronises that launch. <0 on error. */
double flux_ctx_last_kernel_ms(flux_ctx *ctx);

/* Path statistics of the launches since the last reset (requires
 * flux_ctx_enable_stats(ctx,1) before rendering; off by default):
 * [0] samples, [1] ray segments, [2] Matte bounces, [3] glossy bounces,
 * [4] perfect-specular bounces, [5] emissive terminations, [6] misses,
 * [7] depth-exhausted paths, [8] BVH nodes visited, [9] triangles tested,
 * [10..15] reserved (0). */
#define FLUX_NUM_STATS 16
int flux_ctx_enable_stats(flux_ctx *ctx, int on);
int flux_ctx_stats(flux_ctx *ctx, uint64_t out[FLUX_NUM_STATS], int reset);

/* BVH introspection (extension): out[0] nodes of the binary tree, [1] triangles, [2] its max depth, [3] max leaf size,
 * [4] its node bytes, [5] triangle-record bytes, [6] build microseconds; the 4-wide tree the FAST traversal kernel walks:
 * [7] nodes, [8] leaf records, [9] of them holding two triangles (the halves of a quad), [10] most stack entries at once
 * (one 32-bit entry per node with two children or more on a path: the 4-wide tree's depth), [11] node bytes, [12] leaf-record
 * bytes, [13] 1 if a full-frame render with the context's current settings walks the
 * 4-wide tree (flux_ctx_launch_plan's kernel == FLUX_PLAN_BVH4); [14..15] reserved (0).
 * Writes min(out_words, FLUX_BVH_INFO_WORDS) words: a caller states the capacity of its buffer. */
#define FLUX_BVH_INFO_WORDS 16
int flux_ctx_bvh_info(flux_ctx *ctx, uint64_t *out, uint64_t out_words);

/* What a render call WOULD launch with the context's current kernel variant, arithmetic and traversal -- the decision
 * itself (the library's one launch planner), not a restatement of it: for flux_render_rows* of `num_rows` rows when
 * num_sets == 0, for flux_render_sets_device of `num_sets` sets (all rows) otherwise.
 * out[0] kernel (FLUX_PLAN_*), [1] threads per block, [2] blocks, [3] dynamic LDS bytes per block,
 * [4] waves that share one pixel's samples (K: 1, 2 or 4 -- from the sample count, and in the STRICT arithmetic the LDS its
 * recursion stack leaves), [5] the arithmetic the launch runs with (FLUX_MATH_*: STRICT also under FLUX_MATH_FAST when the
 * scene has a plane with a non-unit normal, see flux_ctx_set_math), [6] FLUX_ROUTE_*: NONE; TO_STRICT = FLUX_MATH_FAST was
 * requested, the scene has a plane with a non-unit normal, the launch runs STRICT (several times slower); KEPT_FAST = the same
 * scene, but max_trace_depth (and the mesh's BVH depth) leave no room for STRICT's LDS recursion stack, so the job stays FAST
 * with the reference's long-form glossy weights -- its pixels can differ from the reference's NaN pixels in the rarest orderings
 * of an overflow and a zero.  The decision depends on the job alone, not on flux_ctx_set_traversal.
 * [7] reserved (0). */
#define FLUX_ROUTE_NONE 0
#define FLUX_ROUTE_TO_STRICT 1
#define FLUX_ROUTE_KEPT_FAST 2
#define FLUX_PLAN_NONE (-1)    /* nothing to launch */
#define FLUX_PLAN_STATIC 0     /* render_static_kernel */
#define FLUX_PLAN_REFILL 1     /* render_refill_kernel */
#define FLUX_PLAN_SPLIT 2      /* render_split_kernel */
#define FLUX_PLAN_BVH_BINARY 3 /* render_bvh_kernel */
#define FLUX_PLAN_BVH4 4       /* render_bvh4_kernel */
#define FLUX_PLAN_WORDS 8
int flux_ctx_launch_plan(flux_ctx *ctx, uint64_t num_rows, uint64_t num_sets, int64_t out[FLUX_PLAN_WORDS]);

/* Introspection used by the parity tests (device -> host copies).
 * which: 0 = pixel_sets [S][N][2], 1 = disc_sets [S][N][2],
 *        2 = hemi_sets  [S][D][N][3] (converted from the device SoA layout). */
#define FLUX_TABLE_PIXEL 0
#define FLUX_TABLE_DISC 1
#define FLUX_TABLE_HEMI 2
int flux_ctx_copy_table(flux_ctx *ctx, int which, double *out, uint64_t out_doubles);
int flux_ctx_copy_row_perm(flux_ctx *ctx, uint64_t row, int32_t *out, uint64_t out_len);
int flux_ctx_camera_basis(flux_ctx *ctx, double uvw[9]); /* CameraBasis::new scene.rs:28-35 */
/* bytes of HBM held by the context's tables + scene */
uint64_t flux_ctx_device_bytes(flux_ctx *ctx);

/* Where the wall time of the flux_ctx_create* call that made this context went, in milliseconds.  The reference's timer
 * (manager.rs:145 -> 170) spans exactly this work -- Scene::from_data + Camera::new incl. MasterSampleSets::new run
 * inside it (workers.rs:46-54) -- so it is reported beside the render time (bench.py `reference_equivalent_s`).
 * out[FLUX_CREATE_MS_*]: TOTAL = the whole call; HOST = validation, scene records, BVH build (host arithmetic);
 * RUNTIME = device selection + the HIP runtime's lazy per-device initialisation (zero in a process that has used the
 * device before); ALLOC = device allocation; UPLOAD = host -> device copies of the scene; TABLES = the sample-table
 * kernels (MasterSampleSets::new, sampling.rs:13-33) incl. the wait for them; FREE = release of the generator's
 * scratch; OTHER = events and the rest.  The parts sum to TOTAL. */
#define FLUX_CREATE_MS_TOTAL 0
#define FLUX_CREATE_MS_HOST 1
#define FLUX_CREATE_MS_RUNTIME 2
#define FLUX_CREATE_MS_ALLOC 3
#define FLUX_CREATE_MS_UPLOAD 4
#define FLUX_CREATE_MS_TABLES 5
#define FLUX_CREATE_MS_FREE 6
#define FLUX_CREATE_MS_OTHER 7
#define FLUX_CREATE_TIMING_WORDS 8
int flux_ctx_create_timing(flux_ctx *ctx, double out_ms[FLUX_CREATE_TIMING_WORDS]);

/* The samplers crate's four generators, evaluated on the device (what sampler-debug plots,
 * sampler-debug/src/main.rs:48-57): kind 0 grid_regular (samplers/src/lib.rs:184-191), 1 grid_jittered
 * (lib.rs:35-44), 2 grid_multi_jittered (lib.rs:64-73; equals hemi-stream set 0 depth 0 before the
 * hemisphere map), 3 grid_correlated_multi_jittered (lib.rs:75-90; equals pixel_sets[0]).  Writes
 * sample_root^2 (x,y) pairs to out_xy and, if out_hemi != NULL, to_hemisphere(.., 0.0) of them
 * (lib.rs:129-142) as sample_root^2 (x,y,z) triples.  Host pointers. */
#define FLUX_SAMPLER_REGULAR 0
#define FLUX_SAMPLER_JITTERED 1
#define FLUX_SAMPLER_MULTI_JITTERED 2
#define FLUX_SAMPLER_CORRELATED_MULTI_JITTERED 3
int flux_sampler_grid(int device, int kind, uint64_t sample_root, uint64_t seed, double *out_xy, double *out_hemi);

/* Test hook: Scene::shade (scene.rs:162-172) on the device for `n` caller-supplied rays (rays = n x {origin[3],
 * direction[3]}, host memory), each traced from `depth` (1 = a primary ray) with sample (set_index, sample_index) of
 * the context's tables and the context's arithmetic (flux_ctx_set_math).  out_rgb: n x 3 (un-clamped radiance, what
 * Scene::shade returns); out_hit: first hit per ray as the shape's YAML index, num_shapes + triangle index, or -1
 * (Scene::hit, scene.rs:156-160); out_t: its distance.  out_hit / out_t may be NULL. */
int flux_debug_shade(flux_ctx *ctx, uint64_t n, const double *rays, uint64_t depth, uint64_t set_index,
                     uint64_t sample_index, double *out_rgb, int32_t *out_hit, double *out_t);

/* Test hook for csrc/flux_math.h: out[i] = fn(a[i], b[i]) evaluated ON THE DEVICE (host pointers in,
 * host pointer out; b may be NULL for unary functions).  fn: 0 frsqrt, 1 fsqrt, 2 fdiv, 3 flog2,
 * 4 fexp2, 5 fpow_pos, 6 sin(2 pi a), 7 cos(2 pi a), 8 raw v_rsq_f64, 9 raw v_rcp_f64. */
int flux_debug_fastmath(int device, int fn, const double *a, const double *b, double *out, uint64_t n);

/* ---- one frame on the GPUs of one node ------------------------------------------------------------------------------
 * Replaces, for GPUs in ONE process, what RenderManager does with a job: the fan-out of the job to every worker
 * (fluxcore/src/manager.rs:156-162: one clone of the job per worker handle) and ImageBuilder's gather of their rows into
 * the image (manager.rs:316-324).  Here the fan-out is one context per device -- each holding ONLY its share of the
 * sample tables (flux_ctx_create_sets(g, G): 1/G of MasterSampleSets::new's work and memory) -- and the gather is ONE
 * collective, ncclAllGather (RCCL, over xGMI between the GPUs of a node) of the shares, after which the frame is
 * reassembled ON THE DEVICE by the row permutation and copied to the caller once.  The frame is bit-identical for every
 * number of devices and equal to flux_render_rows' (a pixel's value depends on (seed, row, column) only).
 *
 * How the frame is split (`shard`):
 *   FLUX_SHARD_SETS: rank g renders, in every row, the pixels whose sample set s has s mod G == g
 *       (flux_render_sets_device): one pixel per row per owned set, i.e. a 1/G share with the cost of an average pixel;
 *       needs sample_root^2 >= 64.  Measured 98 % of ideal at G = 8.
 *   FLUX_SHARD_ROWS: rank g renders rows g, g + G, ... (flux_render_rows_device with row_stride G) -- the reference's
 *       WorkUnit rows, interleaved; every rank then holds ALL sample tables.
 *   FLUX_SHARD_AUTO: sets where they apply, rows below 64 spp.
 *
 * RCCL is bound at run time (dlopen of librccl.so.1; an RCCL the process has already mapped -- PyTorch's -- is used as
 * it is), so the library has no link-time dependency on it; flux_multi_create fails with FLUX_E_DEVICE when none is
 * found.  The communicators of a device list are created once per process (ncclCommInitAll) and kept for later
 * flux_multi_create calls on the same list; flux_multi_release_comms() destroys them.
 * Threading: a flux_multi is used by one host thread at a time. */
#define FLUX_SHARD_AUTO 0
#define FLUX_SHARD_SETS 1
#define FLUX_SHARD_ROWS 2
/* Test hook, OR-ed into `shard`: the ranks may share devices (the same ordinal listed several times) and the all-gather is
 * replaced by device-to-device copies -- no RCCL is loaded or called.  Every other step (per-rank share contexts, launches,
 * padding, reassembly) is the product path, so G = 2, 3, 8 can be checked on a one-GPU box. */
#define FLUX_SHARD_LOOPBACK 0x100

typedef struct flux_multi flux_multi;

/* devices[0..num_devices): HIP device ordinals, all different; devices[0] assembles the frame.  The contexts are
 * created concurrently (one host thread per device) while the communicators come up. */
int flux_multi_create(const flux_scene_desc *scene, const flux_job_cfg *cfg, uint64_t seed, const int *devices,
                      uint64_t num_devices, int shard, flux_multi **out);
void flux_multi_destroy(flux_multi *m); /* NULL is a no-op */

/* Camera::render for the WHOLE image (every WorkUnit of Job::work_units at once) on all devices: writes
 * image_height * image_width * 3 doubles, row-major RGB, averaged and max_to_one-clamped (what ImageBuilder holds after
 * the last RowsReady, manager.rs:316-324).  Synchronous.  out_rgb is caller-owned host memory. */
int flux_multi_render_frame(flux_multi *m, double *out_rgb);
/* The same frame left in HBM: *d_frame_rgb points at image_height * image_width * 3 doubles on devices[0], valid until
 * the next render call on `m` or its destruction.  Synchronous (returns once every device has finished the gather). */
int flux_multi_render_frame_device(flux_multi *m, const void **d_frame_rgb);

/* flux_ctx_set_kernel / flux_ctx_set_math for every rank's context */
int flux_multi_set_kernel(flux_multi *m, int variant);
int flux_multi_set_math(flux_multi *m, int mode);
/* rank's context, borrowed (owned by `m`): for flux_ctx_launch_plan, statistics, tables */
int flux_multi_ctx(flux_multi *m, uint64_t rank, flux_ctx **ctx);

/* out[0] devices G, [1] the split in use (FLUX_SHARD_SETS / FLUX_SHARD_ROWS), [2] RCCL version code (ncclGetVersion),
 * [3] doubles each rank contributes to the all-gather (H * ceil(S / G) * 3 or ceil(H / G) * W * 3), [4] bytes of HBM held
 * by all ranks' contexts together, [5] bytes of the gather + frame buffers on all devices, [6] 1 if the communicators
 * were taken from the process-wide cache (an earlier flux_multi_create on the same device list), [7] reserved (0). */
#define FLUX_MULTI_INFO_WORDS 8
int flux_multi_info(flux_multi *m, uint64_t out[FLUX_MULTI_INFO_WORDS]);

/* Milliseconds.  [0] flux_multi_create wall time, [1] of it: the slowest rank's flux_ctx_create_sets, [2] of it:
 * communicator creation (0 when cached; runs beside the context creation), then of the most recent frame: [3] wall time of
 * the render call, [4] the slowest rank's render kernel (HIP events on its stream), [5] all-gather (events on devices[0]'s
 * stream: from its own kernel's end, so it includes the wait for slower ranks), [6] reassembly kernel, [7] device -> host
 * copy of the frame (0 for flux_multi_render_frame_device). */
#define FLUX_MULTI_TIMING_WORDS 8
int flux_multi_timing(flux_multi *m, double out_ms[FLUX_MULTI_TIMING_WORDS]);

/* Destroys the process-wide communicator cache (ncclCommDestroy); call when no flux_multi is alive.  Returns the number of
 * device lists released.  Never fails. */
int flux_multi_release_comms(void);

/* One call: flux_multi_create + flux_multi_render_frame + flux_multi_destroy (a job's whole life, manager.rs:145-170).
 * devices may be NULL: the first num_devices devices (num_devices == 0: all visible ones). */
int flux_render_frame_multi(const flux_scene_desc *scene, const flux_job_cfg *cfg, uint64_t seed, const int *devices,
                            uint64_t num_devices, int shard, double *out_rgb);

/* Job::work_units (job.rs:65-88), including its `i < H-1` loop guard.  Writes
 * at most `cap` units and returns the number the reference would issue, or a
 * negative code (rows_per_work_unit == 0 panics in the reference). */
int64_t flux_work_units(uint64_t image_height, uint64_t rows_per_work_unit, flux_work_unit *out,
                        uint64_t cap);

/* Image::write (image.rs:43-61): ASCII P3, maxval 65535, `(c*65535.99) as
 * u16`, one pixel per line; rows whose rows_present[r]==0 (never received) are
 * written as zeros.  rows_present may be NULL (all present). */
int flux_write_ppm(const char *path, const double *rgb, uint64_t width, uint64_t height,
                   const uint8_t *rows_present);

#ifdef __cplusplus
}
#endif
#endif /* FLUX_ABI_H */
