/*
 * flux_oracle.h -- CPU restatement of fluxcore's per-pixel render loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under flux_amd/ may include, link or call
 * this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, as the checker / the timed CPU baseline -- never as the product path.
 *
 * WHAT PINS IT.  The reference (jtdaugherty/flux, Rust) has no tests and no golden
 * vectors, seeds its RNG from OS entropy (samplers/src/lib.rs:27-33: it never
 * reproduces an image itself) and cannot be built here (no Rust toolchain, crates
 * not vendored), so a bit-for-bit pin cannot exist.  The pin is the reference's one
 * published output, demo.png (README.md:1-3: demo2.yml), committed untouched at its
 * real 16-bit precision (tests/golden/demo2_ref_800x600_u16.npy, made by
 * tests/golden/make_demo2_ref16.py) and compared with a measured noise model
 * (tests/ref16.py): whole-image mean within 1e-5 per channel, 38 000 deterministic
 * pixels within one 16-bit quantum, per-pixel z-scores against a held-out seed
 * (tests/test_gpu_ref16.py through the HIP path, which equals this file to ~1e-13 on
 * identical inputs; tests/test_oracle_kat.py for this file itself at 256 spp).
 * Limits of that pin: it is statistical, and it covers demo2 only -- demo1,
 * PerfectSpecular, the tie rule and the NaN quirks rest on the hand-derived
 * known-answer tests of tests/test_oracle_kat.py.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic it follows.  All arithmetic is IEEE f64, compiled with
 * -ffp-contract=off (rustc never contracts a*b+c).
 */
#ifndef FLUX_ORACLE_H
#define FLUX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* shape kinds: fluxcore/src/scene.rs:71-74 */
#define FXO_SHAPE_SPHERE 0
#define FXO_SHAPE_PLANE 1
/* material kinds: fluxcore/src/shapes.rs:42-47 */
#define FXO_MAT_MATTE 0
#define FXO_MAT_EMISSIVE 1
#define FXO_MAT_REFLECTIVE 2
#define FXO_MAT_GLOSSY 3

/* Scene description as flat arrays (all f64), shapes in YAML order.
 * shape_params[8*i]:  sphere: cx,cy,cz,radius,invert(0/1),-,-,-
 *                     plane : px,py,pz,nx,ny,nz,-,-
 * mat_params[8*i]:    Matte     : dr,dg,db, ar,ag,ab, kd
 *                     Emissive  : r,g,b, power
 *                     Reflective: r,g,b, reflect_amount
 *                     Glossy    : r,g,b, reflect_amount, reflect_exponent
 * camera[13]: eye[3], look_at[3], up[3], zoom_factor, view_plane_distance,
 *             focal_distance, lens_radius
 */
typedef struct fxo_ctx fxo_ctx;

fxo_ctx *fxo_ctx_create(const double *camera, int image_width, int image_height,
                        double pixel_size, const double *background,
                        int num_shapes, const int32_t *shape_kinds,
                        const double *shape_params, const int32_t *mat_kinds,
                        const double *mat_params, int sample_root,
                        int max_trace_depth, uint64_t seed);
void fxo_ctx_destroy(fxo_ctx *c);
/* EXTENSION (triangles do not exist in the reference): append an indexed mesh; its triangles follow
 * all shapes (and earlier meshes) in hit order and are intersected by brute force. Returns 0/-1. */
int fxo_ctx_add_mesh(fxo_ctx *c, const double *vertices, size_t num_vertices, const uint32_t *indices,
                     size_t num_triangles, int mat_kind, const double *mat_params);

/* Camera::render (trace.rs:53-97) for rows [row_start,row_end] inclusive.
 * out: (row_end-row_start+1)*W*3 f64, averaged and max_to_one-clamped.
 * threads<=1: single thread; else a pthread pool pulling rows. Returns 0/-1. */
int fxo_render_rows(fxo_ctx *c, size_t row_start, size_t row_end, double *out,
                    int threads);
/* Same over an arbitrary list of rows (for bounded CPU-baseline samples). */
int fxo_render_row_list(fxo_ctx *c, const int32_t *rows, size_t nrows,
                        double *out, int threads);

/* statistics accumulated since the last reset (all rows rendered):
 * [0] samples (camera paths), [1] ray segments (Scene::hit calls),
 * [2] Matte bounces, [3] glossy bounces, [4] perfect-specular bounces,
 * [5] emissive terminations, [6] misses, [7] depth-exhausted paths */
void fxo_stats(const fxo_ctx *c, uint64_t out[8]);
void fxo_stats_reset(fxo_ctx *c);

/* table access for table-parity tests.  pixel/disc: [S][N][2]; hemi:
 * [S][D][N][3]; row_perm: permutation of 0..S for one image row. */
const double *fxo_pixel_sets(const fxo_ctx *c);
const double *fxo_disc_sets(const fxo_ctx *c);
const double *fxo_hemi_sets(const fxo_ctx *c);
void fxo_row_perm(const fxo_ctx *c, size_t row, int32_t *out);
void fxo_camera_basis(const fxo_ctx *c, double uvw[9]);

/* ---- unit-level entry points for the KATs ---- */
uint64_t fxo_rng_draw(uint64_t key, uint64_t counter);
uint64_t fxo_rng_key(uint64_t seed, uint64_t kind, uint64_t a, uint64_t b,
                     uint64_t sub);
double fxo_rng_unit(uint64_t key, uint64_t counter);
void fxo_shuffle(uint64_t key, int32_t *v, size_t n);
void fxo_grid_regular(int root, double *out /* n*n*2 */);
void fxo_grid_jittered(uint64_t key, int root, double *out);
void fxo_grid_multi_jittered(uint64_t seed, uint64_t kind, uint64_t a,
                             uint64_t b, int root, double *out);
void fxo_grid_correlated_multi_jittered(uint64_t seed, uint64_t kind,
                                        uint64_t a, uint64_t b, int root,
                                        double *out);
void fxo_to_unit_hemi(double x, double y, double e, double out[3]);
void fxo_to_poisson_disc(double x, double y, double out[2]);
void fxo_max_to_one(double rgb[3]);
int fxo_bbox_hit(const double c0[3], const double c1[3], const double o[3],
                 const double d[3]);
/* returns 1 on hit; t,normal[3],point[3] */
int fxo_sphere_hit(const double center[3], double radius, int invert,
                   const double o[3], const double d[3], double *t,
                   double normal[3], double point[3]);
int fxo_plane_hit(const double p[3], const double n[3], const double o[3],
                  const double d[3], double *t, double normal[3],
                  double point[3]);
/* Scene::hit (scene.rs:156-160): index of nearest shape or -1 */
int fxo_scene_hit(const fxo_ctx *c, const double o[3], const double d[3],
                  double *t, double normal[3], double point[3]);
/* Scene::shade (scene.rs:162-172) for one ray with explicit (set,index) */
void fxo_shade(fxo_ctx *c, const double o[3], const double d[3], int depth,
               size_t set_index, size_t sample_index, double rgb[3]);
/* primary ray of (row,col,sample) exactly as trace.rs:71-80 builds it */
void fxo_primary_ray(const fxo_ctx *c, size_t row, size_t col, size_t set_index,
                     size_t sample_index, double o[3], double d[3]);
/* BRDF::sample_f for one material (brdf.rs): returns wi[3], pdf, f[3] */
void fxo_sample_f(int mat_kind, const double *mat_params, const double n[3],
                  const double wo[3], const double hemi[3], const double sq[2],
                  double wi[3], double *pdf, double f[3]);
/* Job::work_units (job.rs:65-88): writes up to cap (start,end) pairs,
 * returns the count the reference would produce */
size_t fxo_work_units(size_t image_height, size_t rows_per_unit,
                      size_t *starts, size_t *ends, size_t cap);
/* Image::write quantisation (image.rs:50-53): (c*65535.99) as u16 */
uint16_t fxo_ppm_quantize(double c);
/* Image::write (image.rs:43-61) of a full H*W*3 image; rows_present[r]==0
 * means the row was never received and is zero-padded. Returns 0/-1. */
int fxo_write_ppm(const char *path, const double *rgb, size_t width,
                  size_t height, const uint8_t *rows_present);

#ifdef __cplusplus
}
#endif
#endif
