/*
 * flux_oracle.c -- CPU restatement of fluxcore's per-pixel render loop.
 *
 * TEST INFRASTRUCTURE ONLY (see flux_oracle.h).  PIN: the reference's only
 * published output, demo.png, at its real 16-bit precision (a statistical
 * pin by necessity -- the reference seeds its RNG from OS entropy and never
 * reproduces an image itself), plus hand-derived KATs; what the pin covers
 * and what it cannot is spelled out in flux_oracle.h and DESIGN.md section 2.
 *
 * Structure deliberately follows the reference's own decomposition (recursive
 * shade, explicit base-grid / shuffle / transpose sampler pipeline, per-shape
 * Hit construction + min_by) rather than the product's closed forms, so that
 * the two are independent statements of the same algorithm.
 *
 * Third-party arithmetic restated here (not vendored under /root/reference):
 *   nalgebra 0.16.10 (Cargo.lock:309-310): dot = x0*y0 + x1*y1 + x2*y2
 *     left-to-right; normalize = component-wise divide by sqrt(dot(v,v));
 *     standard cross product.
 *   rand 0.5.5 (Cargo.lock:445-446): the reference seeds ISAAC from OS entropy
 *     so no rand output is reproducible even in the reference.  Replaced by a
 *     documented counter-based generator (splitmix64 finaliser, DESIGN.md
 *     "RNG contract") meeting the same contract: U[0,1) doubles and an unbiased
 *     Fisher-Yates in rand 0.5's loop order (i = len-1 .. 1, swap(i, [0,i])).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 */
#include "flux_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* constants: fluxcore/src/constants.rs:4-5                            */
/* ------------------------------------------------------------------ */
#define T_MIN 0.0005
static const double PI = 3.14159265358979323846264338327950288; /* f64::consts::PI */
#define INV_PI (1.0 / PI)

/* ------------------------------------------------------------------ */
/* vectors (nalgebra 0.16.10 semantics, see header comment)            */
/* ------------------------------------------------------------------ */
typedef struct { double x, y, z; } v3;
typedef struct { double r, g, b; } rgb;

static inline v3 v3_new(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_new(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_new(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_scale(v3 a, double s) { return v3_new(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_div(v3 a, double s) { return v3_new(a.x / s, a.y / s, a.z / s); }
static inline v3 v3_neg(v3 a) { return v3_new(-a.x, -a.y, -a.z); }
static inline double v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_new(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 v3_normalize(v3 a) { return v3_div(a, sqrt(v3_dot(a, a))); }

/* Color ops: fluxcore/src/color.rs:47-105 */
static inline rgb rgb_new(double r, double g, double b) { rgb c = {r, g, b}; return c; }
static inline rgb rgb_mul(rgb a, rgb b) { return rgb_new(a.r * b.r, a.g * b.g, a.b * b.b); }
static inline rgb rgb_scale(rgb a, double s) { return rgb_new(a.r * s, a.g * s, a.b * s); }

/* Color::max_to_one: color.rs:35-44 */
static void max_to_one(rgb *c) {
    double mx1 = c->r > c->g ? c->r : c->g;
    double mx2 = mx1 > c->b ? mx1 : c->b;
    if (mx2 > 1.0) {
        double i = 1.0 / mx2;
        c->r *= i;
        c->g *= i;
        c->b *= i;
    }
}

/* ------------------------------------------------------------------ */
/* RNG contract (replaces rand 0.5.5 ISAAC; DESIGN.md "RNG contract")  */
/* ------------------------------------------------------------------ */
#define GOLDEN 0x9E3779B97F4A7C15ULL

static inline uint64_t mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}
static inline uint64_t fold(uint64_t k, uint64_t v) { return mix64((k ^ v) + GOLDEN); }

uint64_t fxo_rng_key(uint64_t seed, uint64_t kind, uint64_t a, uint64_t b, uint64_t sub) {
    return fold(fold(fold(fold(mix64(seed + GOLDEN), kind), a), b), sub);
}
uint64_t fxo_rng_draw(uint64_t key, uint64_t counter) {
    return mix64(key + (counter + 1) * GOLDEN);
}
/* Uniform::from(0.0..1.0) stand-in: 53 random mantissa bits, [0,1) */
double fxo_rng_unit(uint64_t key, uint64_t counter) {
    return (double)(fxo_rng_draw(key, counter) >> 11) * (1.0 / 9007199254740992.0);
}

/* sequential stream over one key */
typedef struct { uint64_t key, ctr; } stream;
static inline uint64_t stream_next(stream *s) { return fxo_rng_draw(s->key, s->ctr++); }
/* unbiased integer in [0,bound): Lemire multiply + rejection */
static uint64_t stream_below(stream *s, uint64_t bound) {
    uint64_t x = stream_next(s);
    __uint128_t m = (__uint128_t)x * bound;
    uint64_t l = (uint64_t)m;
    if (l < bound) {
        uint64_t t = (0 - bound) % bound;
        while (l < t) {
            x = stream_next(s);
            m = (__uint128_t)x * bound;
            l = (uint64_t)m;
        }
    }
    return (uint64_t)(m >> 64);
}

/* rand 0.5.5 Rng::shuffle loop order (call sites samplers/src/lib.rs:81-82,
 * 94,112 and fluxcore/src/sampling.rs:37-38) */
void fxo_shuffle(uint64_t key, int32_t *v, size_t n) {
    stream s = {key, 0};
    size_t i = n;
    while (i >= 2) {
        i -= 1;
        size_t j = (size_t)stream_below(&s, (uint64_t)i + 1);
        int32_t t = v[i];
        v[i] = v[j];
        v[j] = t;
    }
}

/* stream kinds / sub-streams (DESIGN.md "RNG contract") */
#define KIND_PIXEL 1
#define KIND_DISC 2
#define KIND_HEMI 3
#define KIND_ROWPERM 4
#define SUB_JITTER 0
/* CMJ: sub 1 = x_idxs, sub 2 = y_idxs.  MJ: sub 1+i = y-shuffle of row i,
 * sub 1+n+k = x-shuffle of column k. */

/* ------------------------------------------------------------------ */
/* samplers crate: samplers/src/lib.rs                                 */
/* ------------------------------------------------------------------ */
typedef struct { double x, y; } sq2;

/* grid_regular: lib.rs:184-191 (iproduct!: x outer, y inner) */
void fxo_grid_regular(int root, double *out) {
    double increment = 1.0 / (double)root;
    double start = 0.5 * increment;
    for (int i = 0; i < root; i++)
        for (int j = 0; j < root; j++) {
            out[2 * (i * root + j)] = start + increment * (double)i;
            out[2 * (i * root + j) + 1] = start + increment * (double)j;
        }
}

/* grid_jittered: lib.rs:35-44 (not used by the render loop; kept for the
 * sampler-debug style property tests) */
void fxo_grid_jittered(uint64_t key, int root, double *out) {
    double increment = 1.0 / (double)root;
    fxo_grid_regular(root, out);
    stream s = {key, 0};
    for (int p = 0; p < root * root; p++) {
        double a = (double)(stream_next(&s) >> 11) * (1.0 / 9007199254740992.0);
        double b = (double)(stream_next(&s) >> 11) * (1.0 / 9007199254740992.0);
        out[2 * p] = out[2 * p] + (a - 0.5) * increment;
        out[2 * p + 1] = out[2 * p + 1] + (b - 0.5) * increment;
    }
}

/* grid_multi_jittered_base: lib.rs:46-62.  base[i][j], a then b drawn
 * i-major from the jitter sub-stream (counter 2*(i*n+j), +1). */
static sq2 *mj_base(uint64_t seed, uint64_t kind, uint64_t a_, uint64_t b_, int root) {
    size_t n = (size_t)root;
    double r2 = (double)(n * n);
    double r_float = (double)root;
    sq2 *base = malloc(n * n * sizeof(sq2));
    stream s = {fxo_rng_key(seed, kind, a_, b_, SUB_JITTER), 0};
    for (size_t i = 0; i < n; i++) {
        double big_row = (double)i, little_col = (double)(n - 1 - i);
        for (size_t j = 0; j < n; j++) {
            double big_col = (double)j, little_row = (double)(n - 1 - j);
            double a = (double)(stream_next(&s) >> 11) * (1.0 / 9007199254740992.0);
            double b = (double)(stream_next(&s) >> 11) * (1.0 / 9007199254740992.0);
            base[i * n + j].x = (big_row / r_float) + (little_row + a) / r2;
            base[i * n + j].y = (big_col / r_float) + (little_col + b) / r2;
        }
    }
    return base;
}

/* shuffle_y: lib.rs:92-108.  out[k] = (vals[k].x, vals[idxs[k]].y) */
static void shuffle_y(const int32_t *idxs, const sq2 *vals, sq2 *out, size_t n) {
    for (size_t k = 0; k < n; k++) {
        out[k].x = vals[k].x;
        out[k].y = vals[idxs[k]].y;
    }
}
/* shuffle_x: lib.rs:110-126.  out[k] = (vals[idxs[k]].x, vals[k].y) */
static void shuffle_x(const int32_t *idxs, const sq2 *vals, sq2 *out, size_t n) {
    for (size_t k = 0; k < n; k++) {
        out[k].x = vals[idxs[k]].x;
        out[k].y = vals[k].y;
    }
}
/* transpose: lib.rs:204-215 (square n x n here) */
static void transpose(const sq2 *in, sq2 *out, size_t n) {
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < n; j++) out[i * n + j] = in[j * n + i];
}
static void iota(int32_t *v, size_t n) {
    for (size_t i = 0; i < n; i++) v[i] = (int32_t)i;
}

/* common tail of lib.rs:64-73 and :75-90.  correlated!=0: one shared x/y
 * permutation (sub-streams 1,2); else an independent permutation per row
 * (sub 1+i) and per column (sub 1+n+k).  The reference's correlated variant
 * also draws (and discards) one permutation per row/column (lib.rs:93-94,
 * 111-112); with a non-reproducible RNG those draws are unobservable and are
 * not restated. */
static void mj_finish(uint64_t seed, uint64_t kind, uint64_t a_, uint64_t b_, int root,
                      int correlated, double *out) {
    size_t n = (size_t)root;
    sq2 *samples = mj_base(seed, kind, a_, b_, root);
    sq2 *y_shuffled = malloc(n * n * sizeof(sq2));
    sq2 *t1 = malloc(n * n * sizeof(sq2));
    sq2 *t2 = malloc(n * n * sizeof(sq2));
    int32_t *x_idxs = malloc(n * sizeof(int32_t));
    int32_t *y_idxs = malloc(n * sizeof(int32_t));
    int32_t *tmp = malloc(n * sizeof(int32_t));

    if (correlated) {
        iota(x_idxs, n);
        iota(y_idxs, n);
        fxo_shuffle(fxo_rng_key(seed, kind, a_, b_, 1), x_idxs, n);
        fxo_shuffle(fxo_rng_key(seed, kind, a_, b_, 2), y_idxs, n);
    }
    /* y_shuffled = samples.iter().map(|vec| shuffle_y(..)) */
    for (size_t i = 0; i < n; i++) {
        const int32_t *idxs = y_idxs;
        if (!correlated) {
            iota(tmp, n);
            fxo_shuffle(fxo_rng_key(seed, kind, a_, b_, 1 + i), tmp, n);
            idxs = tmp;
        }
        shuffle_y(idxs, samples + i * n, y_shuffled + i * n, n);
    }
    /* x_shuffled = transpose(transpose(y_shuffled).map(|v| shuffle_x(..))) */
    transpose(y_shuffled, t1, n);
    for (size_t k = 0; k < n; k++) {
        const int32_t *idxs = x_idxs;
        if (!correlated) {
            iota(tmp, n);
            fxo_shuffle(fxo_rng_key(seed, kind, a_, b_, 1 + n + k), tmp, n);
            idxs = tmp;
        }
        shuffle_x(idxs, t1 + k * n, t2 + k * n, n);
    }
    transpose(t2, t1, n);
    /* concat_vec: lib.rs:193-202 -> flat index i*n + j */
    for (size_t p = 0; p < n * n; p++) {
        out[2 * p] = t1[p].x;
        out[2 * p + 1] = t1[p].y;
    }
    free(samples); free(y_shuffled); free(t1); free(t2);
    free(x_idxs); free(y_idxs); free(tmp);
}

/* grid_multi_jittered: lib.rs:64-73 */
void fxo_grid_multi_jittered(uint64_t seed, uint64_t kind, uint64_t a, uint64_t b, int root,
                             double *out) {
    mj_finish(seed, kind, a, b, root, 0, out);
}
/* grid_correlated_multi_jittered: lib.rs:75-90 */
void fxo_grid_correlated_multi_jittered(uint64_t seed, uint64_t kind, uint64_t a, uint64_t b,
                                        int root, double *out) {
    mj_finish(seed, kind, a, b, root, 1, out);
}

/* to_unit_hemi: lib.rs:133-142 */
static v3 to_unit_hemi(sq2 p, double e) {
    double cos_phi = cos(2.0 * PI * p.x);
    double sin_phi = sin(2.0 * PI * p.x);
    double cos_theta = pow(1.0 - p.y, 1.0 / (e + 1.0));
    double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
    double pu = sin_theta * cos_phi;
    double pv = sin_theta * sin_phi;
    double pw = cos_theta;
    return v3_normalize(v3_new(pu, pv, pw));
}
void fxo_to_unit_hemi(double x, double y, double e, double out[3]) {
    sq2 p = {x, y};
    v3 h = to_unit_hemi(p, e);
    out[0] = h.x; out[1] = h.y; out[2] = h.z;
}

/* to_poisson_disc: lib.rs:144-182 (Shirley concentric map) */
static sq2 to_poisson_disc(sq2 p) {
    double spx = 2.0 * p.x - 1.0;
    double spy = 2.0 * p.y - 1.0;
    double phi, r;
    if (spx > -spy) {
        if (spx > spy) {
            r = spx;
            phi = spy / spx;
        } else {
            r = spy;
            phi = 2.0 - spx / spy;
        }
    } else {
        if (spx < spy) {
            r = -spx;
            phi = 4.0 + spy / spx;
        } else {
            r = -spy;
            if (spy != 0.0) {
                phi = 6.0 - spx / spy;
            } else {
                phi = 0.0;
            }
        }
    }
    phi *= PI / 4.0;
    sq2 o = {r * cos(phi), r * sin(phi)};
    return o;
}
void fxo_to_poisson_disc(double x, double y, double out[2]) {
    sq2 p = {x, y};
    sq2 d = to_poisson_disc(p);
    out[0] = d.x; out[1] = d.y;
}

/* ------------------------------------------------------------------ */
/* scene                                                               */
/* ------------------------------------------------------------------ */
typedef struct {
    int kind;
    double p[8];
} material;

typedef struct {
    int kind;
    /* sphere */
    v3 center;
    double radius;
    int invert;
    v3 corner0, corner1; /* Sphere::new: shapes.rs:154-169 */
    /* plane */
    v3 point, normal;
    material mat;
} shape;

/* EXTENSION (no reference counterpart; scene.rs:71-74 is Sphere | Plane only): triangle with the
 * conventions of Plane (shapes.rs:135-152) -- two-sided, t > T_MIN, stored geometric normal. */
typedef struct {
    v3 v0, e1, e2, normal;
    material mat;
} triangle;

typedef struct { v3 origin, direction; } ray;

/* Hit: common.rs:7-14 */
typedef struct {
    v3 local_hit_point;
    v3 normal;
    const material *mat;
    double distance;
    ray r;
    int depth;
    int shape_index;
} hit;

struct fxo_ctx {
    /* SceneData / CameraData / OutputSettings: scene.rs:37-66 */
    v3 eye, look_at, up;
    double zoom_factor, view_plane_distance, focal_distance, lens_radius;
    int image_width, image_height;
    double pixel_size;
    rgb background;
    int num_shapes;
    shape *shapes;
    size_t num_tris, cap_tris; /* extension: triangles follow the shapes in hit order */
    triangle *tris;
    /* CameraBasis: scene.rs:22-35 */
    v3 u, v, w;
    /* JobConfiguration: job.rs:49-53 */
    int sample_root, max_trace_depth;
    uint64_t seed;
    /* MasterSampleSets: sampling.rs:5-10 */
    size_t num_sets, nsamp;
    double *pixel_sets; /* [S][N][2] */
    double *disc_sets;  /* [S][N][2] */
    double *hemi_sets;  /* [S][D][N][3] */
    uint64_t stats[8];
};

typedef struct { uint64_t s[8]; } stat_block;

/* CameraBasis::new: scene.rs:28-35 */
static void camera_basis(fxo_ctx *c) {
    c->w = v3_normalize(v3_sub(c->eye, c->look_at));
    c->u = v3_normalize(v3_cross(c->up, c->w));
    c->v = v3_cross(c->w, c->u);
}

/* file-local min/max of shapes.rs:90-96 (NaN behaviour follows this form) */
static inline double smin(double a, double b) { return a < b ? a : b; }
static inline double smax(double a, double b) { return a > b ? a : b; }

/* BoundingBox::hit: shapes.rs:98-133 */
static int bbox_hit(v3 corner0, v3 corner1, const ray *r) {
    double ox = r->origin.x, oy = r->origin.y, oz = r->origin.z;
    double dx = r->direction.x, dy = r->direction.y, dz = r->direction.z;
    double tx_min, tx_max, ty_min, ty_max, tz_min, tz_max;

    double a = 1.0 / dx;
    if (a >= 0.0) {
        tx_min = (corner0.x - ox) * a;
        tx_max = (corner1.x - ox) * a;
    } else {
        tx_min = (corner1.x - ox) * a;
        tx_max = (corner0.x - ox) * a;
    }
    double b = 1.0 / dy;
    if (b >= 0.0) {
        ty_min = (corner0.y - oy) * b;
        ty_max = (corner1.y - oy) * b;
    } else {
        ty_min = (corner1.y - oy) * b;
        ty_max = (corner0.y - oy) * b;
    }
    double c = 1.0 / dz;
    if (c >= 0.0) {
        tz_min = (corner0.z - oz) * c;
        tz_max = (corner1.z - oz) * c;
    } else {
        tz_min = (corner1.z - oz) * c;
        tz_max = (corner0.z - oz) * c;
    }
    double t0 = smax(tx_min, smax(ty_min, tz_min));
    double t1 = smin(tx_max, smin(ty_max, tz_max));
    return (t0 < t1 && t1 > T_MIN);
}

/* Plane::hit: shapes.rs:135-152 */
static int plane_hit(const shape *s, const ray *r, int depth, hit *h) {
    double t = v3_dot(v3_sub(s->point, r->origin), s->normal) / v3_dot(r->direction, s->normal);
    if (t > T_MIN) {
        h->r = *r;
        h->depth = depth;
        h->distance = t;
        h->normal = s->normal;
        h->local_hit_point = v3_add(r->origin, v3_scale(r->direction, t));
        h->mat = &s->mat;
        return 1;
    }
    return 0;
}

/* Sphere::hit: shapes.rs:171-217 */
static int sphere_hit(const shape *s, const ray *r, int depth, hit *h) {
    if (!bbox_hit(s->corner0, s->corner1, r)) return 0;
    v3 temp = v3_sub(r->origin, s->center);
    double a = v3_dot(r->direction, r->direction);
    double b = 2.0 * v3_dot(temp, r->direction);
    double c = v3_dot(temp, temp) - s->radius * s->radius;
    double disc = b * b - 4.0 * a * c;
    double invert_val = s->invert ? -1.0 : 1.0;
    if (disc < 0.0) return 0;
    double e = sqrt(disc);
    double denom = 2.0 * a;
    double t = (-b - e) / denom;
    if (t > T_MIN) {
        h->r = *r;
        h->distance = t;
        h->depth = depth;
        h->normal = v3_div(v3_scale(v3_add(temp, v3_scale(r->direction, t)), invert_val), s->radius);
        h->local_hit_point = v3_add(r->origin, v3_scale(r->direction, t));
        h->mat = &s->mat;
        return 1;
    }
    double t2 = (-b + e) / denom;
    if (t2 > T_MIN) {
        h->r = *r;
        h->distance = t2;
        h->depth = depth;
        h->normal = v3_div(v3_scale(v3_add(temp, v3_scale(r->direction, t2)), invert_val), s->radius);
        h->local_hit_point = v3_add(r->origin, v3_scale(r->direction, t2));
        h->mat = &s->mat;
        return 1;
    }
    return 0;
}

/* EXTENSION: Moeller-Trumbore, f64, no culling.  The product's BVH traversal must return exactly
 * what this brute-force definition returns (DESIGN.md "Triangles and the BVH"). */
static int triangle_hit(const triangle *tr, const ray *r, int depth, hit *h) {
    v3 p = v3_cross(r->direction, tr->e2);
    double det = v3_dot(tr->e1, p);
    if (det == 0.0) return 0;
    double inv = 1.0 / det;
    v3 s = v3_sub(r->origin, tr->v0);
    double u = v3_dot(s, p) * inv;
    if (u < 0.0 || u > 1.0) return 0;
    v3 q = v3_cross(s, tr->e1);
    double v = v3_dot(r->direction, q) * inv;
    if (v < 0.0 || u + v > 1.0) return 0;
    double t = v3_dot(tr->e2, q) * inv;
    if (t > T_MIN) {
        h->r = *r;
        h->depth = depth;
        h->distance = t;
        h->normal = tr->normal;
        h->local_hit_point = v3_add(r->origin, v3_scale(r->direction, t));
        h->mat = &tr->mat;
        return 1;
    }
    return 0;
}

/* Scene::hit: scene.rs:156-160 with Hit::compare (common.rs:17-23) under
 * Iterator::min_by: the accumulated minimum is replaced only when
 * compare(acc,new) == Greater, i.e. when !(acc.distance <= new.distance);
 * ties keep the earlier (lower YAML index) shape. */
static int scene_hit(const fxo_ctx *c, const ray *r, int depth, hit *best) {
    int found = 0;
    for (int i = 0; i < c->num_shapes; i++) {
        hit h;
        int ok = c->shapes[i].kind == FXO_SHAPE_SPHERE ? sphere_hit(&c->shapes[i], r, depth, &h)
                                                       : plane_hit(&c->shapes[i], r, depth, &h);
        if (!ok) continue;
        h.shape_index = i;
        if (!found) {
            *best = h;
            found = 1;
        } else if (!(best->distance <= h.distance)) {
            *best = h;
        }
    }
    /* extension: triangles continue the same ordered scan (indices num_shapes + k) */
    for (size_t k = 0; k < c->num_tris; k++) {
        hit h;
        if (!triangle_hit(&c->tris[k], r, depth, &h)) continue;
        h.shape_index = c->num_shapes + (int)k;
        if (!found) {
            *best = h;
            found = 1;
        } else if (!(best->distance <= h.distance)) {
            *best = h;
        }
    }
    return found;
}

static rgb shade(const fxo_ctx *c, const ray *r, int depth, size_t set_index, size_t sample_index,
                 stat_block *st);

static inline const double *hemi_at(const fxo_ctx *c, size_t set, size_t depth0, size_t idx) {
    return c->hemi_sets + ((set * (size_t)c->max_trace_depth + depth0) * c->nsamp + idx) * 3;
}

/* Lambertian::sample_f: brdf.rs:19-31 */
static void lambertian_sample_f(const material *m, v3 normal, v3 hemi_sample, v3 *wi, double *pdf,
                                rgb *f) {
    v3 w = normal;
    v3 v = v3_normalize(v3_cross(v3_new(0.0034, 1.0, 0.0071), w));
    v3 u = v3_cross(v, w);
    *wi = v3_normalize(v3_add(v3_add(v3_scale(u, hemi_sample.x), v3_scale(v, hemi_sample.y)),
                              v3_scale(w, hemi_sample.z)));
    *pdf = v3_dot(normal, *wi) * INV_PI;
    *f = rgb_scale(rgb_scale(rgb_new(m->p[0], m->p[1], m->p[2]), m->p[6]), INV_PI);
}

/* PerfectSpecular::sample_f: brdf.rs:38-46 */
static void perfect_specular_sample_f(const material *m, v3 normal, v3 wo, v3 *wi, double *pdf,
                                      rgb *f) {
    double ndotwo = v3_dot(normal, wo);
    *wi = v3_add(v3_neg(wo), v3_scale(v3_scale(normal, ndotwo), 2.0));
    *pdf = v3_dot(normal, *wi);
    *f = rgb_scale(rgb_new(m->p[0], m->p[1], m->p[2]), m->p[3]);
}

/* GlossySpecular::sample_f: brdf.rs:54-79 */
static void glossy_sample_f(const material *m, v3 normal, v3 wo, sq2 pixel_sample, v3 *wi,
                            double *pdf, rgb *f) {
    double exp_ = m->p[4];
    double ndotwo = v3_dot(normal, wo);
    v3 r = v3_add(v3_neg(wo), v3_scale(v3_scale(normal, ndotwo), 2.0));
    v3 w = r;
    v3 u = v3_normalize(v3_cross(v3_new(0.00424, 1.0, 0.00764), w));
    v3 v = v3_cross(u, w);
    v3 hs = to_unit_hemi(pixel_sample, exp_);
    v3 wi0 = v3_add(v3_add(v3_scale(u, hs.x), v3_scale(v, hs.y)), v3_scale(w, hs.z));
    if (v3_dot(normal, wi0) < 0.0) {
        *wi = v3_add(v3_sub(v3_scale(u, -hs.x), v3_scale(v, hs.y)), v3_scale(w, hs.z));
    } else {
        *wi = wi0;
    }
    double phong_lobe = pow(v3_dot(r, *wi), exp_);
    *pdf = phong_lobe * v3_dot(normal, *wi);
    *f = rgb_scale(rgb_scale(rgb_new(m->p[0], m->p[1], m->p[2]), m->p[3]), phong_lobe);
}

void fxo_sample_f(int mat_kind, const double *mat_params, const double n[3], const double wo[3],
                  const double hemi[3], const double sq[2], double wi[3], double *pdf,
                  double f[3]) {
    material m;
    m.kind = mat_kind;
    memcpy(m.p, mat_params, sizeof(m.p));
    v3 wi_ = {0, 0, 0};
    rgb f_ = {0, 0, 0};
    *pdf = 0.0;
    v3 N = v3_new(n[0], n[1], n[2]), WO = v3_new(wo[0], wo[1], wo[2]);
    if (mat_kind == FXO_MAT_MATTE) {
        lambertian_sample_f(&m, N, v3_new(hemi[0], hemi[1], hemi[2]), &wi_, pdf, &f_);
    } else if (mat_kind == FXO_MAT_REFLECTIVE) {
        perfect_specular_sample_f(&m, N, WO, &wi_, pdf, &f_);
    } else if (mat_kind == FXO_MAT_GLOSSY) {
        sq2 s = {sq[0], sq[1]};
        glossy_sample_f(&m, N, WO, s, &wi_, pdf, &f_);
    }
    wi[0] = wi_.x; wi[1] = wi_.y; wi[2] = wi_.z;
    f[0] = f_.r; f[1] = f_.g; f[2] = f_.b;
}

/* Material::path_shade: materials.rs:18-34 (Matte), :41-50 (Emissive),
 * :56-72 (Reflective wrapping PerfectSpecular or GlossySpecular) */
static rgb path_shade(const fxo_ctx *c, const hit *h, size_t set_index, size_t sample_index,
                      stat_block *st) {
    const material *m = h->mat;
    switch (m->kind) {
    case FXO_MAT_EMISSIVE: {
        st->s[5]++;
        if (v3_dot(v3_scale(h->normal, -1.0), h->r.direction) > 0.0)
            return rgb_scale(rgb_new(m->p[0], m->p[1], m->p[2]), m->p[3]);
        return rgb_new(0.0, 0.0, 0.0);
    }
    case FXO_MAT_MATTE: {
        st->s[2]++;
        const double *hp = hemi_at(c, set_index, (size_t)(h->depth - 1), sample_index);
        v3 hemi_sample = v3_new(hp[0], hp[1], hp[2]);
        v3 wi;
        double pdf;
        rgb f;
        lambertian_sample_f(m, h->normal, hemi_sample, &wi, &pdf, &f);
        double ndotwi = v3_dot(h->normal, wi);
        ray reflected = {h->local_hit_point, wi};
        rgb child = shade(c, &reflected, h->depth + 1, set_index, sample_index, st);
        return rgb_scale(rgb_mul(f, child), ndotwi / pdf);
    }
    case FXO_MAT_REFLECTIVE:
    case FXO_MAT_GLOSSY: {
        v3 wo = v3_scale(h->r.direction, -1.0);
        v3 wi;
        double pdf;
        rgb fr;
        if (m->kind == FXO_MAT_REFLECTIVE) {
            st->s[4]++;
            perfect_specular_sample_f(m, h->normal, wo, &wi, &pdf, &fr);
        } else {
            st->s[3]++;
            const double *pp = c->pixel_sets + (set_index * c->nsamp + sample_index) * 2;
            sq2 sq = {pp[0], pp[1]};
            glossy_sample_f(m, h->normal, wo, sq, &wi, &pdf, &fr);
        }
        ray reflected = {h->local_hit_point, wi};
        rgb child = shade(c, &reflected, h->depth + 1, set_index, sample_index, st);
        return rgb_scale(rgb_mul(fr, child), v3_dot(h->normal, wi) / pdf);
    }
    }
    return rgb_new(0.0, 0.0, 0.0);
}

/* Scene::shade: scene.rs:162-172 */
static rgb shade(const fxo_ctx *c, const ray *r, int depth, size_t set_index, size_t sample_index,
                 stat_block *st) {
    if (depth > c->max_trace_depth) {
        st->s[7]++;
        return rgb_new(0.0, 0.0, 0.0);
    }
    hit h;
    st->s[1]++;
    if (!scene_hit(c, r, depth, &h)) {
        st->s[6]++;
        return c->background;
    }
    return path_shade(c, &h, set_index, sample_index, st);
}

/* Camera::ray_direction: trace.rs:44-51 */
static v3 ray_direction(const fxo_ctx *c, double px, double py, double lx, double ly) {
    double factor = c->focal_distance / c->view_plane_distance;
    double px2 = px * factor;
    double py2 = py * factor;
    return v3_normalize(v3_sub(v3_add(v3_scale(c->u, px2 - lx), v3_scale(c->v, py2 - ly)),
                               v3_scale(c->w, c->focal_distance)));
}

/* MasterSampleSets::shuffle_indices: sampling.rs:35-40; the reference seeds a
 * fresh entropy ISAAC per row (trace.rs:64) -- here the stream is keyed by
 * (seed,row) so the image does not depend on how rows are sharded. */
void fxo_row_perm(const fxo_ctx *c, size_t row, int32_t *out) {
    iota(out, c->num_sets);
    fxo_shuffle(fxo_rng_key(c->seed, KIND_ROWPERM, row, 0, 0), out, c->num_sets);
}

static ray primary_ray(const fxo_ctx *c, size_t row, size_t col, size_t set, size_t index) {
    /* trace.rs:54-60,72-80 */
    int img_h = c->image_height, img_w = c->image_width;
    double half_img_h = (double)img_h * 0.5;
    double half_img_w = (double)img_w * 0.5;
    double adjusted_pixel_size = c->pixel_size / c->zoom_factor;
    const double *point = c->pixel_sets + (set * c->nsamp + index) * 2;
    const double *lens_sample = c->disc_sets + (set * c->nsamp + index) * 2;
    double u = adjusted_pixel_size * ((double)col - half_img_w + point[0]);
    double v = adjusted_pixel_size * ((double)((size_t)img_h - row) - half_img_h + point[1]);
    double lpx = lens_sample[0] * c->lens_radius;
    double lpy = lens_sample[1] * c->lens_radius;
    ray r;
    r.direction = ray_direction(c, u, v, lpx, lpy);
    r.origin = v3_add(v3_add(c->eye, v3_scale(c->u, lpx)), v3_scale(c->v, lpy));
    return r;
}

/* one row of Camera::render: trace.rs:63-91 */
static void render_row(const fxo_ctx *c, size_t row, double *out_row, stat_block *st) {
    size_t img_w = (size_t)c->image_width;
    size_t nn = (size_t)c->sample_root * (size_t)c->sample_root;
    double pixel_denom = 1.0 / (double)nn;
    int32_t *sample_set_indexes = malloc(c->num_sets * sizeof(int32_t));
    fxo_row_perm(c, row, sample_set_indexes);
    for (size_t col = 0; col < img_w; col++) {
        rgb color = {0.0, 0.0, 0.0};
        size_t set = (size_t)sample_set_indexes[col] % c->num_sets;
        for (size_t index = 0; index < nn; index++) {
            ray r = primary_ray(c, row, col, set, index);
            st->s[0]++;
            rgb s = shade(c, &r, 1, set, index, st);
            color.r += s.r;
            color.g += s.g;
            color.b += s.b;
        }
        color.r *= pixel_denom;
        color.g *= pixel_denom;
        color.b *= pixel_denom;
        max_to_one(&color);
        out_row[3 * col] = color.r;
        out_row[3 * col + 1] = color.g;
        out_row[3 * col + 2] = color.b;
    }
    free(sample_set_indexes);
}

/* ------------------------------------------------------------------ */
/* context                                                             */
/* ------------------------------------------------------------------ */
typedef struct {
    fxo_ctx *c;
    size_t begin, end;
} table_job;

/* MasterSampleSets::new: sampling.rs:13-33 for sets [begin,end) */
static void *build_tables_range(void *arg) {
    table_job *j = arg;
    fxo_ctx *c = j->c;
    size_t N = c->nsamp, D = (size_t)c->max_trace_depth;
    double *tmp = malloc(N * 2 * sizeof(double));
    for (size_t s = j->begin; s < j->end; s++) {
        /* pixel_sets: CMJ */
        fxo_grid_correlated_multi_jittered(c->seed, KIND_PIXEL, s, 0, c->sample_root,
                                           c->pixel_sets + s * N * 2);
        /* disc_sets: to_poisson_disc(CMJ) -- a separate draw */
        fxo_grid_correlated_multi_jittered(c->seed, KIND_DISC, s, 0, c->sample_root, tmp);
        for (size_t p = 0; p < N; p++) {
            sq2 q = {tmp[2 * p], tmp[2 * p + 1]};
            sq2 d = to_poisson_disc(q);
            c->disc_sets[(s * N + p) * 2] = d.x;
            c->disc_sets[(s * N + p) * 2 + 1] = d.y;
        }
        /* hemi_sets: to_hemisphere(MJ, 0.0) per depth */
        for (size_t d = 0; d < D; d++) {
            fxo_grid_multi_jittered(c->seed, KIND_HEMI, s, d, c->sample_root, tmp);
            for (size_t p = 0; p < N; p++) {
                sq2 q = {tmp[2 * p], tmp[2 * p + 1]};
                v3 h = to_unit_hemi(q, 0.0);
                double *o = c->hemi_sets + ((s * D + d) * N + p) * 3;
                o[0] = h.x; o[1] = h.y; o[2] = h.z;
            }
        }
    }
    free(tmp);
    return NULL;
}

fxo_ctx *fxo_ctx_create(const double *camera, int image_width, int image_height,
                        double pixel_size, const double *background, int num_shapes,
                        const int32_t *shape_kinds, const double *shape_params,
                        const int32_t *mat_kinds, const double *mat_params, int sample_root,
                        int max_trace_depth, uint64_t seed) {
    if (sample_root < 1 || max_trace_depth < 1 || image_width < 1 || image_height < 1 ||
        num_shapes < 0)
        return NULL;
    fxo_ctx *c = calloc(1, sizeof(*c));
    c->eye = v3_new(camera[0], camera[1], camera[2]);
    c->look_at = v3_new(camera[3], camera[4], camera[5]);
    c->up = v3_new(camera[6], camera[7], camera[8]);
    c->zoom_factor = camera[9];
    c->view_plane_distance = camera[10];
    c->focal_distance = camera[11];
    c->lens_radius = camera[12];
    c->image_width = image_width;
    c->image_height = image_height;
    c->pixel_size = pixel_size;
    c->background = rgb_new(background[0], background[1], background[2]);
    c->num_shapes = num_shapes;
    c->shapes = calloc((size_t)(num_shapes > 0 ? num_shapes : 1), sizeof(shape));
    for (int i = 0; i < num_shapes; i++) {
        shape *s = &c->shapes[i];
        const double *p = shape_params + 8 * i;
        s->kind = shape_kinds[i];
        if (s->kind == FXO_SHAPE_SPHERE) {
            s->center = v3_new(p[0], p[1], p[2]);
            s->radius = p[3];
            s->invert = p[4] != 0.0;
            /* Sphere::new: shapes.rs:154-169 */
            v3 delta = v3_new(s->radius, s->radius, s->radius);
            s->corner0 = v3_sub(s->center, delta);
            s->corner1 = v3_add(s->center, delta);
        } else {
            s->point = v3_new(p[0], p[1], p[2]);
            s->normal = v3_new(p[3], p[4], p[5]);
        }
        s->mat.kind = mat_kinds[i];
        memcpy(s->mat.p, mat_params + 8 * i, 8 * sizeof(double));
    }
    camera_basis(c);
    c->sample_root = sample_root;
    c->max_trace_depth = max_trace_depth;
    c->seed = seed;
    /* workers.rs:47-54: num_sets = image_width */
    c->num_sets = (size_t)image_width;
    c->nsamp = (size_t)sample_root * (size_t)sample_root;
    size_t S = c->num_sets, N = c->nsamp, D = (size_t)max_trace_depth;
    c->pixel_sets = malloc(S * N * 2 * sizeof(double));
    c->disc_sets = malloc(S * N * 2 * sizeof(double));
    c->hemi_sets = malloc(S * D * N * 3 * sizeof(double));
    if (!c->pixel_sets || !c->disc_sets || !c->hemi_sets) {
        fxo_ctx_destroy(c);
        return NULL;
    }
    /* The reference builds the tables single-threaded (sampling.rs:13-33);
     * sets are independent streams here so a few threads may split them. */
    int nt = 8;
    if ((size_t)nt > S) nt = (int)S;
    pthread_t th[8];
    table_job jobs[8];
    for (int t = 0; t < nt; t++) {
        jobs[t].c = c;
        jobs[t].begin = S * (size_t)t / (size_t)nt;
        jobs[t].end = S * (size_t)(t + 1) / (size_t)nt;
        pthread_create(&th[t], NULL, build_tables_range, &jobs[t]);
    }
    for (int t = 0; t < nt; t++) pthread_join(th[t], NULL);
    return c;
}

int fxo_ctx_add_mesh(fxo_ctx *c, const double *vertices, size_t num_vertices, const uint32_t *indices,
                     size_t num_triangles, int mat_kind, const double *mat_params) {
    if (!c || (num_triangles && (!vertices || !indices))) return -1;
    for (size_t k = 0; k < 3 * num_triangles; k++)
        if (indices[k] >= num_vertices) return -1;
    if (c->num_tris + num_triangles > c->cap_tris) {
        size_t cap = (c->num_tris + num_triangles) * 2;
        triangle *t = realloc(c->tris, cap * sizeof(triangle));
        if (!t) return -1;
        c->tris = t;
        c->cap_tris = cap;
    }
    for (size_t k = 0; k < num_triangles; k++) {
        triangle *t = &c->tris[c->num_tris + k];
        const double *a = vertices + 3 * (size_t)indices[3 * k];
        const double *b = vertices + 3 * (size_t)indices[3 * k + 1];
        const double *d = vertices + 3 * (size_t)indices[3 * k + 2];
        t->v0 = v3_new(a[0], a[1], a[2]);
        t->e1 = v3_sub(v3_new(b[0], b[1], b[2]), t->v0);
        t->e2 = v3_sub(v3_new(d[0], d[1], d[2]), t->v0);
        {
            /* extension rule: a triangle whose e1 x e2 is exactly zero (repeated or exactly collinear vertices)
             * has no surface and is never hit -- its edges are cleared, so det == 0 in fxo tri test */
            v3 nn = v3_cross(t->e1, t->e2);
            if (nn.x == 0.0 && nn.y == 0.0 && nn.z == 0.0) {
                t->e1 = v3_new(0.0, 0.0, 0.0);
                t->e2 = v3_new(0.0, 0.0, 0.0);
                t->normal = v3_new(0.0, 0.0, 0.0);
            } else {
                t->normal = v3_normalize(nn);
            }
        }
        t->mat.kind = mat_kind;
        memcpy(t->mat.p, mat_params, 8 * sizeof(double));
    }
    c->num_tris += num_triangles;
    return 0;
}

void fxo_ctx_destroy(fxo_ctx *c) {
    if (!c) return;
    free(c->tris);
    free(c->shapes);
    free(c->pixel_sets);
    free(c->disc_sets);
    free(c->hemi_sets);
    free(c);
}

const double *fxo_pixel_sets(const fxo_ctx *c) { return c->pixel_sets; }
const double *fxo_disc_sets(const fxo_ctx *c) { return c->disc_sets; }
const double *fxo_hemi_sets(const fxo_ctx *c) { return c->hemi_sets; }
void fxo_camera_basis(const fxo_ctx *c, double uvw[9]) {
    uvw[0] = c->u.x; uvw[1] = c->u.y; uvw[2] = c->u.z;
    uvw[3] = c->v.x; uvw[4] = c->v.y; uvw[5] = c->v.z;
    uvw[6] = c->w.x; uvw[7] = c->w.y; uvw[8] = c->w.z;
}
void fxo_stats(const fxo_ctx *c, uint64_t out[8]) { memcpy(out, c->stats, sizeof(c->stats)); }
void fxo_stats_reset(fxo_ctx *c) { memset(c->stats, 0, sizeof(c->stats)); }

/* ------------------------------------------------------------------ */
/* row rendering, optionally over a pthread pool (the reference uses a
 * rayon par_iter over the rows of one unit, trace.rs:62-63)           */
/* ------------------------------------------------------------------ */
typedef struct {
    fxo_ctx *c;
    const int32_t *rows;
    size_t nrows;
    double *out;
    size_t next; /* guarded by mu */
    pthread_mutex_t mu;
    stat_block total;
} row_pool;

static void *row_worker(void *arg) {
    row_pool *p = arg;
    stat_block st;
    memset(&st, 0, sizeof(st));
    size_t W = (size_t)p->c->image_width;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        size_t k = p->next++;
        pthread_mutex_unlock(&p->mu);
        if (k >= p->nrows) break;
        render_row(p->c, (size_t)p->rows[k], p->out + k * W * 3, &st);
    }
    pthread_mutex_lock(&p->mu);
    for (int i = 0; i < 8; i++) p->total.s[i] += st.s[i];
    pthread_mutex_unlock(&p->mu);
    return NULL;
}

int fxo_render_row_list(fxo_ctx *c, const int32_t *rows, size_t nrows, double *out, int threads) {
    if (!c || !out) return -1;
    for (size_t k = 0; k < nrows; k++)
        if (rows[k] < 0 || rows[k] >= c->image_height) return -1;
    row_pool p;
    memset(&p, 0, sizeof(p));
    p.c = c;
    p.rows = rows;
    p.nrows = nrows;
    p.out = out;
    pthread_mutex_init(&p.mu, NULL);
    if (threads <= 1) {
        row_worker(&p);
    } else {
        if (threads > 256) threads = 256;
        pthread_t th[256];
        for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, row_worker, &p);
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    }
    for (int i = 0; i < 8; i++) c->stats[i] += p.total.s[i];
    pthread_mutex_destroy(&p.mu);
    return 0;
}

int fxo_render_rows(fxo_ctx *c, size_t row_start, size_t row_end, double *out, int threads) {
    if (!c || row_end < row_start || row_end >= (size_t)c->image_height) return -1;
    size_t n = row_end - row_start + 1;
    int32_t *rows = malloc(n * sizeof(int32_t));
    for (size_t k = 0; k < n; k++) rows[k] = (int32_t)(row_start + k);
    int rc = fxo_render_row_list(c, rows, n, out, threads);
    free(rows);
    return rc;
}

/* ------------------------------------------------------------------ */
/* unit-level entry points                                             */
/* ------------------------------------------------------------------ */
void fxo_max_to_one(double c[3]) {
    rgb x = {c[0], c[1], c[2]};
    max_to_one(&x);
    c[0] = x.r; c[1] = x.g; c[2] = x.b;
}

int fxo_bbox_hit(const double c0[3], const double c1[3], const double o[3], const double d[3]) {
    ray r = {v3_new(o[0], o[1], o[2]), v3_new(d[0], d[1], d[2])};
    return bbox_hit(v3_new(c0[0], c0[1], c0[2]), v3_new(c1[0], c1[1], c1[2]), &r);
}

static void hit_out(const hit *h, double *t, double normal[3], double point[3]) {
    *t = h->distance;
    normal[0] = h->normal.x; normal[1] = h->normal.y; normal[2] = h->normal.z;
    point[0] = h->local_hit_point.x; point[1] = h->local_hit_point.y;
    point[2] = h->local_hit_point.z;
}

int fxo_sphere_hit(const double center[3], double radius, int invert, const double o[3],
                   const double d[3], double *t, double normal[3], double point[3]) {
    shape s;
    memset(&s, 0, sizeof(s));
    s.kind = FXO_SHAPE_SPHERE;
    s.center = v3_new(center[0], center[1], center[2]);
    s.radius = radius;
    s.invert = invert;
    v3 delta = v3_new(radius, radius, radius);
    s.corner0 = v3_sub(s.center, delta);
    s.corner1 = v3_add(s.center, delta);
    ray r = {v3_new(o[0], o[1], o[2]), v3_new(d[0], d[1], d[2])};
    hit h;
    if (!sphere_hit(&s, &r, 1, &h)) return 0;
    hit_out(&h, t, normal, point);
    return 1;
}

int fxo_plane_hit(const double p[3], const double n[3], const double o[3], const double d[3],
                  double *t, double normal[3], double point[3]) {
    shape s;
    memset(&s, 0, sizeof(s));
    s.kind = FXO_SHAPE_PLANE;
    s.point = v3_new(p[0], p[1], p[2]);
    s.normal = v3_new(n[0], n[1], n[2]);
    ray r = {v3_new(o[0], o[1], o[2]), v3_new(d[0], d[1], d[2])};
    hit h;
    if (!plane_hit(&s, &r, 1, &h)) return 0;
    hit_out(&h, t, normal, point);
    return 1;
}

int fxo_scene_hit(const fxo_ctx *c, const double o[3], const double d[3], double *t,
                  double normal[3], double point[3]) {
    ray r = {v3_new(o[0], o[1], o[2]), v3_new(d[0], d[1], d[2])};
    hit h;
    if (!scene_hit(c, &r, 1, &h)) return -1;
    hit_out(&h, t, normal, point);
    return h.shape_index;
}

void fxo_shade(fxo_ctx *c, const double o[3], const double d[3], int depth, size_t set_index,
               size_t sample_index, double out[3]) {
    ray r = {v3_new(o[0], o[1], o[2]), v3_new(d[0], d[1], d[2])};
    stat_block st;
    memset(&st, 0, sizeof(st));
    rgb s = shade(c, &r, depth, set_index, sample_index, &st);
    out[0] = s.r; out[1] = s.g; out[2] = s.b;
}

void fxo_primary_ray(const fxo_ctx *c, size_t row, size_t col, size_t set_index,
                     size_t sample_index, double o[3], double d[3]) {
    ray r = primary_ray(c, row, col, set_index, sample_index);
    o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
    d[0] = r.direction.x; d[1] = r.direction.y; d[2] = r.direction.z;
}

/* Job::work_units: job.rs:65-88 (note the `i < H - 1` loop guard) */
size_t fxo_work_units(size_t image_height, size_t rows_per_unit, size_t *starts, size_t *ends,
                      size_t cap) {
    if (rows_per_unit == 0) return (size_t)-1; /* the reference panics */
    size_t count = 0, i = 0;
    while (i < image_height - 1) {
        size_t remaining_rows = image_height - i;
        size_t num_rows = rows_per_unit < remaining_rows ? rows_per_unit : remaining_rows;
        if (count < cap) {
            starts[count] = i;
            ends[count] = i + num_rows - 1;
        }
        count++;
        i += num_rows;
    }
    return count;
}

/* `(c * 65535.99) as u16`: image.rs:50-53 (Rust float->int casts saturate,
 * NaN -> 0) */
uint16_t fxo_ppm_quantize(double c) {
    double v = c * 65535.99;
    if (!(v > 0.0)) return 0;
    if (v >= 65535.0) return 65535;
    return (uint16_t)v;
}

/* Image::write: image.rs:43-61 */
int fxo_write_ppm(const char *path, const double *rgbv, size_t width, size_t height,
                  const uint8_t *rows_present) {
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    fprintf(f, "P3\n%zu %zu\n65535\n", width, height);
    for (size_t r = 0; r < height; r++) {
        int present = rows_present ? rows_present[r] : 1;
        for (size_t col = 0; col < width; col++) {
            if (present) {
                const double *p = rgbv + (r * width + col) * 3;
                fprintf(f, "%u %u %u\n", fxo_ppm_quantize(p[0]), fxo_ppm_quantize(p[1]),
                        fxo_ppm_quantize(p[2]));
            } else {
                fprintf(f, "0 0 0\n");
            }
        }
    }
    fclose(f);
    return 0;
}
