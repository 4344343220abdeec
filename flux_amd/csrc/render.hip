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
//  * one lane = one camera path; a wave owns one pixel (n*n >= 64) and walks its
//    samples 64 at a time, so pixel/lens/hemisphere table reads are contiguous
//    across the wave (16 B or 8 B per lane);
//  * the scene (<= a few dozen shapes) is read with a wave-uniform index ->
//    scalar loads, operands sit in SGPRs; only the nearest hit's shape and
//    material are gathered per lane;
//  * the reference's recursion  L = (f1 (*) ((f2 (*) (...)) * s2)) * s1  is run
//    iteratively: each bounce pushes (f, s = n.wi/pdf) on a per-lane stack in
//    LDS ([level][4][lane], conflict-free) and the stack is folded from the
//    deepest bounce outward when the path ends, reproducing the reference's
//    multiplication order;
//  * REFILL variant: a lane whose path ended immediately takes the pixel's next
//    unstarted sample (ballot + mbcnt prefix), so lanes stay busy although path
//    lengths differ (1..D segments);
//  * per-lane partial sums are combined in lane order by the pixel's leader
//    lane (fixed order => bit-reproducible run to run), then * 1/n^2 and
//    max_to_one, and the 3 doubles are written once.  No atomics.
//
// Compiled with -ffp-contract=off: the operation order below is the
// reference's, and a*b+c is not fused (rustc never contracts).
#include "flux_device.h"
#include "flux_tables.h"
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

namespace flux {

struct Ray {
    double ox, oy, oz, dx, dy, dz;
};

struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 mk(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ V3 normalize(V3 a) {
    double len = sqrt(dot(a, a));
    return mk(a.x / len, a.y / len, a.z / len);
}

// trace.rs:72-80 + ray_direction trace.rs:44-51
__device__ __forceinline__ Ray primary_ray(const RenderParams &P, int row, int col, double2 sq,
                                           double2 lens) {
    double u = P.aps * (((double)col - P.half_w) + sq.x);
    double v = P.aps * (((double)(P.img_h - row) - P.half_h) + sq.y);
    double lpx = lens.x * P.lens_radius;
    double lpy = lens.y * P.lens_radius;
    double a = u * P.factor - lpx;
    double b = v * P.factor - lpy;
    V3 dir = normalize(mk((a * P.Ux + b * P.Vx) - P.focal * P.Wx, (a * P.Uy + b * P.Vy) - P.focal * P.Wy,
                          (a * P.Uz + b * P.Vz) - P.focal * P.Wz));
    Ray r;
    r.dx = dir.x;
    r.dy = dir.y;
    r.dz = dir.z;
    r.ox = (P.ex + lpx * P.Ux) + lpy * P.Vx;
    r.oy = (P.ey + lpx * P.Uy) + lpy * P.Vy;
    r.oz = (P.ez + lpx * P.Uz) + lpy * P.Vz;
    return r;
}

// Scene::hit (scene.rs:156-160): nearest of all shapes; a later shape replaces
// the current best only when !(best <= t) (Hit::compare under min_by,
// common.rs:17-23), so ties keep the lower index.  Returns -1 on miss.
// path statistics (STATS builds only)
struct Stats {
    unsigned c[10];
};

// ---- extension: triangles (no reference counterpart; DESIGN.md "Triangles and the BVH") ----------
// Moeller-Trumbore, f64, two-sided; the CPU checker restates the same sequence of operations.
__device__ __forceinline__ bool tri_hit(const DevTri &T, const Ray &r, double &t) {
    const V3 d = mk(r.dx, r.dy, r.dz);
    const V3 e1 = mk(T.e1x, T.e1y, T.e1z), e2 = mk(T.e2x, T.e2y, T.e2z);
    const V3 p = cross(d, e2);
    const double det = dot(e1, p);
    if (det == 0.0) return false;
    const double inv = 1.0 / det;
    const V3 s = mk(r.ox - T.v0x, r.oy - T.v0y, r.oz - T.v0z);
    const double u = dot(s, p) * inv;
    if (u < 0.0 || u > 1.0) return false;
    const V3 q = cross(s, e1);
    const double v = dot(d, q) * inv;
    if (v < 0.0 || u + v > 1.0) return false;
    t = dot(e2, q) * inv;
    return t > kTMin;
}

// nearest-hit rule of Scene::hit for candidates visited in ANY order: smaller t wins, equal t keeps
// the lower hit-order index (what min_by + Hit::compare give for an ordered scan)
__device__ __forceinline__ void consider(double t, int id, int slot, int &best, int &bslot, double &tb) {
    if (best < 0 || t < tb || (t == tb && id < best)) {
        best = id;
        bslot = slot;
        tb = t;
    }
}

// conservative slab test against an f32 box (already padded by the builder); inv* are finite
__device__ __forceinline__ bool box_hit(const float *lo, const float *hi, const Ray &r, double ix, double iy,
                                        double iz, double tb, bool have, double &tnear) {
    const double x0 = ((double)lo[0] - r.ox) * ix, x1 = ((double)hi[0] - r.ox) * ix;
    const double y0 = ((double)lo[1] - r.oy) * iy, y1 = ((double)hi[1] - r.oy) * iy;
    const double z0 = ((double)lo[2] - r.oz) * iz, z1 = ((double)hi[2] - r.oz) * iz;
    const double tn = fmax(fmax(fmin(x0, x1), fmin(y0, y1)), fmin(z0, z1));
    const double tf = fmin(fmin(fmax(x0, x1), fmax(y0, y1)), fmax(z0, z1));
    tnear = tn;
    return tn <= tf && tf >= 0.0 && (!have || tn <= tb);
}

template <bool STATS>
__device__ __forceinline__ void bvh_traverse(const RenderParams &P, const Ray &r, int *tstack, int stride,
                                             int &best, int &bslot, double &tb, Stats &st) {
    // reciprocal direction with zero components nudged to +-1e-300: no inf*0 = NaN in the slab test
    const double tiny = 1e-300;
    const double ix = 1.0 / (fabs(r.dx) < tiny ? copysign(tiny, r.dx) : r.dx);
    const double iy = 1.0 / (fabs(r.dy) < tiny ? copysign(tiny, r.dy) : r.dy);
    const double iz = 1.0 / (fabs(r.dz) < tiny ? copysign(tiny, r.dz) : r.dz);
    int sp = 0;
    int cur = 0;
    for (;;) {
        const DevNode N = P.nodes[cur];
        if (STATS) st.c[8]++;
        double tn0, tn1;
        bool h0 = box_hit(N.lo0, N.hi0, r, ix, iy, iz, tb, best >= 0, tn0);
        bool h1 = box_hit(N.lo1, N.hi1, r, ix, iy, iz, tb, best >= 0, tn1);
        if (h0 && N.child0 < 0) {
            const int first = ~N.child0;
            for (int k = 0; k < N.count0; ++k) {
                double t;
                if (STATS) st.c[9]++;
                if (tri_hit(P.tris[first + k], r, t)) consider(t, P.tris[first + k].id, first + k, best, bslot, tb);
            }
            h0 = false;
        }
        if (h1 && N.child1 < 0) {
            const int first = ~N.child1;
            for (int k = 0; k < N.count1; ++k) {
                double t;
                if (STATS) st.c[9]++;
                if (tri_hit(P.tris[first + k], r, t)) consider(t, P.tris[first + k].id, first + k, best, bslot, tb);
            }
            h1 = false;
        }
        if (h0 && h1) {
            const bool near0 = tn0 <= tn1;
            tstack[sp * stride] = near0 ? N.child1 : N.child0;
            ++sp;
            cur = near0 ? N.child0 : N.child1;
        } else if (h0) {
            cur = N.child0;
        } else if (h1) {
            cur = N.child1;
        } else {
            if (sp == 0) break;
            --sp;
            cur = tstack[sp * stride];
        }
    }
}

template <bool STATS, bool TRIS>
__device__ __forceinline__ int scene_hit(const RenderParams &P, const Ray &r, double &t_out, int &slot_out,
                                         int *tstack, int stride, Stats &st) {
    // BoundingBox::hit's reciprocals depend on the ray only (shapes.rs:107,114,121)
    const double ax = 1.0 / r.dx, ay = 1.0 / r.dy, az = 1.0 / r.dz;
    const double a = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;  // shapes.rs:177
    const double denom = 2.0 * a;                               // shapes.rs:187
    int best = -1;
    int bslot = -1;
    double tb = 0.0;
    // The reference scans every shape and runs BoundingBox::hit before each sphere (shapes.rs:173).
    // Same tests, regrouped for the machine: phase 1 walks the shapes with a wave-uniform index
    // (operands in SGPRs, all lanes active) doing the plane test and only the box test of each sphere;
    // phase 2 runs the sphere quadratic for each lane's OWN candidates, lowest index first, so the
    // sqrt/divides execute with most lanes active instead of once per shape under a sparse mask.
    // consider() applies min_by's rule order-independently (smaller t; equal t -> lower index).
    for (int base = 0; base < P.n_shapes; base += 32) {
        const int lim = (P.n_shapes - base) < 32 ? (P.n_shapes - base) : 32;
        uint32_t cand = 0;
        for (int k = 0; k < lim; ++k) {
            const DevShape &S = P.shapes[base + k];
            if (S.kind == kShapeSphere) {
                // BoundingBox::hit: shapes.rs:98-133
                double lox = (S.c0x - r.ox) * ax, hix = (S.c1x - r.ox) * ax;
                double loy = (S.c0y - r.oy) * ay, hiy = (S.c1y - r.oy) * ay;
                double loz = (S.c0z - r.oz) * az, hiz = (S.c1z - r.oz) * az;
                double tx_min = ax >= 0.0 ? lox : hix, tx_max = ax >= 0.0 ? hix : lox;
                double ty_min = ay >= 0.0 ? loy : hiy, ty_max = ay >= 0.0 ? hiy : loy;
                double tz_min = az >= 0.0 ? loz : hiz, tz_max = az >= 0.0 ? hiz : loz;
                double m0 = ty_min > tz_min ? ty_min : tz_min;  // max(a,b) = a > b ? a : b (shapes.rs:94-96)
                double t0 = tx_min > m0 ? tx_min : m0;
                double m1 = ty_max < tz_max ? ty_max : tz_max;  // min(a,b) = a < b ? a : b (shapes.rs:90-92)
                double t1 = tx_max < m1 ? tx_max : m1;
                if (t0 < t1 && t1 > kTMin) cand |= 1u << k;
            } else {
                // Plane::hit: shapes.rs:135-152 (normal stored in c0)
                double num = (S.px - r.ox) * S.c0x + (S.py - r.oy) * S.c0y + (S.pz - r.oz) * S.c0z;
                double den = r.dx * S.c0x + r.dy * S.c0y + r.dz * S.c0z;
                double t = num / den;
                if (t > kTMin) consider(t, base + k, -1, best, bslot, tb);
            }
        }
        while (cand) {
            const int k = __builtin_ctz(cand);
            cand &= cand - 1;
            const DevShape *S = P.shapes + (base + k);  // per-lane gather of 32 contiguous bytes
            // Sphere::hit: shapes.rs:176-214
            V3 temp = mk(r.ox - S->px, r.oy - S->py, r.oz - S->pz);
            double b = 2.0 * (temp.x * r.dx + temp.y * r.dy + temp.z * r.dz);
            double c = dot(temp, temp) - S->rr;
            double disc = b * b - 4.0 * a * c;
            if (!(disc < 0.0)) {
                double e = sqrt(disc);
                double t = (-b - e) / denom;
                bool ok = t > kTMin;
                if (!ok) {
                    t = (-b + e) / denom;
                    ok = t > kTMin;
                }
                if (ok) consider(t, base + k, -1, best, bslot, tb);
            }
        }
    }
    if (TRIS) {
        // triangles continue the scan with hit-order indices n_shapes + k
        if (P.bvh_stack > 0) {
            bvh_traverse<STATS>(P, r, tstack, stride, best, bslot, tb, st);
        } else {
            for (int k = 0; k < P.n_tris; ++k) {  // brute force, wave-uniform index
                double t;
                if (STATS) st.c[9]++;
                if (tri_hit(P.tris[k], r, t)) consider(t, P.tris[k].id, k, best, bslot, tb);
            }
        }
    }
    t_out = tb;
    slot_out = bslot;
    return best;
}

// to_unit_hemi: samplers/src/lib.rs:133-142
__device__ __forceinline__ V3 to_unit_hemi(double2 p, double inv_e1) {
    double phi = 2.0 * kPi * p.x;
    double sin_phi, cos_phi;
    sincos(phi, &sin_phi, &cos_phi);
    double cos_theta = pow(1.0 - p.y, inv_e1);
    double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
    return normalize(mk(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta));
}

// Per-lane path state.
struct Path {
    Ray r;
    int depth;    // 1-based depth of the segment being traced (scene.rs:162)
    int nb;       // entries on this lane's (f,s) stack
    double2 sq;   // pixel_sets[set][i] (also the glossy lobe sample, brdf.rs:64)
};

// One Scene::shade level (scene.rs:162-172) for a live lane.  Returns true if
// the path continues (a bounce was pushed and `p.r` now holds the child ray);
// otherwise (Lr,Lg,Lb) is the value this level returns.
template <bool STATS, bool TRIS>
__device__ __forceinline__ bool shade_level(const RenderParams &P, Path &p, uint32_t set, uint32_t i,
                                            double *stk, int stk_stride, int *tstack, double &Lr, double &Lg,
                                            double &Lb, Stats &st) {
    if (p.depth > P.max_depth) {  // scene.rs:164-165
        Lr = Lg = Lb = 0.0;
        if (STATS) st.c[7]++;
        return false;
    }
    if (STATS) st.c[1]++;
    double t;
    int slot;
    const int hit = scene_hit<STATS, TRIS>(P, p.r, t, slot, tstack, stk_stride, st);
    if (hit < 0) {  // scene.rs:168
        Lr = P.bgr;
        Lg = P.bgg;
        Lb = P.bgb;
        if (STATS) st.c[6]++;
        return false;
    }
    // Hit fields of the winning shape (shapes.rs:140-146,191-197): gathered per lane
    const bool is_tri = TRIS && slot >= 0;
    const DevShape *S = P.shapes + (is_tri ? 0 : hit);
    const DevMaterial *M = P.mats + (is_tri ? P.tris[slot].mat : hit);
    const V3 d = mk(p.r.dx, p.r.dy, p.r.dz);
    const V3 pt = mk(p.r.ox + t * d.x, p.r.oy + t * d.y, p.r.oz + t * d.z);
    V3 n;
    if (is_tri) {
        n = mk(P.tris[slot].nx, P.tris[slot].ny, P.tris[slot].nz);
    } else if (S->kind == kShapeSphere) {
        const double inv = S->inv, rad = S->radius;
        n = mk(((p.r.ox - S->px) + t * d.x) * inv / rad, ((p.r.oy - S->py) + t * d.y) * inv / rad,
               ((p.r.oz - S->pz) + t * d.z) * inv / rad);
    } else {
        n = mk(S->c0x, S->c0y, S->c0z);
    }
    const int kind = M->kind;
    if (kind == kMatEmissive) {  // materials.rs:41-50
        if (STATS) st.c[5]++;
        const bool front = ((n.x * -1.0) * d.x + (n.y * -1.0) * d.y + (n.z * -1.0) * d.z) > 0.0;
        Lr = front ? M->fr : 0.0;
        Lg = front ? M->fg : 0.0;
        Lb = front ? M->fb : 0.0;
        return false;
    }
    V3 wi;
    double scale;  // (n . wi) / pdf
    double fr = M->fr, fg = M->fg, fb = M->fb;
    if (kind == kMatMatte) {  // materials.rs:18-34 + Lambertian::sample_f brdf.rs:19-31
        if (STATS) st.c[2]++;
        const size_t N = P.nsamp;
        const double *hp = P.hemi + ((size_t)set * P.max_depth + (p.depth - 1)) * 3 * N + i;
        const double hx = hp[0], hy = hp[N], hz = hp[2 * N];
        const V3 w = n;
        const V3 v = normalize(cross(mk(0.0034, 1.0, 0.0071), w));
        const V3 u = cross(v, w);
        wi = normalize(mk((hx * u.x + hy * v.x) + hz * w.x, (hx * u.y + hy * v.y) + hz * w.y,
                          (hx * u.z + hy * v.z) + hz * w.z));
        const double ndotwi = dot(n, wi);
        const double pdf = ndotwi * kInvPi;
        scale = ndotwi / pdf;
    } else {
        // Reflective::path_shade materials.rs:56-72; wo = -d, so -wo = d exactly
        const V3 wo = mk(d.x * -1.0, d.y * -1.0, d.z * -1.0);
        const double ndotwo = dot(n, wo);
        const V3 r = mk(-wo.x + n.x * ndotwo * 2.0, -wo.y + n.y * ndotwo * 2.0, -wo.z + n.z * ndotwo * 2.0);
        if (kind == kMatReflective) {  // PerfectSpecular::sample_f brdf.rs:38-46
            if (STATS) st.c[4]++;
            wi = r;
            const double pdf = dot(n, wi);
            scale = dot(n, wi) / pdf;
        } else {  // GlossySpecular::sample_f brdf.rs:54-79
            if (STATS) st.c[3]++;
            const V3 w = r;
            const V3 u = normalize(cross(mk(0.00424, 1.0, 0.00764), w));
            const V3 v = cross(u, w);
            const V3 h = to_unit_hemi(p.sq, M->inv_e1);
            const V3 wi0 = mk((u.x * h.x + v.x * h.y) + w.x * h.z, (u.y * h.x + v.y * h.y) + w.y * h.z,
                              (u.z * h.x + v.z * h.y) + w.z * h.z);
            if (dot(n, wi0) < 0.0) {
                wi = mk((u.x * -h.x - v.x * h.y) + w.x * h.z, (u.y * -h.x - v.y * h.y) + w.y * h.z,
                        (u.z * -h.x - v.z * h.y) + w.z * h.z);
            } else {
                wi = wi0;
            }
            const double lobe = pow(dot(r, wi), M->exponent);
            const double pdf = lobe * dot(n, wi);
            fr = fr * lobe;
            fg = fg * lobe;
            fb = fb * lobe;
            scale = dot(n, wi) / pdf;
        }
    }
    // push (f, s); child ray starts at the hit point (materials.rs:26-29,65-68)
    double *e = stk + (size_t)p.nb * 4 * stk_stride;
    e[0] = fr;
    e[stk_stride] = fg;
    e[2 * stk_stride] = fb;
    e[3 * stk_stride] = scale;
    p.nb++;
    p.depth++;
    p.r.ox = pt.x;
    p.r.oy = pt.y;
    p.r.oz = pt.z;
    p.r.dx = wi.x;
    p.r.dy = wi.y;
    p.r.dz = wi.z;
    return true;
}

// Unwind the recursion: L <- (f (*) L) * s from the deepest bounce outward
// (materials.rs:31-33,70-71).
__device__ __forceinline__ void fold_stack(const double *stk, int stk_stride, int nb, double &Lr,
                                           double &Lg, double &Lb) {
    for (int k = nb - 1; k >= 0; --k) {
        const double *e = stk + (size_t)k * 4 * stk_stride;
        const double s = e[3 * stk_stride];
        Lr = (e[0] * Lr) * s;
        Lg = (e[stk_stride] * Lg) * s;
        Lb = (e[2 * stk_stride] * Lb) * s;
    }
}

// Sum the per-lane partials of one pixel in lane order (leader = first lane of
// the pixel's lane group), then trace.rs:85-87: * 1/n^2, max_to_one, store.
__device__ __forceinline__ void finish_pixel(const RenderParams &P, bool lane_on, uint32_t lane,
                                             uint32_t seg_base, uint32_t lpp, uint64_t pixel, double sr,
                                             double sg, double sb) {
    double r = 0.0, g = 0.0, b = 0.0;
    for (uint32_t k = 0; k < lpp; ++k) {
        const int src = (int)((seg_base + k) & 63u);
        r += __shfl(sr, src);
        g += __shfl(sg, src);
        b += __shfl(sb, src);
    }
    if (lane_on && lane == seg_base) {
        r *= P.pixel_denom;
        g *= P.pixel_denom;
        b *= P.pixel_denom;
        // Color::max_to_one: color.rs:35-44
        const double mx1 = r > g ? r : g;
        const double mx2 = mx1 > b ? mx1 : b;
        if (mx2 > 1.0) {
            const double inv = 1.0 / mx2;
            r *= inv;
            g *= inv;
            b *= inv;
        }
        double *o = P.out + pixel * 3;
        o[0] = r;
        o[1] = g;
        o[2] = b;
    }
}

template <bool STATS>
__device__ __forceinline__ void flush_stats(const RenderParams &P, Stats &st, uint32_t lane) {
    if (!STATS) return;
    for (int c = 0; c < 10; ++c) {
        unsigned v = st.c[c];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0 && v) atomicAdd(P.stats + c, (unsigned long long)v);
    }
}

// ---------------------------------------------------------------------------
// STATIC variant: lane l of a pixel's lane group traces samples l, l+64, ...;
// a lane idles from the end of its path until the slowest lane of the chunk is done.
// Handles every n (also n*n < 64: 64/(n*n) pixels share a wave).
// ---------------------------------------------------------------------------
template <bool STATS, bool TRIS>
__global__ __launch_bounds__(FLUX_BLOCK_THREADS, FLUX_WAVES_PER_EU) void render_static_kernel(const RenderParams P) {
    extern __shared__ double lds_stack[];
    const int tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const int stride = blockDim.x;
    double *stk = lds_stack + tid;
    int *tstack = reinterpret_cast<int *>(lds_stack + (size_t)P.max_depth * 4 * stride) + tid;

    const uint32_t N = P.nsamp;
    const uint32_t lpp = N >= 64u ? 64u : N;  // lanes per pixel
    const uint32_t ppw = 64u / lpp;           // pixels per wave
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (tid >> 6);
    const uint32_t slot = lane / lpp;
    const uint32_t seg_base = slot * lpp;
    const uint32_t i0 = lane - seg_base;
    const uint64_t npix = (uint64_t)P.num_rows * P.img_w;
    const uint64_t pixel = wave * ppw + slot;
    const bool lane_on = slot < ppw && pixel < npix;

    int row = 0, col = 0;
    uint32_t set = 0;
    if (lane_on) {
        const uint64_t lrow = pixel / P.img_w;
        col = (int)(pixel - lrow * P.img_w);
        row = P.first_row + (int)lrow * P.row_stride;
        set = (uint32_t)P.rowperm[(size_t)row * P.num_sets + col] % P.num_sets;  // trace.rs:68-69
    }
    Stats st = {};
    double sr = 0.0, sg = 0.0, sb = 0.0;
    const uint32_t nchunks = (N + 63u) / 64u;
    for (uint32_t c = 0; c < nchunks; ++c) {
        const uint32_t i = i0 + c * 64u;
        if (lane_on && i < N) {
            Path p;
            p.sq = P.pix[(size_t)set * N + i];
            const double2 lens = P.disc[(size_t)set * N + i];
            p.r = primary_ray(P, row, col, p.sq, lens);
            p.depth = 1;
            p.nb = 0;
            if (STATS) st.c[0]++;
            double Lr, Lg, Lb;
            while (shade_level<STATS, TRIS>(P, p, set, i, stk, stride, tstack, Lr, Lg, Lb, st)) {
            }
            fold_stack(stk, stride, p.nb, Lr, Lg, Lb);
            sr += Lr;  // trace.rs:82
            sg += Lg;
            sb += Lb;
        }
    }
    finish_pixel(P, lane_on, lane, seg_base, lpp, pixel, sr, sg, sb);
    flush_stats<STATS>(P, st, lane);
}

// ---------------------------------------------------------------------------
// REFILL variant (n*n >= 64): the wave keeps a cursor into its pixel's sample
// list; every iteration all live lanes advance one Scene::shade level, and the
// lanes whose path just ended are compacted (ballot + mbcnt prefix) onto the
// next unstarted samples.
// ---------------------------------------------------------------------------
template <bool STATS, bool TRIS>
__global__ __launch_bounds__(FLUX_BLOCK_THREADS, FLUX_WAVES_PER_EU) void render_refill_kernel(const RenderParams P) {
    extern __shared__ double lds_stack[];
    const int tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const int stride = blockDim.x;
    double *stk = lds_stack + tid;
    int *tstack = reinterpret_cast<int *>(lds_stack + (size_t)P.max_depth * 4 * stride) + tid;

    const uint32_t N = P.nsamp;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (tid >> 6);
    const uint64_t npix = (uint64_t)P.num_rows * P.img_w;
    const uint64_t pixel = wave;
    const bool wave_on = pixel < npix;  // wave-uniform

    Stats st = {};
    double sr = 0.0, sg = 0.0, sb = 0.0;
    if (wave_on) {
        const uint64_t lrow = pixel / P.img_w;
        const int col = (int)(pixel - lrow * P.img_w);
        const int row = P.first_row + (int)lrow * P.row_stride;
        const uint32_t set = (uint32_t)P.rowperm[(size_t)row * P.num_sets + col] % P.num_sets;
        const double2 *pix = P.pix + (size_t)set * N;
        const double2 *disc = P.disc + (size_t)set * N;

        uint32_t next = 0;  // wave-uniform cursor: first unstarted sample
        uint32_t i = 0;
        bool live = false;
        bool want = true;  // lane needs a new sample
        Path p;
        p.depth = 1;
        p.nb = 0;
        p.sq = make_double2(0.0, 0.0);
        p.r = Ray{0, 0, 0, 0, 0, 1};
        for (;;) {
            // --- compaction: hand the next samples to the lanes that are free
            const unsigned long long freemask = __ballot(want);
            if (freemask) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(freemask >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)freemask, 0u));
                if (want) {
                    i = next + rank;
                    live = i < N;
                    if (live) {
                        p.sq = pix[i];
                        p.r = primary_ray(P, row, col, p.sq, disc[i]);
                        p.depth = 1;
                        p.nb = 0;
                        if (STATS) st.c[0]++;
                    }
                    want = false;
                }
                next += (uint32_t)__popcll(freemask);
                if (next > N) next = N;
            }
            if (!__any(live)) break;
            // --- one Scene::shade level for every live lane
            if (live) {
                double Lr, Lg, Lb;
                if (!shade_level<STATS, TRIS>(P, p, set, i, stk, stride, tstack, Lr, Lg, Lb, st)) {
                    fold_stack(stk, stride, p.nb, Lr, Lg, Lb);
                    sr += Lr;
                    sg += Lg;
                    sb += Lb;
                    live = false;
                    want = next < N;  // nothing left to start -> lane retires
                }
            }
        }
    }
    finish_pixel(P, wave_on, lane, 0u, 64u, pixel, sr, sg, sb);
    flush_stats<STATS>(P, st, lane);
}

hipError_t launch_render(const RenderParams &p, int variant, hipStream_t stream) {
    const uint32_t N = p.nsamp;
    const uint64_t npix = (uint64_t)p.num_rows * (uint64_t)p.img_w;
    if (npix == 0) return hipSuccess;
    if (variant == FLUX_KERNEL_DEFAULT) variant = FLUX_KERNEL_REFILL;
    if (N < 64u) variant = FLUX_KERNEL_STATIC;  // nothing to refill from
    const uint32_t lpp = N >= 64u ? 64u : N;
    const uint32_t ppw = 64u / lpp;
    const uint64_t waves = (npix + ppw - 1) / ppw;
    const unsigned block = FLUX_BLOCK_THREADS;
    const unsigned wpb = block / 64;
    const uint64_t blocks = (waves + wpb - 1) / wpb;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    const bool tris = p.n_tris > 0;
    const size_t lds = (size_t)p.max_depth * 4 * block * sizeof(double) +
                       (tris ? (size_t)p.bvh_stack * block * sizeof(int) : 0);
    const bool stats = p.stats != nullptr;
    const dim3 g((unsigned)blocks), b(block);
#define FLUX_LAUNCH(K)                                                     \
    do {                                                                   \
        if (stats && tris) K<true, true><<<g, b, lds, stream>>>(p);        \
        else if (stats) K<true, false><<<g, b, lds, stream>>>(p);          \
        else if (tris) K<false, true><<<g, b, lds, stream>>>(p);           \
        else K<false, false><<<g, b, lds, stream>>>(p);                    \
    } while (0)
    if (variant == FLUX_KERNEL_STATIC)
        FLUX_LAUNCH(render_static_kernel);
    else
        FLUX_LAUNCH(render_refill_kernel);
#undef FLUX_LAUNCH
    return hipGetLastError();
}

}  // namespace flux
