// abi.hip -- implementation of include/flux_abi.h (the drop-in boundary).
//
// flux_ctx_create  = Scene::from_data (scene.rs:128-154) + Camera::new
//                    (trace.rs:26-42) as called from workers.rs:46-54
// flux_render_rows = Camera::render (trace.rs:53-97) as called from workers.rs:60
// There is no CPU fallback: without a usable HIP device every compute entry
// point fails with FLUX_E_DEVICE.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <algorithm>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "../../include/flux_abi.h"
#include "flux_device.h"
#include "flux_tables.h"
#include "flux_math_coeffs.h"  // kExp2Poly -> RenderParams::exp2c

#include "flux_ctx.h"

namespace flux {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

}  // namespace flux

using flux::DeviceGuard;
using flux::fail;

extern "C" {

uint32_t flux_abi_version(void) { return FLUX_ABI_VERSION; }

const char *flux_last_error(void) { return flux::g_last_error.c_str(); }

int flux_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

#ifndef FLUX_BUILD_ID
#define FLUX_BUILD_ID "lib:unknown kernels:unknown"
#endif
const char *flux_build_id(void) { return FLUX_BUILD_ID; }

int flux_device_warmup(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(FLUX_E_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(FLUX_E_INVALID, "device %d out of range [0,%d)", device, ndev);
    DeviceGuard guard(device);
    if (!guard.ok) return fail(FLUX_E_DEVICE, "hipSetDevice(%d) failed", device);
    // the runtime initialises lazily, per subsystem: the device context, the copy engine's staging (first hipMemcpy), the
    // compute queue + this library's code object (first launch), the stream pool (scripts/micro/cold_start.hip times each)
    double xy[2 * 4];
    double *d = nullptr;
    hipStream_t s = nullptr;
    HIP_TRY(hipMalloc((void **)&d, sizeof(xy)));
    hipError_t e = hipMemcpy(d, xy, sizeof(xy), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = flux::generate_sampler_grid(FLUX_SAMPLER_REGULAR, 0, 2, d, nullptr, nullptr);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (s) (void)hipStreamDestroy(s);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(FLUX_E_DEVICE, "flux_device_warmup: %s", hipGetErrorString(e));
    return FLUX_OK;
}

static void free_ctx(flux_ctx *c) {
    if (!c) return;
    DeviceGuard g(c->device);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    (void)hipFree(c->d_shapes);
    (void)hipFree(c->d_mats);
    (void)hipFree(c->d_fscene);
    (void)hipFree(c->d_pix);
    (void)hipFree(c->d_disc);
    (void)hipFree(c->d_hemi);
    (void)hipFree(c->d_gloss);
    (void)hipFree(c->d_setrows);
    (void)hipFree(c->d_rowperm);
    (void)hipFree(c->d_invperm);
    (void)hipFree(c->d_stats);
    (void)hipFree(c->d_tris);
    (void)hipFree(c->d_nodes);
    (void)hipFree(c->d_nodesq);
    (void)hipFree(c->d_nodes4);
    (void)hipFree(c->d_leaves);
    (void)hipFree(c->d_out);
    delete c;
}

void flux_ctx_destroy(flux_ctx *ctx) { free_ctx(ctx); }

static void normalize3(const double in[3], double out[3]) {
    double len = std::sqrt(in[0] * in[0] + in[1] * in[1] + in[2] * in[2]);
    out[0] = in[0] / len;
    out[1] = in[1] / len;
    out[2] = in[2] / len;
}
static void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// material_from_data (scene.rs:87-125) + the per-material constants of brdf.rs:30,45,76 / materials.rs:45
static void fill_material(flux::DevMaterial &dm, const flux_material &m) {
    dm.kind = m.kind;
    dm.exponent = m.exponent;
    dm.inv_e1 = 1.0 / (m.exponent + 1.0);
    // powf(negative, e): +|x|^e for an even integral e, -|x|^e for an odd one, NaN otherwise
    dm.exp_parity = 0;
    if (std::isfinite(m.exponent) && std::floor(m.exponent) == m.exponent)
        dm.exp_parity = (std::fabs(m.exponent) >= 9007199254740992.0 || std::fmod(m.exponent, 2.0) == 0.0) ? 1 : 2;
    double f[3];
    for (int ch = 0; ch < 3; ch++) {
        f[ch] = m.color[ch] * m.k;
        if (m.kind == FLUX_MAT_MATTE) f[ch] = f[ch] * flux::kInvPi;  // brdf.rs:30
    }
    dm.fr = f[0];
    dm.fg = f[1];
    dm.fb = f[2];
}

int flux_ctx_create(const flux_scene_desc *scene, const flux_job_cfg *cfg, uint64_t seed, int device,
                    flux_ctx **out) {
    return flux_ctx_create_sets(scene, cfg, seed, device, 0, 1, out);
}

int flux_ctx_create_sets(const flux_scene_desc *scene, const flux_job_cfg *cfg, uint64_t seed, int device,
                         uint64_t first_set, uint64_t set_stride, flux_ctx **out) {
    if (!scene || !cfg || !out) return fail(FLUX_E_INVALID, "flux_ctx_create: null argument");
    *out = nullptr;
    // where the wall time of this call goes (flux_ctx_create_timing): lap(k) books the time since the previous lap under word k
    double laps[FLUX_CREATE_TIMING_WORDS] = {};
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](int k) {
        const auto now = std::chrono::steady_clock::now();
        laps[k] += std::chrono::duration<double, std::milli>(now - t_last).count();
        t_last = now;
    };
    if (set_stride < 1 || first_set >= set_stride)
        return fail(FLUX_E_INVALID, "sample-set share: need set_stride >= 1 and first_set < set_stride, got %llu / %llu",
                    (unsigned long long)first_set, (unsigned long long)set_stride);
    if (set_stride > 0xffffffffull)  // stored as 32-bit (SetRange); first_set < set_stride is then in range too
        return fail(FLUX_E_INVALID, "sample-set share: set_stride %llu exceeds 2^32 - 1", (unsigned long long)set_stride);
    if (cfg->sample_root < 1 || cfg->sample_root > 4096)
        return fail(FLUX_E_INVALID, "sample_root must be in [1,4096], got %llu",
                    (unsigned long long)cfg->sample_root);
    if (cfg->max_trace_depth < 1 || cfg->max_trace_depth > 256)
        return fail(FLUX_E_INVALID, "max_trace_depth must be in [1,256], got %llu",
                    (unsigned long long)cfg->max_trace_depth);
    // the render kernels address a pixel's Lambertian samples with a 32-bit byte offset from the set's base:
    // max_trace_depth * sample_root^2 * 32 B must stay below 4 GiB (the table of even ONE narrow image row would
    // otherwise be enormous: this only excludes e.g. sample_root 4096 with depth >= 8)
    if ((uint64_t)cfg->max_trace_depth * cfg->sample_root * cfg->sample_root * 32ull >= (1ull << 32))
        return fail(FLUX_E_INVALID, "max_trace_depth * sample_root^2 = %llu is too large (limit 2^27 = 134217728)",
                    (unsigned long long)(cfg->max_trace_depth * cfg->sample_root * cfg->sample_root));
    if (scene->image_width < 1 || scene->image_height < 1 || scene->image_width > 65535 ||
        scene->image_height > (1u << 20))
        return fail(FLUX_E_INVALID, "image size %llux%llu out of range",
                    (unsigned long long)scene->image_width, (unsigned long long)scene->image_height);
    if (scene->num_shapes > 0 && !scene->shapes)
        return fail(FLUX_E_INVALID, "num_shapes > 0 but shapes is null");
    if (scene->num_shapes > 4096)
        return fail(FLUX_E_INVALID, "flat shape list limited to 4096 shapes, got %llu",
                    (unsigned long long)scene->num_shapes);
    for (uint64_t i = 0; i < scene->num_shapes; i++) {
        const flux_shape &s = scene->shapes[i];
        if (s.kind != FLUX_SHAPE_SPHERE && s.kind != FLUX_SHAPE_PLANE)
            return fail(FLUX_E_INVALID, "shape %llu: unknown kind %d", (unsigned long long)i, s.kind);
        if (s.material.kind < FLUX_MAT_MATTE || s.material.kind > FLUX_MAT_GLOSSY)
            return fail(FLUX_E_INVALID, "shape %llu: unknown material kind %d", (unsigned long long)i,
                        s.material.kind);
    }
    if (scene->num_meshes > 0 && !scene->meshes)
        return fail(FLUX_E_INVALID, "num_meshes > 0 but meshes is null");
    uint64_t total_tris = 0;
    for (uint64_t m = 0; m < scene->num_meshes; m++) {
        const flux_mesh &me = scene->meshes[m];
        if (me.material.kind < FLUX_MAT_MATTE || me.material.kind > FLUX_MAT_GLOSSY)
            return fail(FLUX_E_INVALID, "mesh %llu: unknown material kind %d", (unsigned long long)m, me.material.kind);
        if (me.num_triangles && (!me.vertices || !me.indices))
            return fail(FLUX_E_INVALID, "mesh %llu: null vertices/indices", (unsigned long long)m);
        for (uint64_t k = 0; k < 3 * me.num_triangles; k++)
            if (me.indices[k] >= me.num_vertices)
                return fail(FLUX_E_INVALID, "mesh %llu: vertex index %u out of range (%llu vertices)",
                            (unsigned long long)m, me.indices[k], (unsigned long long)me.num_vertices);
        total_tris += me.num_triangles;
    }
    if (total_tris >= (1ull << 27))  // leaf references pack (first << 3 | count) into 31 bits
        return fail(FLUX_E_INVALID, "too many triangles: %llu", (unsigned long long)total_tris);
    lap(FLUX_CREATE_MS_HOST);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(FLUX_E_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(FLUX_E_INVALID, "device %d out of range [0,%d)", device, ndev);

    DeviceGuard guard(device);
    if (!guard.ok) return fail(FLUX_E_DEVICE, "hipSetDevice(%d) failed", device);
    (void)hipFree(nullptr);  // the runtime's lazy per-device initialisation, booked under its own word (zero once the process has used the device)
    lap(FLUX_CREATE_MS_RUNTIME);

    flux_ctx *c = new (std::nothrow) flux_ctx();
    if (!c) return fail(FLUX_E_NOMEM, "host allocation failed");
    c->device = device;
    c->seed = seed;
    c->n = (uint32_t)cfg->sample_root;
    c->N = c->n * c->n;
    c->D = (uint32_t)cfg->max_trace_depth;
    c->W = (uint32_t)scene->image_width;
    c->H = (uint32_t)scene->image_height;
    c->S = c->W;  // workers.rs:50: num_sets = image_width
    // the sets this context holds tables for: all of them, or one rank's share of a set-sharded render
    c->sets.first = (uint32_t)first_set;
    c->sets.stride = (uint32_t)set_stride;
    c->sets.count = first_set < c->S ? (uint32_t)((c->S - first_set + set_stride - 1) / set_stride) : 0u;

    // ---- Scene::from_data: per-shape constants -------------------------------
    const size_t ns = (size_t)scene->num_shapes;
    std::vector<flux::DevShape> shapes(ns ? ns : 1);
    std::vector<flux::DevMaterial> mats(ns + (size_t)scene->num_meshes + 1);
    std::memset(shapes.data(), 0, shapes.size() * sizeof(flux::DevShape));
    std::memset(mats.data(), 0, mats.size() * sizeof(flux::DevMaterial));
    for (size_t i = 0; i < ns; i++) {
        const flux_shape &s = scene->shapes[i];
        flux::DevShape &d = shapes[i];
        d.kind = s.kind;
        d.px = s.p[0];
        d.py = s.p[1];
        d.pz = s.p[2];
        if (s.kind == FLUX_SHAPE_SPHERE) {
            d.radius = s.radius;
            d.rr = s.radius * s.radius;
            d.inv = s.invert ? -1.0 : 1.0;
            d.inv_rad = d.inv / s.radius;
            // Sphere::new: shapes.rs:154-169
            d.c0x = s.p[0] - s.radius;
            d.c0y = s.p[1] - s.radius;
            d.c0z = s.p[2] - s.radius;
            d.c1x = s.p[0] + s.radius;
            d.c1y = s.p[1] + s.radius;
            d.c1z = s.p[2] + s.radius;
        } else {
            d.c0x = s.n[0];
            d.c0y = s.n[1];
            d.c0z = s.n[2];
        }
        fill_material(mats[i], s.material);
    }
    // FAST path layout of the same shapes: scan records (spheres, planes) + hit records in scan order
    std::vector<flux::DevScanSphere> fsph;
    std::vector<flux::DevScanPlane> fpln;
    std::vector<flux::DevHitRec> frec_s, frec_p;
    for (size_t i = 0; i < ns; i++) {
        const flux::DevShape &d = shapes[i];
        const flux::DevMaterial &m = mats[i];
        flux::DevHitRec r;
        std::memset(&r, 0, sizeof(r));
        // (the FAST bounce weight: flux_device.h DevHitRec; the product is the one the kernels formed per bounce, `fr * scale`)
        const double wsc = m.kind == flux::kMatMatte ? 1.0 / flux::kInvPi : 1.0;
        r.fr = m.kind == flux::kMatMatte ? m.fr * wsc : m.fr;
        r.fg = m.kind == flux::kMatMatte ? m.fg * wsc : m.fg;
        r.fb = m.kind == flux::kMatMatte ? m.fb * wsc : m.fb;
        r.inv_e1 = m.inv_e1;
        r.ax = m.kind == flux::kMatMatte ? 0.0034 : 0.00424;  // brdf.rs:22 / brdf.rs:58
        r.az = m.kind == flux::kMatMatte ? 0.0071 : 0.00764;
        r.shape_kind = d.kind; r.mat_kind = m.kind; r.orig_id = (int32_t)i;
        // spheres: |(hit - centre) / radius| = 1 to rounding; planes use the stored normal as is (shapes.rs:135-152)
        r.unit_normal = d.kind == flux::kShapeSphere ||
                        std::fabs((d.c0x * d.c0x + d.c0y * d.c0y + d.c0z * d.c0z) - 1.0) <= 4.0 * 2.220446049250313e-16;
        if (d.kind == flux::kShapeSphere) {
            r.cx = d.px; r.cy = d.py; r.cz = d.pz; r.inv_rad = d.inv_rad;
            fsph.push_back(flux::DevScanSphere{d.px, d.py, d.pz, d.rr});
            frec_s.push_back(r);
        } else {
            r.cx = d.c0x; r.cy = d.c0y; r.cz = d.c0z;
            flux::DevScanPlane pl;
            std::memset(&pl, 0, sizeof(pl));
            pl.px = d.px; pl.py = d.py; pl.pz = d.pz; pl.nx = d.c0x; pl.ny = d.c0y; pl.nz = d.c0z; pl.id = (int32_t)i;
            fpln.push_back(pl);
            frec_p.push_back(r);
        }
    }
    const size_t fs_sph_bytes = (fsph.size() + 1) * sizeof(flux::DevScanSphere);  // +1: the scan reads one record ahead
    const size_t fs_pln_bytes = (fpln.size() + 1) * sizeof(flux::DevScanPlane);
    const size_t fs_rec_bytes = (ns + 1) * sizeof(flux::DevHitRec);
    // f32 records of the conservative candidate filter (flux_device.h DevScanSphere32): valid while every magnitude
    // stays far inside f32's range (squares are formed), else the f64 filter is used
    std::vector<flux::DevScanSphere32> fsph32((fsph.size() + 1) / 2);
    std::memset(fsph32.data(), 0, fsph32.size() * sizeof(flux::DevScanSphere32));
    bool filter32_ok = true;
    for (size_t k = 0; k < fsph.size(); k++) {
        const flux::DevScanSphere &sp = fsph[k];
        const double pp = sp.px * sp.px + sp.py * sp.py + sp.pz * sp.pz;
        if (!(pp < 1e30) || !(sp.rr < 1e30)) filter32_ok = false;
    }
    // `invert` spheres (environments: nearly every ray is inside and hits them) are tested for all lanes together with
    // scalar operands instead of through every lane's candidate list (render_body.inc scan_shapes_fast): up to two,
    // given a filter record that never passes (c = 3e38: "entirely behind the origin" or dq < 0)
    int n_uni = 0, uni_idx[2] = {0, 0};
    if (filter32_ok)
        for (size_t k = 0; k < fsph.size() && n_uni < FLUX_UNI_SPHERES; k++)
            if (frec_s[k].inv_rad < 0.0) uni_idx[n_uni++] = (int)k;
    for (size_t k = 0; k < fsph.size(); k++) {
        const flux::DevScanSphere &sp = fsph[k];
        const double pp = sp.px * sp.px + sp.py * sp.py + sp.pz * sp.pz;
        const double ppr = (pp - sp.rr) - 8e-6 * (pp + sp.rr) - 1e-30;
        float f = (float)ppr;
        if ((double)f > ppr) f = std::nextafterf(f, -INFINITY);  // rounded down: the bias is never reduced
        flux::DevScanSphere32 &d = fsph32[k / 2];
        d.px[k & 1] = -(float)sp.px;  // the NEGATED centre (flux_device.h DevScanSphere32)
        d.py[k & 1] = -(float)sp.py;
        d.pz[k & 1] = -(float)sp.pz;
        d.ppr[k & 1] = f;
        if ((n_uni > 0 && uni_idx[0] == (int)k) || (n_uni > 1 && uni_idx[1] == (int)k)) {
            d.px[k & 1] = d.py[k & 1] = d.pz[k & 1] = 0.0f;
            d.ppr[k & 1] = 3.0e38f;
        }
    }
    const size_t fs_s32_bytes = (fsph32.size() + 4) * sizeof(flux::DevScanSphere32);  // +4 pairs: the filter loads whole groups of 8 spheres
    // STRICT's sphere records in SCAN order (round 5: its scan takes its candidates from the same f32 filter, whose bit k is scan
    // sphere k): the DevShape as it is, with the YAML index -- the tie rule's key -- in pad0
    std::vector<flux::DevShape> sshapes;
    for (size_t i = 0; i < ns; i++)
        if (shapes[i].kind == flux::kShapeSphere) {
            sshapes.push_back(shapes[i]);
            sshapes.back().pad0 = (int32_t)i;
        }
    const size_t fs_ss_off = (fs_sph_bytes + fs_pln_bytes + fs_rec_bytes + fs_s32_bytes + 127) & ~(size_t)127;
    const size_t fs_ss_bytes = (sshapes.size() + 1) * sizeof(flux::DevShape);
    // The split kernel's per-pixel constants of the primary ray (trace.rs:56-57, 93-94) as two tables a wave reads with scalar loads in
    // its ray-generation step: x - half_w for every column, (H - row) - half_h for every row -- the same two IEEE operations the
    // kernels perform, done once here
    const size_t fs_px_off = (fs_ss_off + fs_ss_bytes + 127) & ~(size_t)127;
    std::vector<unsigned char> fscene(fs_px_off + ((size_t)c->W + c->H) * sizeof(double), 0);
    {
        double *pxc = reinterpret_cast<double *>(fscene.data() + fs_px_off);
        const double half_w = (double)c->W * 0.5, half_h = (double)c->H * 0.5;
        for (uint32_t x = 0; x < c->W; x++) pxc[x] = (double)(int32_t)x - half_w;
        for (uint32_t y = 0; y < c->H; y++) pxc[c->W + y] = (double)((int32_t)c->H - (int32_t)y) - half_h;
    }
    if (!sshapes.empty()) std::memcpy(fscene.data() + fs_ss_off, sshapes.data(), sshapes.size() * sizeof(flux::DevShape));
    if (!fsph32.empty())
        std::memcpy(fscene.data() + fs_sph_bytes + fs_pln_bytes + fs_rec_bytes, fsph32.data(), fsph32.size() * sizeof(flux::DevScanSphere32));
    if (!fsph.empty()) std::memcpy(fscene.data(), fsph.data(), fsph.size() * sizeof(flux::DevScanSphere));
    if (!fpln.empty()) std::memcpy(fscene.data() + fs_sph_bytes, fpln.data(), fpln.size() * sizeof(flux::DevScanPlane));
    if (!frec_s.empty()) std::memcpy(fscene.data() + fs_sph_bytes + fs_pln_bytes, frec_s.data(), frec_s.size() * sizeof(flux::DevHitRec));
    if (!frec_p.empty())
        std::memcpy(fscene.data() + fs_sph_bytes + fs_pln_bytes + frec_s.size() * sizeof(flux::DevHitRec), frec_p.data(),
                    frec_p.size() * sizeof(flux::DevHitRec));

    // extension: meshes -> triangle records (hit order: after all shapes) + BVH
    std::vector<flux::DevTri> tris((size_t)total_tris);
    std::vector<flux::DevNode> nodes;
    {
        // one record per triangle: edges, the geometric normal (a square root and three divisions each).  A million of them take
        // tens of milliseconds on one core, so large meshes are dealt to threads by index range (FLUX_BUILD_THREADS, as for the BVH)
        std::vector<size_t> first((size_t)scene->num_meshes + 1, 0);
        for (uint64_t m = 0; m < scene->num_meshes; m++) {
            fill_material(mats[ns + (size_t)m], scene->meshes[m].material);
            first[(size_t)m + 1] = first[(size_t)m] + (size_t)scene->meshes[m].num_triangles;
        }
        auto make = [&](size_t lo, size_t hi) {
            size_t m = 0;
            for (size_t g = lo; g < hi; g++) {
                while (g >= first[m + 1]) m++;
                const flux_mesh &me = scene->meshes[m];
                const size_t k = g - first[m];
                const double *a = me.vertices + 3 * (size_t)me.indices[3 * k];
                const double *b = me.vertices + 3 * (size_t)me.indices[3 * k + 1];
                const double *d = me.vertices + 3 * (size_t)me.indices[3 * k + 2];
                flux::DevTri t;
                std::memset(&t, 0, sizeof(t));
                t.v0x = a[0]; t.v0y = a[1]; t.v0z = a[2];
                t.e1x = b[0] - a[0]; t.e1y = b[1] - a[1]; t.e1z = b[2] - a[2];
                t.e2x = d[0] - a[0]; t.e2y = d[1] - a[1]; t.e2z = d[2] - a[2];
                const double e1[3] = {t.e1x, t.e1y, t.e1z}, e2[3] = {t.e2x, t.e2y, t.e2z};
                double nn[3], nu[3];
                cross3(e1, e2, nn);
                if (nn[0] == 0.0 && nn[1] == 0.0 && nn[2] == 0.0) {
                    // a triangle whose e1 x e2 is exactly zero (repeated or exactly collinear vertices) has no surface:
                    // clearing the edges makes Moeller-Trumbore's det exactly 0, so it is never hit (and never NaN)
                    t.e1x = t.e1y = t.e1z = t.e2x = t.e2y = t.e2z = 0.0;
                    nu[0] = nu[1] = nu[2] = 0.0;
                } else {
                    normalize3(nn, nu);
                }
                t.nx = nu[0]; t.ny = nu[1]; t.nz = nu[2];
                t.id = (int32_t)(ns + g);
                t.mat = (int32_t)(ns + m);
                tris[g] = t;
            }
        };
        unsigned threads = std::thread::hardware_concurrency();
        if (const char *env = std::getenv("FLUX_BUILD_THREADS")) threads = (unsigned)std::max(1, std::atoi(env));
        threads = std::min(std::max(threads, 1u), 16u);
        const size_t n = tris.size();
        if (threads > 1 && n >= 65536) {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < threads; t++) pool.emplace_back(make, n * t / threads, n * (t + 1) / threads);
            for (std::thread &t : pool) t.join();
        } else {
            make(0, n);
        }
    }
    // the traversal addresses node and triangle records by 32-bit byte offsets from their bases (render_body.inc)
    if (tris.size() * sizeof(flux::DevTri) >= (1ull << 32)) {
        int code = fail(FLUX_E_INVALID, "%zu triangles exceed the %llu a context can hold", tris.size(),
                        (unsigned long long)((1ull << 32) / sizeof(flux::DevTri)));
        delete c;
        return code;
    }
    std::vector<flux::DevNodeQ> nodesq;
    // the FAST traversal kernel's layout: 4-wide nodes + leaf records (flux_bvh.h)
    std::vector<flux::DevNode4Q> nodes4;
    std::vector<flux::DevLeafRec> leafrecs;
    flux::build_bvh(tris, nodes, c->bvh);
    if (!flux::quantize_bvh(nodes, nodesq, c->bvh)) {
        int code = fail(FLUX_E_INVALID, "BVH quantisation lost containment (mesh coordinates beyond the 16-bit grid's reach)");
        delete c;
        return code;
    }
    // round 5: nodes and leaf records in ONE arena of 64-B units, a node's children contiguous (flux_bvh.h DevNode4A); an
    // arena beyond the 26-bit unit index comes back empty and the mesh is walked by the binary tree's kernel
    std::vector<flux::DevNode4A> arena;
    flux::build_wide_arena(nodes, nodesq, tris, arena, c->bvh);
    if (nodes4.size() * sizeof(flux::DevNode4Q) >= (1ull << 32) || leafrecs.size() * sizeof(flux::DevLeafRec) >= (1ull << 32) ||
        leafrecs.size() >= (1ull << 28)) {
        int code = fail(FLUX_E_INVALID, "mesh too large for the traversal kernel's 32-bit record offsets");
        delete c;
        return code;
    }
    if (c->bvh.max_depth > (uint64_t)flux::kBvhMaxDepth) {
        int code = fail(FLUX_E_INVALID, "BVH depth %llu exceeds %d (degenerate mesh)",
                        (unsigned long long)c->bvh.max_depth, flux::kBvhMaxDepth);
        delete c;
        return code;
    }

    // ---- CameraBasis::new: scene.rs:28-35 --------------------------------------
    double em[3] = {scene->eye[0] - scene->look_at[0], scene->eye[1] - scene->look_at[1],
                    scene->eye[2] - scene->look_at[2]};
    double upxw[3];
    normalize3(em, c->Wv);
    cross3(scene->up, c->Wv, upxw);
    normalize3(upxw, c->U);
    cross3(c->Wv, c->U, c->V);

    flux::RenderParams &rp = c->rp;
    rp.ex = scene->eye[0];
    rp.ey = scene->eye[1];
    rp.ez = scene->eye[2];
    rp.Ux = c->U[0];
    rp.Uy = c->U[1];
    rp.Uz = c->U[2];
    rp.Vx = c->V[0];
    rp.Vy = c->V[1];
    rp.Vz = c->V[2];
    rp.Wx = c->Wv[0];
    rp.Wy = c->Wv[1];
    rp.Wz = c->Wv[2];
    rp.aps = scene->pixel_size / scene->zoom_factor;                  // trace.rs:60
    rp.half_w = (double)c->W * 0.5;                                   // trace.rs:57
    rp.half_h = (double)c->H * 0.5;                                   // trace.rs:56
    rp.factor = scene->focal_distance / scene->view_plane_distance;   // trace.rs:45
    rp.focal = scene->focal_distance;
    rp.lens_radius = scene->lens_radius;
    rp.bgr = scene->background[0];
    rp.bgg = scene->background[1];
    rp.bgb = scene->background[2];
    rp.pixel_denom = 1.0 / (double)((uint64_t)c->n * c->n);           // trace.rs:59
    rp.img_w = (int32_t)c->W;
    rp.img_h = (int32_t)c->H;
    rp.n_shapes = (int32_t)ns;
    rp.max_depth = (int32_t)c->D;
    rp.nsamp = c->N;
    rp.num_sets = c->S;

    lap(FLUX_CREATE_MS_HOST);
    // ---- HBM allocations ------------------------------------------------------
    const size_t own = c->sets.count ? c->sets.count : 1;  // a share past the last set holds nothing (allocate one slot)
    const size_t pix_bytes = own * c->N * sizeof(double2);
    const size_t hemi_bytes = own * c->D * c->N * flux::kHemiDoubles * sizeof(double);
    const size_t perm_bytes = (size_t)c->H * c->S * sizeof(int32_t);
    hipError_t e = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) {
        if (e != hipSuccess) return;
        lap(FLUX_CREATE_MS_UPLOAD);
        e = hipMalloc(p, bytes);
        if (e == hipSuccess) c->device_bytes += bytes;
        lap(FLUX_CREATE_MS_ALLOC);
    };
    alloc((void **)&c->d_shapes, shapes.size() * sizeof(flux::DevShape));
    // the materials, followed by their bounce weights {f * (n.wi)/pdf in FAST's closed form: f / INV_PI for Matte, f otherwise; pad}
    // of 32 B each (render_bvh4_kernel keeps a path's material indices and multiplies the weights when the path ends)
    std::vector<double> wtab(mats.size() * 4, 0.0);
    for (size_t k = 0; k < mats.size(); k++) {
        const double sc = mats[k].kind == flux::kMatMatte ? 1.0 / flux::kInvPi : 1.0;
        wtab[4 * k] = mats[k].fr * sc;
        wtab[4 * k + 1] = mats[k].fg * sc;
        wtab[4 * k + 2] = mats[k].fb * sc;
    }
    alloc((void **)&c->d_mats, mats.size() * sizeof(flux::DevMaterial) + wtab.size() * sizeof(double));
    alloc((void **)&c->d_fscene, fscene.size());
    alloc((void **)&c->d_pix, pix_bytes);
    alloc((void **)&c->d_disc, pix_bytes);
    alloc((void **)&c->d_hemi, hemi_bytes);
    alloc((void **)&c->d_gloss, pix_bytes * 2);
    alloc((void **)&c->d_setrows, own * sizeof(flux::DevSetRows));
    alloc((void **)&c->d_rowperm, perm_bytes);
    alloc((void **)&c->d_invperm, perm_bytes);
    alloc((void **)&c->d_stats, FLUX_NUM_STATS * sizeof(unsigned long long));
    if (!tris.empty()) {
        alloc((void **)&c->d_tris, tris.size() * sizeof(flux::DevTri));
        alloc((void **)&c->d_nodes, nodes.size() * sizeof(flux::DevNode));
        if (e == hipSuccess) e = hipMemcpy(c->d_tris, tris.data(), tris.size() * sizeof(flux::DevTri), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(c->d_nodes, nodes.data(), nodes.size() * sizeof(flux::DevNode), hipMemcpyHostToDevice);
        alloc((void **)&c->d_nodesq, nodesq.size() * sizeof(flux::DevNodeQ));
        if (e == hipSuccess) e = hipMemcpy(c->d_nodesq, nodesq.data(), nodesq.size() * sizeof(flux::DevNodeQ), hipMemcpyHostToDevice);
        if (!arena.empty()) {
            alloc((void **)&c->d_nodes4, arena.size() * sizeof(flux::DevNode4A));
            if (e == hipSuccess) e = hipMemcpy(c->d_nodes4, arena.data(), arena.size() * sizeof(flux::DevNode4A), hipMemcpyHostToDevice);
        }
    }
    if (e == hipSuccess) e = hipMemcpy(c->d_shapes, shapes.data(), shapes.size() * sizeof(flux::DevShape), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->d_mats, mats.data(), mats.size() * sizeof(flux::DevMaterial), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->d_mats + mats.size(), wtab.data(), wtab.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->d_fscene, fscene.data(), fscene.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(c->d_stats, 0, FLUX_NUM_STATS * sizeof(unsigned long long));
    lap(FLUX_CREATE_MS_UPLOAD);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    lap(FLUX_CREATE_MS_OTHER);
    // ---- MasterSampleSets::new on the device (sampling.rs:13-33) --------------
    double tab_ms[3] = {0, 0, 0};
    if (e == hipSuccess)
        e = flux::generate_tables(seed, c->S, c->sets, c->D, c->n, c->H, c->d_pix, c->d_disc, c->d_hemi, c->d_rowperm, c->d_invperm, nullptr, tab_ms);
    if (e == hipSuccess) {
        e = flux::generate_gloss_table(c->d_pix, (size_t)c->sets.count * c->N, c->d_gloss, nullptr);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    }
    lap(FLUX_CREATE_MS_TABLES);
    // the generator's scratch (the permutations of every grid: 262 MB at 16384 spp) is allocation, not table work
    laps[FLUX_CREATE_MS_TABLES] -= tab_ms[0] + tab_ms[2];
    laps[FLUX_CREATE_MS_ALLOC] += tab_ms[0];
    laps[FLUX_CREATE_MS_FREE] += tab_ms[2];
    if (e == hipSuccess) {
        // where each held set's rows of the sample tables start: one 32-B record per slot, so that a kernel forms a table address with one
        // scalar load instead of a 64-bit multiply-add chain per table and pass (render_body.inc FLUX_SET_ROWS)
        std::vector<flux::DevSetRows> rows(own);
        for (size_t m = 0; m < own; m++) {
            rows[m].pix = c->d_pix + m * c->N;
            rows[m].disc = c->d_disc + m * c->N;
            rows[m].hemi = c->d_hemi + m * c->D * c->N * flux::kHemiDoubles;
            rows[m].gloss = c->d_gloss ? c->d_gloss + m * c->N * 4 : nullptr;
        }
        e = hipMemcpy(c->d_setrows, rows.data(), own * sizeof(flux::DevSetRows), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        int code = fail(e == hipErrorOutOfMemory ? FLUX_E_NOMEM : FLUX_E_DEVICE, "flux_ctx_create: %s",
                        hipGetErrorString(e));
        free_ctx(c);
        return code;
    }
    rp.shapes = c->d_shapes;
    rp.mats = c->d_mats;
    rp.pix = c->d_pix;
    rp.disc = c->d_disc;
    rp.hemi = c->d_hemi;
    rp.gloss = c->d_gloss;
    rp.set_rows = c->d_setrows;
    for (int k = 0; k < 12; k++) rp.exp2c[k] = flux::fastmath::kExp2Poly[k];
    rp.rowperm = c->d_rowperm;
    rp.invperm = c->d_invperm;
    rp.stats = nullptr;
    rp.tris = c->d_tris;
    rp.nodes = c->d_nodes;
    rp.n_tris = (int32_t)tris.size();
    rp.bvh_stack = (int32_t)c->bvh.max_depth;
    rp.nodesq = c->d_nodesq;
    rp.nodes4 = c->d_nodes4;
    rp.leaves = c->d_leaves;
    rp.bvh4_stack = (int32_t)c->bvh.wide_stack;
    rp.n_mats = (int32_t)mats.size();
    rp.mat_bits = 1;
    while ((size_t)1 << rp.mat_bits < mats.size()) rp.mat_bits++;
    for (int a = 0; a < 3; a++) {
        rp.bvh_qmin[a] = c->bvh.qmin[a];
        rp.bvh_qstep[a] = c->bvh.qstep[a];
    }
    rp.fsph = reinterpret_cast<const flux::DevScanSphere *>(c->d_fscene);
    rp.fpln = reinterpret_cast<const flux::DevScanPlane *>(c->d_fscene + fs_sph_bytes);
    rp.frec = reinterpret_cast<const flux::DevHitRec *>(c->d_fscene + fs_sph_bytes + fs_pln_bytes);
    rp.fsph32 = filter32_ok
                    ? reinterpret_cast<const flux::DevScanSphere32 *>(c->d_fscene + fs_sph_bytes + fs_pln_bytes + fs_rec_bytes)
                    : nullptr;
    rp.sshapes = reinterpret_cast<const flux::DevShape *>(c->d_fscene + fs_ss_off);
    rp.pxc = reinterpret_cast<const double *>(c->d_fscene + fs_px_off);
    rp.fwx = rp.focal * rp.Wx;  // trace.rs:96-98's focal_distance * w, one product per frame instead of per wave
    rp.fwy = rp.focal * rp.Wy;
    rp.fwz = rp.focal * rp.Wz;
    rp.bvh_mag = c->bvh.mag;
    rp.set_first = 0;
    rp.set_stride = 1;
    rp.set_count = (int32_t)c->S;
    rp.out_by_set = 0;
    rp.slot_first = 0;
    rp.slot_stride = 1;
    rp.glossy_long = 0;
    for (const flux::DevHitRec &hr : frec_p)
        if (!hr.unit_normal) rp.glossy_long = 1;
    rp.n_uni = n_uni;
    rp.uni_idx[0] = uni_idx[0];
    rp.uni_idx[1] = uni_idx[1];
    rp.unit_dirs = rp.glossy_long ? 0 : 1;
    rp.self_skip = rp.glossy_long ? 0 : 1;
    {   // the environment shortcut (flux_device.h env_short): exactly one `invert` sphere, Emissive, of ordinary size
        int inverted = 0;
        for (const flux::DevHitRec &hr : frec_s)
            if (hr.inv_rad < 0.0) inverted++;
        rp.env_short = 0;
        rp.pad_env = 0;
        rp.env_radius = 0.0;
        if (n_uni == 1 && inverted == 1 && frec_s[uni_idx[0]].mat_kind == flux::kMatEmissive) {
            const double rad = std::sqrt(fsph[uni_idx[0]].rr);
            if (rad > 1e-3 && rad < 1e6) {
                rp.env_short = 1;
                rp.env_radius = rad * (1.0 + 1e-12);  // never below the true radius: it bounds the exit distance from above
            }
        }
    }
    rp.t_min = flux::kTMin;
    rp.env_deep = -(4.0 * flux::kTMin) * rp.env_radius;  // (the kernels' own expression, evaluated once)
    rp.env_px = rp.env_py = rp.env_pz = rp.env_rr = 0.0;
    rp.env_eps = 1e-9;
    if (n_uni == 1) {
        const flux::DevScanSphere &es = fsph[uni_idx[0]];
        rp.env_px = es.px; rp.env_py = es.py; rp.env_pz = es.pz; rp.env_rr = es.rr;
    }
    {   // the filter's group walk for at most 32 spheres (render_body.inc sphere_filter32: the same arithmetic, done once)
        rp.f32_half = nullptr;
        rp.f32_top = rp.fsph32;
        rp.f32_groups = 0;
        rp.f32_valid = fsph.size() >= 32 ? 0xffffffffu : (1u << fsph.size()) - 1u;
        if (rp.fsph32 != nullptr && fsph.size() <= 32) {
            int pairs = ((int)fsph.size() + 1) >> 1;
            const int rem = pairs & 3;
            if (rem == 1 || rem == 2) {
                rp.f32_half = rp.fsph32 + (pairs - rem);
                pairs -= rem;
            } else if (rem == 3) {
                pairs += 1;  // its fourth pair is padding (zeros)
            }
            rp.f32_groups = pairs / 4;
            rp.f32_top = rp.fsph32 + pairs;
        }
    }
    for (const flux::DevScanSphere &sp : fsph)
        if (!(std::fabs(sp.px) < 1e3 && std::fabs(sp.py) < 1e3 && std::fabs(sp.pz) < 1e3 && sp.rr < 1e6)) rp.self_skip = 0;
    rp.n_sph = (int32_t)fsph.size();
    rp.n_pln = (int32_t)fpln.size();
    lap(FLUX_CREATE_MS_UPLOAD);
    laps[FLUX_CREATE_MS_TOTAL] = 0.0;
    for (int k = 1; k < FLUX_CREATE_TIMING_WORDS; k++) laps[FLUX_CREATE_MS_TOTAL] += laps[k];  // (the parts sum to the total by construction)
    for (int k = 0; k < FLUX_CREATE_TIMING_WORDS; k++) c->create_ms[k] = laps[k];
    *out = c;
    return FLUX_OK;
}

static bool holds_all_sets(const flux_ctx *c) { return c->sets.stride == 1 && c->sets.first == 0; }

// The arithmetic a render really runs with.  FLUX_MATH_FAST is defined for scenes whose surface normals are unit vectors (every
// sphere; a plane stored with a unit normal): there its closed-form bounce weights and its front-to-back throughput product
// are the reference's values to rounding.  A plane stored with a NON-unit normal (RenderParams::glossy_long) makes reflected
// directions non-unit, Phong lobes under- / overflow, and the reference's recursion L = (f (*) L) * s (materials.rs:31-33,
// 69-71) then meets its zeros and infinities in an order no reordered product reproduces (a 240 000-scene soak: three
// pixels' channels finite in FAST where the reference holds NaN).  Such a scene is rendered with the
// STRICT arithmetic, whatever flux_ctx_set_math says: the reference's own operation order, identical in every soak scene.
// ... unless STRICT cannot run the job at all: it keeps 32 B of recursion stack per level and lane in LDS (one wave per block
// holds 31 levels beside nothing else), FAST keeps none.  A FAST job deeper than that, on such a scene, stays with FAST and its
// long-form glossy weights (RenderParams::glossy_long; what every FAST render of such a scene was before round 4: the
// reference's NaN pixels in all but the rarest orderings of an overflow and a zero) instead of being refused.
// (a function of the JOB alone -- max_trace_depth and the mesh's BVH depth --, not of flux_ctx_set_traversal: the arithmetic a scene
// is rendered with must not flip with a test hook; a brute-force traversal leaves the BVH stack's bytes unused)
static bool strict_fits_lds(const flux_ctx *ctx) {
    const size_t stack = (size_t)ctx->D * 4 * 64 * sizeof(double);
    const size_t bvh = ctx->rp.n_tris > 0 ? (size_t)ctx->bvh.max_depth * 64 * sizeof(int) : 0;
    return stack + bvh + 512 <= 64 * 1024;  // (the smallest block the planner can choose: one wave; plan_render_impl)
}
static int effective_math(const flux_ctx *ctx) {
    return (ctx->math == FLUX_MATH_FAST && ctx->rp.glossy_long && strict_fits_lds(ctx)) ? FLUX_MATH_STRICT : ctx->math;
}

// LDS one block of the kernel about to be launched needs: asked of the launch plan itself (render_body.inc plan_render_impl:
// STRICT keeps the (f,s) recursion stack, 4 doubles per level per lane; mesh scenes the BVH traversal stack, one int per
// level per lane -- 64 lanes in the FAST state-machine kernel, the block's in the others --; the split kernel its path
// queues), plus the few static words of the refill / split kernels.  64 KiB per block is the launch limit.
static int check_lds_budget(const flux_ctx *ctx, const flux::RenderParams &p, const char *what) {
    const size_t dyn = flux::plan_render(p, ctx->variant, effective_math(ctx)).lds;
    const size_t fixed = 512;
    if (dyn + fixed > 64 * 1024)
        return fail(FLUX_E_INVALID, "%s: %zu B of LDS per block (max_trace_depth %u%s, BVH depth %llu) exceed the 64 KiB limit; use %s", what,
                    dyn + fixed, ctx->D,
                    effective_math(ctx) == FLUX_MATH_STRICT ? " in the STRICT arithmetic: 32 B of recursion stack per level and lane" : "",
                    (unsigned long long)(p.bvh_stack > 0 ? ctx->bvh.max_depth : 0),
                    // (a FAST context only gets here in FAST: the routing to STRICT is dropped where STRICT does not fit)
                    ctx->math == FLUX_MATH_STRICT ? "FLUX_MATH_FAST or a smaller max_trace_depth" : "a smaller max_trace_depth");
    return FLUX_OK;
}
// flux_debug_shade: 64-thread blocks, STRICT recursion stack + per-lane BVH stack (render_body.inc launch_shade_rays_impl)
static int check_lds_budget_rays(const flux_ctx *ctx, const flux::RenderParams &p) {
    const size_t strict = effective_math(ctx) == FLUX_MATH_STRICT ? (size_t)ctx->D * 4 * 64 * sizeof(double) : 0;
    const size_t stack = p.n_tris > 0 ? (size_t)p.bvh_stack * 64 * sizeof(int) : 0;
    if (strict + stack > 64 * 1024)
        return fail(FLUX_E_INVALID, "flux_debug_shade: %zu B of LDS per block exceed the 64 KiB limit", strict + stack);
    return FLUX_OK;
}

int flux_ctx_set_kernel(flux_ctx *ctx, int variant) {
    if (!ctx) return fail(FLUX_E_INVALID, "null context");
    if (variant < FLUX_KERNEL_DEFAULT || variant > FLUX_KERNEL_SPLIT)
        return fail(FLUX_E_INVALID, "unknown kernel variant %d", variant);
    ctx->variant = variant;
    return FLUX_OK;
}

int flux_ctx_set_math(flux_ctx *ctx, int mode) {
    if (!ctx) return fail(FLUX_E_INVALID, "null context");
    if (mode != FLUX_MATH_FAST && mode != FLUX_MATH_STRICT) return fail(FLUX_E_INVALID, "unknown math mode %d", mode);
    ctx->math = mode;
    return FLUX_OK;
}

int flux_debug_shade(flux_ctx *ctx, uint64_t n, const double *rays, uint64_t depth, uint64_t set_index,
                     uint64_t sample_index, double *out_rgb, int32_t *out_hit, double *out_t) {
    if (!ctx || !rays || !out_rgb) return fail(FLUX_E_INVALID, "null argument");
    if (n == 0) return FLUX_OK;
    if (n > (1u << 24)) return fail(FLUX_E_INVALID, "too many rays");
    if (!holds_all_sets(ctx)) return fail(FLUX_E_INVALID, "this context holds a share of the sample sets only (flux_ctx_create_sets)");
    if (depth < 1 || set_index >= ctx->S || sample_index >= ctx->N)
        return fail(FLUX_E_INVALID, "depth >= 1, set_index < %u and sample_index < %u required", ctx->S, ctx->N);
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(FLUX_E_DEVICE, "hipSetDevice(%d) failed", ctx->device);
    double *d_rays = nullptr, *d_rgb = nullptr, *d_t = nullptr;
    int *d_hit = nullptr;
    hipError_t e = hipMalloc((void **)&d_rays, (size_t)n * 6 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&d_rgb, (size_t)n * 3 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&d_t, (size_t)n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&d_hit, (size_t)n * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice);
    flux::RenderParams p = ctx->rp;
    if (ctx->traversal == FLUX_TRAVERSE_BRUTE) p.bvh_stack = 0;
    if (ctx->traversal == FLUX_TRAVERSE_BVH_BINARY) p.nodes4 = nullptr;
    p.glossy_long = 1;  // caller-supplied directions need not be unit vectors
    p.unit_dirs = 0;
    p.self_skip = 0;
    p.env_short = 0;
    if (int rc = check_lds_budget_rays(ctx, p)) {
        (void)hipFree(d_rays);
        (void)hipFree(d_rgb);
        (void)hipFree(d_t);
        (void)hipFree(d_hit);
        return rc;
    }
    if (e == hipSuccess)
        e = flux::launch_shade_rays(p, effective_math(ctx), d_rays, (int)n, (int)depth, (uint32_t)set_index, (uint32_t)sample_index,
                                    d_rgb, d_hit, d_t, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_rgb, d_rgb, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && out_hit) e = hipMemcpy(out_hit, d_hit, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess && out_t) e = hipMemcpy(out_t, d_t, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_rays);
    (void)hipFree(d_rgb);
    (void)hipFree(d_t);
    (void)hipFree(d_hit);
    if (e != hipSuccess) return fail(FLUX_E_DEVICE, "debug shade: %s", hipGetErrorString(e));
    return FLUX_OK;
}

int flux_sampler_grid(int device, int kind, uint64_t sample_root, uint64_t seed, double *out_xy, double *out_hemi) {
    if (!out_xy) return fail(FLUX_E_INVALID, "null output");
    if (kind < FLUX_SAMPLER_REGULAR || kind > FLUX_SAMPLER_CORRELATED_MULTI_JITTERED)
        return fail(FLUX_E_INVALID, "unknown sampler kind %d", kind);
    if (sample_root < 1 || sample_root > 4096) return fail(FLUX_E_INVALID, "sample_root must be in [1, 4096]");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
        return fail(FLUX_E_DEVICE, "no HIP device %d (this library has no CPU fallback)", device);
    DeviceGuard guard(device);
    const size_t N = (size_t)sample_root * sample_root;
    double *dxy = nullptr, *dh = nullptr;
    hipError_t e = hipMalloc((void **)&dxy, N * 2 * sizeof(double));
    if (e == hipSuccess && out_hemi) e = hipMalloc((void **)&dh, N * 3 * sizeof(double));
    if (e == hipSuccess) e = flux::generate_sampler_grid(kind, seed, (uint32_t)sample_root, dxy, dh, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_xy, dxy, N * 2 * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && out_hemi) e = hipMemcpy(out_hemi, dh, N * 3 * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(dxy);
    (void)hipFree(dh);
    if (e != hipSuccess) return fail(FLUX_E_DEVICE, "sampler grid: %s", hipGetErrorString(e));
    return FLUX_OK;
}

int flux_debug_fastmath(int device, int fn, const double *a, const double *b, double *out, uint64_t n) {
    if (!a || !out) return fail(FLUX_E_INVALID, "null argument");
    if (fn < 0 || fn > 9) return fail(FLUX_E_INVALID, "unknown function %d", fn);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
        return fail(FLUX_E_DEVICE, "no HIP device %d (this library has no CPU fallback)", device);
    DeviceGuard guard(device);
    if (n == 0) return FLUX_OK;
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    const size_t bytes = (size_t)n * sizeof(double);
    hipError_t e = hipMalloc((void **)&da, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&dout, bytes);
    if (e == hipSuccess && b) e = hipMalloc((void **)&db, bytes);
    if (e == hipSuccess) e = hipMemcpy(da, a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && b) e = hipMemcpy(db, b, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = flux::launch_fastmath_probe(fn, da, db, dout, (size_t)n, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(da);
    (void)hipFree(db);
    (void)hipFree(dout);
    if (e != hipSuccess) return fail(FLUX_E_DEVICE, "fastmath probe: %s", hipGetErrorString(e));
    return FLUX_OK;
}

// The work fields of a launch: ONE place per entry point, shared by the render call and flux_ctx_launch_plan.
static flux::RenderParams apply_traversal(const flux_ctx *ctx, flux::RenderParams p) {
    p.stats = ctx->stats_on ? ctx->d_stats : nullptr;
    if (ctx->traversal == FLUX_TRAVERSE_BRUTE) p.bvh_stack = 0;
    if (ctx->traversal == FLUX_TRAVERSE_BVH_BINARY) p.nodes4 = nullptr;
    return p;
}
static flux::RenderParams rows_params(const flux_ctx *ctx, uint64_t first_row, uint64_t row_stride, uint64_t num_rows, double *out) {
    flux::RenderParams p = ctx->rp;
    p.out = out;
    p.first_row = (int32_t)first_row;
    p.row_stride = (int32_t)row_stride;
    p.num_rows = (int32_t)num_rows;
    return apply_traversal(ctx, p);
}
static flux::RenderParams sets_params(const flux_ctx *ctx, uint64_t first_set, uint64_t set_stride, uint64_t num_sets, double *out) {
    flux::RenderParams p = ctx->rp;
    p.out = out;
    p.first_row = 0;
    p.row_stride = 1;
    p.num_rows = (int32_t)ctx->H;
    p.set_first = (int32_t)first_set;
    p.set_stride = (int32_t)set_stride;
    p.set_count = (int32_t)num_sets;
    p.out_by_set = 1;
    p.slot_first = (int32_t)((first_set - ctx->sets.first) / ctx->sets.stride);
    p.slot_stride = (int32_t)(set_stride / ctx->sets.stride);
    return apply_traversal(ctx, p);
}

int flux_ctx_launch_plan(flux_ctx *ctx, uint64_t num_rows, uint64_t num_sets, int64_t out[FLUX_PLAN_WORDS]) {
    if (!ctx || !out) return fail(FLUX_E_INVALID, "null argument");
    if (num_rows > ctx->H) return fail(FLUX_E_INVALID, "%llu rows exceed image height %u", (unsigned long long)num_rows, ctx->H);
    flux::RenderParams p;
    if (num_sets == 0) {
        p = rows_params(ctx, 0, 1, num_rows, nullptr);
    } else {
        const uint64_t held = ctx->sets.count;
        if (num_sets > held) return fail(FLUX_E_INVALID, "%llu sets exceed the %llu this context holds", (unsigned long long)num_sets,
                                         (unsigned long long)held);
        p = sets_params(ctx, ctx->sets.first, ctx->sets.stride, num_sets, nullptr);
    }
    const flux::LaunchPlan L = flux::plan_render(p, ctx->variant, effective_math(ctx));
    out[0] = L.kernel;
    out[1] = L.block;
    out[2] = (int64_t)L.blocks;
    out[3] = (int64_t)L.lds;
    out[4] = L.waves_per_pixel;
    out[5] = effective_math(ctx);
    // why the arithmetic is not the requested one, or why it SHOULD not be: a FAST job on a scene with a non-unit plane normal
    out[6] = (ctx->math == FLUX_MATH_FAST && ctx->rp.glossy_long) ? (strict_fits_lds(ctx) ? FLUX_ROUTE_TO_STRICT : FLUX_ROUTE_KEPT_FAST)
                                                                 : FLUX_ROUTE_NONE;
    out[7] = 0;
    return FLUX_OK;
}

int flux_render_rows_device(flux_ctx *ctx, uint64_t first_row, uint64_t row_stride, uint64_t num_rows,
                            void *d_out_rgb, void *hip_stream) {
    if (!ctx) return fail(FLUX_E_INVALID, "null context");
    if (num_rows == 0) return FLUX_OK;
    if (!d_out_rgb) return fail(FLUX_E_INVALID, "null output pointer");
    if (row_stride < 1) return fail(FLUX_E_INVALID, "row_stride must be >= 1");
    if (!holds_all_sets(ctx))  // every row uses every sample set (trace.rs:64-69)
        return fail(FLUX_E_INVALID, "rendering rows needs all sample sets; this context holds the share %u + k*%u "
                    "(flux_ctx_create_sets): use flux_render_sets_device", ctx->sets.first, ctx->sets.stride);
    if (first_row >= ctx->H || first_row + (num_rows - 1) * row_stride >= ctx->H)
        return fail(FLUX_E_INVALID, "rows %llu + k*%llu (k<%llu) exceed image height %u",
                    (unsigned long long)first_row, (unsigned long long)row_stride,
                    (unsigned long long)num_rows, ctx->H);
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(FLUX_E_DEVICE, "hipSetDevice(%d) failed", ctx->device);
    hipStream_t stream = (hipStream_t)hip_stream;
    const flux::RenderParams p = rows_params(ctx, first_row, row_stride, num_rows, (double *)d_out_rgb);
    if (int rc = check_lds_budget(ctx, p, "flux_render_rows")) return rc;
    HIP_TRY(hipEventRecord(ctx->ev0, stream));
    HIP_TRY(flux::launch_render(p, ctx->variant, effective_math(ctx), stream));
    HIP_TRY(hipEventRecord(ctx->ev1, stream));
    ctx->timed = true;
    return FLUX_OK;
}

int flux_render_sets_device(flux_ctx *ctx, uint64_t first_set, uint64_t set_stride, uint64_t num_sets, void *d_out_rgb,
                            void *hip_stream) {
    if (!ctx) return fail(FLUX_E_INVALID, "null context");
    if (num_sets == 0) return FLUX_OK;
    if (!d_out_rgb) return fail(FLUX_E_INVALID, "null output pointer");
    if (set_stride < 1) return fail(FLUX_E_INVALID, "set_stride must be >= 1");
    if (first_set >= ctx->S || first_set + (num_sets - 1) * set_stride >= ctx->S)
        return fail(FLUX_E_INVALID, "sets %llu + k*%llu (k<%llu) exceed the %u sample sets", (unsigned long long)first_set,
                    (unsigned long long)set_stride, (unsigned long long)num_sets, ctx->S);
    if (ctx->N < 64 || ctx->variant == FLUX_KERNEL_STATIC)
        return fail(FLUX_E_INVALID, "set-sharded rendering needs the refill kernel (sample_root^2 >= 64)");
    // every requested set must have its tables here: first_set + m*set_stride = sets.first + (slot_first + m*slot_stride)*sets.stride
    if (first_set < ctx->sets.first || (first_set - ctx->sets.first) % ctx->sets.stride != 0 ||
        (num_sets > 1 && set_stride % ctx->sets.stride != 0))
        return fail(FLUX_E_INVALID, "sets %llu + k*%llu are not all among this context's share %u + k*%u (flux_ctx_create_sets)",
                    (unsigned long long)first_set, (unsigned long long)set_stride, ctx->sets.first, ctx->sets.stride);
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(FLUX_E_DEVICE, "hipSetDevice(%d) failed", ctx->device);
    hipStream_t stream = (hipStream_t)hip_stream;
    const flux::RenderParams p = sets_params(ctx, first_set, set_stride, num_sets, (double *)d_out_rgb);
    if (int rc = check_lds_budget(ctx, p, "flux_render_sets_device")) return rc;
    HIP_TRY(hipEventRecord(ctx->ev0, stream));
    HIP_TRY(flux::launch_render(p, ctx->variant, effective_math(ctx), stream));
    HIP_TRY(hipEventRecord(ctx->ev1, stream));
    ctx->timed = true;
    return FLUX_OK;
}

int flux_render_rows(flux_ctx *ctx, uint64_t row_start, uint64_t row_end, double *out_rgb) {
    if (!ctx) return fail(FLUX_E_INVALID, "null context");
    if (!out_rgb) return fail(FLUX_E_INVALID, "null output pointer");
    if (row_end < row_start || row_end >= ctx->H)
        return fail(FLUX_E_INVALID, "work unit rows [%llu,%llu] outside image height %u",
                    (unsigned long long)row_start, (unsigned long long)row_end, ctx->H);
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(FLUX_E_DEVICE, "hipSetDevice(%d) failed", ctx->device);
    const uint64_t rows = row_end - row_start + 1;
    const size_t doubles = (size_t)rows * ctx->W * 3;
    if (doubles > ctx->d_out_doubles) {
        (void)hipFree(ctx->d_out);
        ctx->d_out = nullptr;
        ctx->d_out_doubles = 0;
        HIP_TRY(hipMalloc((void **)&ctx->d_out, doubles * sizeof(double)));
        ctx->d_out_doubles = doubles;
    }
    int rc = flux_render_rows_device(ctx, row_start, 1, rows, ctx->d_out, nullptr);
    if (rc != FLUX_OK) return rc;
    HIP_TRY(hipMemcpy(out_rgb, ctx->d_out, doubles * sizeof(double), hipMemcpyDeviceToHost));
    return FLUX_OK;
}

double flux_ctx_last_kernel_ms(flux_ctx *ctx) {
    if (!ctx || !ctx->timed) return -1.0;
    DeviceGuard guard(ctx->device);
    if (hipEventSynchronize(ctx->ev1) != hipSuccess) return -1.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) != hipSuccess) return -1.0;
    return (double)ms;
}

int flux_ctx_enable_stats(flux_ctx *ctx, int on) {
    if (!ctx) return fail(FLUX_E_INVALID, "null context");
    ctx->stats_on = on != 0;
    return FLUX_OK;
}

int flux_ctx_set_traversal(flux_ctx *ctx, int mode) {
    if (!ctx) return fail(FLUX_E_INVALID, "null context");
    if (mode != FLUX_TRAVERSE_BVH && mode != FLUX_TRAVERSE_BRUTE && mode != FLUX_TRAVERSE_BVH_BINARY)
        return fail(FLUX_E_INVALID, "unknown traversal mode %d", mode);
    ctx->traversal = mode;
    return FLUX_OK;
}

int flux_ctx_bvh_info(flux_ctx *ctx, uint64_t *out, uint64_t out_words) {
    if (!ctx || !out) return fail(FLUX_E_INVALID, "null argument");
    uint64_t w[FLUX_BVH_INFO_WORDS] = {};
    w[0] = ctx->bvh.nodes;
    w[1] = ctx->bvh.tris;
    w[2] = ctx->bvh.max_depth;
    w[3] = ctx->bvh.max_leaf;
    w[4] = sizeof(flux::DevNode);
    w[5] = sizeof(flux::DevTri);
    w[6] = ctx->bvh.build_us;
    w[7] = ctx->bvh.wide_nodes;
    w[8] = ctx->bvh.leaf_records;
    w[9] = ctx->bvh.fused_leaves;
    w[10] = ctx->bvh.wide_stack;
    w[11] = sizeof(flux::DevNode4Q);
    w[12] = sizeof(flux::DevLeafRec);
    // [13]: asked of the launch planner (a full-frame render with the current variant, arithmetic and traversal)
    w[13] = flux::plan_render(rows_params(ctx, 0, 1, ctx->H, nullptr), ctx->variant, effective_math(ctx)).kernel == FLUX_PLAN_BVH4 ? 1 : 0;
    for (uint64_t k = 0; k < out_words && k < FLUX_BVH_INFO_WORDS; k++) out[k] = w[k];
    return FLUX_OK;
}

int flux_ctx_stats(flux_ctx *ctx, uint64_t out[FLUX_NUM_STATS], int reset) {
    if (!ctx || !out) return fail(FLUX_E_INVALID, "null argument");
    DeviceGuard guard(ctx->device);
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long tmp[FLUX_NUM_STATS];
    HIP_TRY(hipMemcpy(tmp, ctx->d_stats, sizeof(tmp), hipMemcpyDeviceToHost));
    for (int i = 0; i < FLUX_NUM_STATS; i++) out[i] = tmp[i];
    if (reset) HIP_TRY(hipMemset(ctx->d_stats, 0, sizeof(tmp)));
    return FLUX_OK;
}

int flux_ctx_copy_table(flux_ctx *ctx, int which, double *out, uint64_t out_doubles) {
    if (!ctx || !out) return fail(FLUX_E_INVALID, "null argument");
    DeviceGuard guard(ctx->device);
    const size_t SN = (size_t)ctx->sets.count * ctx->N;  // the sets held here, in slot order
    if (which == FLUX_TABLE_PIXEL || which == FLUX_TABLE_DISC) {
        if (out_doubles < SN * 2) return fail(FLUX_E_INVALID, "output too small: need %zu doubles", SN * 2);
        HIP_TRY(hipMemcpy(out, which == FLUX_TABLE_PIXEL ? ctx->d_pix : ctx->d_disc, SN * 2 * sizeof(double),
                          hipMemcpyDeviceToHost));
        return FLUX_OK;
    }
    if (which == FLUX_TABLE_HEMI) {
        const size_t total = SN * ctx->D * 3;
        if (out_doubles < total) return fail(FLUX_E_INVALID, "output too small: need %zu doubles", total);
        double *tmp = nullptr;
        HIP_TRY(hipMalloc((void **)&tmp, total * sizeof(double)));
        hipError_t e = flux::hemi_to_aos((size_t)ctx->sets.count * ctx->D, ctx->N, ctx->d_hemi, tmp, nullptr);
        if (e == hipSuccess) e = hipMemcpy(out, tmp, total * sizeof(double), hipMemcpyDeviceToHost);
        (void)hipFree(tmp);
        if (e != hipSuccess) return fail(FLUX_E_DEVICE, "hemi copy: %s", hipGetErrorString(e));
        return FLUX_OK;
    }
    return fail(FLUX_E_INVALID, "unknown table %d", which);
}

int flux_ctx_copy_row_perm(flux_ctx *ctx, uint64_t row, int32_t *out, uint64_t out_len) {
    if (!ctx || !out) return fail(FLUX_E_INVALID, "null argument");
    if (row >= ctx->H) return fail(FLUX_E_INVALID, "row %llu outside image height %u", (unsigned long long)row, ctx->H);
    if (out_len < ctx->S) return fail(FLUX_E_INVALID, "output too small: need %u entries", ctx->S);
    DeviceGuard guard(ctx->device);
    HIP_TRY(hipMemcpy(out, ctx->d_rowperm + (size_t)row * ctx->S, (size_t)ctx->S * sizeof(int32_t),
                      hipMemcpyDeviceToHost));
    return FLUX_OK;
}

int flux_ctx_camera_basis(flux_ctx *ctx, double uvw[9]) {
    if (!ctx || !uvw) return fail(FLUX_E_INVALID, "null argument");
    for (int i = 0; i < 3; i++) {
        uvw[i] = ctx->U[i];
        uvw[3 + i] = ctx->V[i];
        uvw[6 + i] = ctx->Wv[i];
    }
    return FLUX_OK;
}

uint64_t flux_ctx_device_bytes(flux_ctx *ctx) { return ctx ? ctx->device_bytes : 0; }

int flux_ctx_create_timing(flux_ctx *ctx, double out_ms[FLUX_CREATE_TIMING_WORDS]) {
    if (!ctx || !out_ms) return fail(FLUX_E_INVALID, "null argument");
    for (int k = 0; k < FLUX_CREATE_TIMING_WORDS; k++) out_ms[k] = ctx->create_ms[k];
    return FLUX_OK;
}

// ---- host-side pieces of the boundary ------------------------------------------

// Job::work_units: job.rs:65-88
int64_t flux_work_units(uint64_t image_height, uint64_t rows_per_work_unit, flux_work_unit *out, uint64_t cap) {
    if (rows_per_work_unit == 0)
        return fail(FLUX_E_INVALID, "Job row per work unit count invalid: 0");  // job.rs:67-70 panics
    if (image_height == 0) return fail(FLUX_E_INVALID, "image_height must be >= 1");
    int64_t count = 0;
    uint64_t i = 0;
    while (i < image_height - 1) {  // sic: job.rs:74
        uint64_t remaining = image_height - i;
        uint64_t rows = rows_per_work_unit < remaining ? rows_per_work_unit : remaining;
        if (out && (uint64_t)count < cap) {
            out[count].row_start = i;
            out[count].row_end = i + rows - 1;
        }
        count++;
        i += rows;
    }
    return count;
}

static unsigned quantize16(double c) {  // `(c * 65535.99) as u16`: image.rs:50-53 (saturating cast)
    double v = c * 65535.99;
    if (!(v > 0.0)) return 0;
    if (v >= 65535.0) return 65535;
    return (unsigned)v;
}

// Image::write: image.rs:43-61
int flux_write_ppm(const char *path, const double *rgb, uint64_t width, uint64_t height,
                   const uint8_t *rows_present) {
    if (!path || !rgb) return fail(FLUX_E_INVALID, "null argument");
    FILE *f = std::fopen(path, "w");
    if (!f) return fail(FLUX_E_IO, "cannot open %s for writing", path);
    std::vector<char> buf(1 << 20);
    std::setvbuf(f, buf.data(), _IOFBF, buf.size());
    std::fprintf(f, "P3\n%llu %llu\n65535\n", (unsigned long long)width, (unsigned long long)height);
    for (uint64_t r = 0; r < height; r++) {
        const bool present = rows_present ? rows_present[r] != 0 : true;
        for (uint64_t col = 0; col < width; col++) {
            if (present) {
                const double *p = rgb + (r * width + col) * 3;
                std::fprintf(f, "%u %u %u\n", quantize16(p[0]), quantize16(p[1]), quantize16(p[2]));
            } else {
                std::fputs("0 0 0\n", f);
            }
        }
    }
    int bad = std::ferror(f);
    if (std::fclose(f) != 0 || bad) return fail(FLUX_E_IO, "write to %s failed", path);
    return FLUX_OK;
}

}  // extern "C"
