// tables.hip -- sample-table generation on the device.
//
// Replaces MasterSampleSets::new (fluxcore/src/sampling.rs:13-33) and the
// samplers crate pipeline beneath it (samplers/src/lib.rs:46-182).  The
// reference builds each set as base grid -> shuffle_y per row -> transpose ->
// shuffle_x per column -> transpose -> flatten; composing those steps gives a
// closed form per output sample, so one GPU thread produces one sample:
//
//   base[i][j] = ( i/n + ((n-1-j) + a_ij)/n^2 ,  j/n + ((n-1-i) + b_ij)/n^2 )   lib.rs:46-62
//   out[i*n+k] = ( base[px_k[i]][k].x ,  base[i][py_i[k]].y )                    lib.rs:64-126
//
// with py_i the y-permutation used for row i and px_k the x-permutation used
// for column k (one shared pair for the correlated variant, lib.rs:75-90).
// The permutations are drawn first by one thread each (shuffle kernel).
#include <chrono>

#include "flux_device.h"
#include "flux_rng.h"
#include "flux_tables.h"

namespace flux {

// ---- permutations --------------------------------------------------------
// perms layout (uint16): CMJ kind : [S][2][n]            (0 = x_idxs, 1 = y_idxs)
//                        MJ (hemi): [S][D][2n][n]         (0..n-1 = y-shuffle of row i, n..2n-1 = x-shuffle of column k)
// Tables may hold a SUBSET of the sample sets (a rank of a set-sharded render builds only the sets it owns): local
// slot m holds global set sets.first + m * sets.stride; the RNG streams are keyed by the GLOBAL index, so a set's
// contents do not depend on which context builds it.
__device__ __forceinline__ uint32_t global_set(SetRange sets, uint32_t slot) { return sets.first + slot * sets.stride; }

__global__ void cmj_perm_kernel(uint64_t seed, uint64_t kind, SetRange sets, uint32_t n, uint16_t *perms) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= sets.count * 2u) return;
    uint32_t s = t >> 1, which = t & 1u;
    shuffle_iota(stream_key(seed, kind, global_set(sets, s), 0, 1 + which), perms + (size_t)t * n, n);
}

__global__ void mj_perm_kernel(uint64_t seed, SetRange sets, uint32_t D, uint32_t n, uint16_t *perms) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)sets.count * D * 2u * n;
    if (t >= total) return;
    uint32_t sub = (uint32_t)(t % (2u * n));
    size_t sd = t / (2u * n);
    uint32_t d = (uint32_t)(sd % D), s = (uint32_t)(sd / D);
    shuffle_iota(stream_key(seed, kKindHemi, global_set(sets, s), d, 1 + sub), perms + t * n, n);
}

// shuffle_indices (sampling.rs:35-40) for every image row, keyed by (seed,row)
__global__ void row_perm_kernel(uint64_t seed, uint32_t H, uint32_t S, int32_t *rowperm) {
    uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= H) return;
    shuffle_iota(stream_key(seed, kKindRowPerm, row, 0, 0), rowperm + (size_t)row * S, S);
}

// inverse of each row's permutation: invperm[row][set] = the column whose pixel uses `set`
__global__ void inv_perm_kernel(uint32_t H, uint32_t S, const int32_t *__restrict__ rowperm, int32_t *__restrict__ invperm) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)H * S) return;
    const size_t row = t / S;
    invperm[row * S + (size_t)rowperm[t]] = (int32_t)(t - row * S);
}

// ---- sample maps ---------------------------------------------------------
__device__ __forceinline__ double2 mj_point(uint64_t jkey, uint32_t n, uint32_t i, uint32_t k,
                                            uint32_t xi /* px_k[i] */, uint32_t yk /* py_i[k] */) {
    const double rf = (double)n;
    const double r2 = (double)((uint64_t)n * n);
    // x of base[xi][k]
    double a = unit(jkey, 2ull * ((uint64_t)xi * n + k));
    double x = ((double)xi / rf) + ((double)(n - 1 - k) + a) / r2;
    // y of base[i][yk]
    double b = unit(jkey, 2ull * ((uint64_t)i * n + yk) + 1ull);
    double y = ((double)yk / rf) + ((double)(n - 1 - i) + b) / r2;
    return make_double2(x, y);
}

// to_poisson_disc: samplers/src/lib.rs:144-182
__device__ __forceinline__ double2 to_disc(double2 p) {
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
            phi = (spy != 0.0) ? 6.0 - spx / spy : 0.0;
        }
    }
    phi *= kPi / 4.0;
    return make_double2(r * cos(phi), r * sin(phi));
}

// to_unit_hemi(p, e = 0.0): lib.rs:133-142 (what to_hemisphere(.., 0.0) applies, lib.rs:129-131)
__device__ __forceinline__ void unit_hemi_e0(double2 q, double &ox, double &oy, double &oz) {
    double cos_phi = cos(2.0 * kPi * q.x);
    double sin_phi = sin(2.0 * kPi * q.x);
    double cos_theta = pow(1.0 - q.y, 1.0 / (0.0 + 1.0));
    double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
    double pu = sin_theta * cos_phi, pv = sin_theta * sin_phi, pw = cos_theta;
    double len = sqrt(pu * pu + pv * pv + pw * pw);
    ox = pu / len;
    oy = pv / len;
    oz = pw / len;
}

// pixel_sets (sampling.rs:16-17) and disc_sets (sampling.rs:19-21)
__global__ void cmj_fill_kernel(uint64_t seed, uint64_t kind, SetRange sets, uint32_t n,
                                const uint16_t *__restrict__ perms, double2 *__restrict__ out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t N = (size_t)n * n;
    if (t >= (size_t)sets.count * N) return;
    uint32_t s = (uint32_t)(t / N);
    uint32_t p = (uint32_t)(t % N);
    uint32_t i = p / n, k = p % n;
    const uint16_t *xp = perms + ((size_t)s * 2) * n;
    const uint16_t *yp = xp + n;
    double2 q = mj_point(stream_key(seed, kind, global_set(sets, s), 0, kSubJitter), n, i, k, xp[i], yp[k]);
    out[t] = (kind == kKindDisc) ? to_disc(q) : q;
}

// hemi_sets: to_hemisphere(grid_multi_jittered(n), 0.0) per (set, depth)
// (sampling.rs:23-29, lib.rs:129-142), stored SoA [S][D][3][N].
__global__ void hemi_fill_kernel(uint64_t seed, SetRange sets, uint32_t D, uint32_t n,
                                 const uint16_t *__restrict__ perms, double *__restrict__ out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t N = (size_t)n * n;
    if (t >= (size_t)sets.count * D * N) return;
    size_t sd = t / N;
    uint32_t p = (uint32_t)(t % N);
    uint32_t d = (uint32_t)(sd % D), s = (uint32_t)(sd / D);
    uint32_t i = p / n, k = p % n;
    const uint16_t *pb = perms + sd * (2ull * n) * n;
    uint32_t yk = pb[(size_t)i * n + k];        // y-shuffle of row i, element k
    uint32_t xi = pb[((size_t)n + k) * n + i];  // x-shuffle of column k, element i
    double2 q = mj_point(stream_key(seed, kKindHemi, global_set(sets, s), d, kSubJitter), n, i, k, xi, yk);
    double pu, pv, pw;
    unit_hemi_e0(q, pu, pv, pw);
    double *o = out + (sd * N + p) * 4;  // [S][D][N][4]: one 32-B sector per sample
    o[0] = pu;
    o[1] = pv;
    o[2] = pw;
    o[3] = 0.0;
}

// SoA [S][D][3][N] -> reference order [S][D][N][3] (introspection only)
__global__ void hemi_to_aos_kernel(size_t SD, size_t N, const double *__restrict__ in,
                                   double *__restrict__ out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= SD * N) return;
    size_t sd = t / N, p = t % N;
    const double *b = in + (sd * N + p) * 4;
    out[t * 3] = b[0];
    out[t * 3 + 1] = b[1];
    out[t * 3 + 2] = b[2];
}

// One set of any of the samplers crate's four generators (sampler-debug/src/main.rs:48-57) and, optionally,
// its to_hemisphere(.., 0.0) image: kind 0 grid_regular (lib.rs:184-191), 1 grid_jittered (lib.rs:35-44),
// 2 grid_multi_jittered (= hemi stream, set 0, depth 0), 3 grid_correlated_multi_jittered (= pixel stream, set 0).
__global__ void sampler_grid_kernel(int kind, uint64_t seed, uint32_t n, const uint16_t *__restrict__ perms,
                                    double *__restrict__ out_xy, double *__restrict__ out_hemi) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t N = n * n;
    if (p >= N) return;
    const uint32_t i = p / n, k = p % n;
    double2 q;
    if (kind <= 1) {
        const double increment = 1.0 / (double)n;
        const double start = 0.5 * increment;
        q = make_double2(start + increment * (double)i, start + increment * (double)k);
        if (kind == 1) {
            const uint64_t key = stream_key(seed, kKindDebugJitter, 0, 0, 0);
            q.x = q.x + (unit(key, 2ull * p) - 0.5) * increment;
            q.y = q.y + (unit(key, 2ull * p + 1ull) - 0.5) * increment;
        }
    } else if (kind == 2) {
        const uint32_t yk = perms[(size_t)i * n + k];
        const uint32_t xi = perms[((size_t)n + k) * n + i];
        q = mj_point(stream_key(seed, kKindHemi, 0, 0, kSubJitter), n, i, k, xi, yk);
    } else {
        q = mj_point(stream_key(seed, kKindPixel, 0, 0, kSubJitter), n, i, k, perms[i], perms[n + k]);
    }
    out_xy[2 * (size_t)p] = q.x;
    out_xy[2 * (size_t)p + 1] = q.y;
    if (out_hemi) unit_hemi_e0(q, out_hemi[3 * (size_t)p], out_hemi[3 * (size_t)p + 1], out_hemi[3 * (size_t)p + 2]);
}

static inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

hipError_t generate_sampler_grid(int kind, uint64_t seed, uint32_t n, double *d_xy, double *d_hemi,
                                 hipStream_t stream) {
    uint16_t *perms = nullptr;
    hipError_t e = hipSuccess;
    if (kind == 2) {
        if ((e = hipMalloc(&perms, (size_t)2 * n * n * sizeof(uint16_t))) != hipSuccess) return e;
        mj_perm_kernel<<<blocks_for((size_t)2 * n, 64), 64, 0, stream>>>(seed, SetRange{0, 1, 1}, 1, n, perms);
    } else if (kind == 3) {
        if ((e = hipMalloc(&perms, (size_t)2 * n * sizeof(uint16_t))) != hipSuccess) return e;
        cmj_perm_kernel<<<1, 64, 0, stream>>>(seed, kKindPixel, SetRange{0, 1, 1}, n, perms);
    }
    sampler_grid_kernel<<<blocks_for((size_t)n * n, 256), 256, 0, stream>>>(kind, seed, n, perms, d_xy, d_hemi);
    e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(stream);
    (void)hipFree(perms);
    return e != hipSuccess ? e : e2;
}

hipError_t generate_tables(uint64_t seed, uint32_t S, SetRange sets, uint32_t D, uint32_t n, uint32_t H,
                           double2 *pix, double2 *disc, double *hemi, int32_t *rowperm, int32_t *invperm,
                           hipStream_t stream, double *phase_ms) {
    const size_t N = (size_t)n * n;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](int k) {  // phase_ms[0] scratch allocation, [1] launches + the wait for the kernels, [2] scratch release
        const auto now = std::chrono::steady_clock::now();
        if (phase_ms) phase_ms[k] += std::chrono::duration<double, std::milli>(now - t_last).count();
        t_last = now;
    };
    const uint32_t So = sets.count;  // sets held by this context
    uint16_t *cmj_perms = nullptr, *mj_perms = nullptr;
    hipError_t e;
    if (So == 0) {  // an empty share (more ranks than sample sets): no per-set tables, only who-uses-which-set
        row_perm_kernel<<<blocks_for(H, 64), 64, 0, stream>>>(seed, H, S, rowperm);
        inv_perm_kernel<<<blocks_for((size_t)H * S, 256), 256, 0, stream>>>(H, S, rowperm, invperm);
        e = hipGetLastError();
        const hipError_t e2 = hipStreamSynchronize(stream);
        lap(1);
        return e != hipSuccess ? e : e2;
    }
    size_t cmj_elems = (size_t)So * 2 * n;
    size_t mj_elems = (size_t)So * D * 2 * n * n;
    if ((e = hipMalloc(&cmj_perms, 2 * cmj_elems * sizeof(uint16_t))) != hipSuccess) return e;
    if ((e = hipMalloc(&mj_perms, mj_elems * sizeof(uint16_t))) != hipSuccess) {
        (void)hipFree(cmj_perms);
        return e;
    }
    uint16_t *pix_perms = cmj_perms, *disc_perms = cmj_perms + cmj_elems;
    const unsigned bs = 256;
    lap(0);
    cmj_perm_kernel<<<blocks_for((size_t)So * 2, 64), 64, 0, stream>>>(seed, kKindPixel, sets, n, pix_perms);
    cmj_perm_kernel<<<blocks_for((size_t)So * 2, 64), 64, 0, stream>>>(seed, kKindDisc, sets, n, disc_perms);
    mj_perm_kernel<<<blocks_for((size_t)So * D * 2 * n, bs), bs, 0, stream>>>(seed, sets, D, n, mj_perms);
    row_perm_kernel<<<blocks_for(H, 64), 64, 0, stream>>>(seed, H, S, rowperm);  // always all S sets: who uses which
    inv_perm_kernel<<<blocks_for((size_t)H * S, bs), bs, 0, stream>>>(H, S, rowperm, invperm);
    cmj_fill_kernel<<<blocks_for((size_t)So * N, bs), bs, 0, stream>>>(seed, kKindPixel, sets, n, pix_perms, pix);
    cmj_fill_kernel<<<blocks_for((size_t)So * N, bs), bs, 0, stream>>>(seed, kKindDisc, sets, n, disc_perms, disc);
    hemi_fill_kernel<<<blocks_for((size_t)So * D * N, bs), bs, 0, stream>>>(seed, sets, D, n, mj_perms, hemi);
    e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(stream);
    lap(1);
    (void)hipFree(cmj_perms);
    (void)hipFree(mj_perms);
    lap(2);
    return e != hipSuccess ? e : e2;
}

hipError_t hemi_to_aos(size_t SD, size_t N, const double *in, double *out, hipStream_t stream) {
    hemi_to_aos_kernel<<<blocks_for(SD * N, 256), 256, 0, stream>>>(SD, N, in, out);
    return hipGetLastError();
}

}  // namespace flux
