// multi.hip -- one frame on the GPUs of one node, behind the C ABI (include/flux_abi.h flux_multi_*).
//
// What it replaces in the reference: RenderManager hands a clone of the job to every worker (fluxcore/src/manager.rs:156-162),
// the workers pull WorkUnits from one shared channel (manager.rs:100, workers.rs:56-60) and ImageBuilder places the rows they
// send back by row_start (manager.rs:316-324).  For the GPUs of one process the same two steps are
//   fan-out : one flux_ctx per device, created concurrently, each holding its 1/G share of the sample tables, and ONE kernel
//             launch per device and frame on that device's own stream -- a static split by sample set (or by interleaved rows),
//             balanced by construction, instead of 12 fifty-row units pulled by 8 workers (two rounds, 75 % at best);
//   gather  : ONE collective -- ncclAllGather (RCCL) of the shares, device to device over xGMI -- then one kernel on
//             devices[0] that reads the gathered shares through the row permutation into the frame, and one copy to the caller.
// The frame never passes through the host before it is complete.
//
// RCCL is bound with dlopen at the first flux_multi_create: libflux_hip.so keeps no link-time dependency on it, and in a
// process that has already mapped an RCCL (PyTorch ships its own beside its own HIP runtime) that copy is the one used, so the
// process keeps ONE RCCL on ONE HIP runtime (flux_amd/_lib.py says why that matters).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: every call goes through the table below

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "flux_ctx.h"

using flux::DeviceGuard;
using flux::fail;

namespace {

// ---- RCCL, bound at run time ---------------------------------------------------------------------------------------
struct Rccl {
    void *handle = nullptr;
    std::string origin;  // how the library was found (flux_multi_info does not expose it; error messages do)
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
};

std::mutex g_rccl_mu;
Rccl g_rccl;

// under g_rccl_mu
bool rccl_load(std::string &err) {
    if (g_rccl.handle) return true;
    std::vector<std::pair<std::string, int>> tries;
    if (const char *env = std::getenv("FLUX_RCCL_LIB")) tries.push_back({env, RTLD_NOW | RTLD_LOCAL});
    // a copy the process has mapped already (SONAME librccl.so.1 in both ROCm's and PyTorch's builds) wins: never a second RCCL
    tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD});
    tries.push_back({"librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD});
    tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_LOCAL});  // libflux_hip.so's RUNPATH (/opt/rocm/lib), then the loader's search path
    tries.push_back({"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL});
    tries.push_back({"librccl.so", RTLD_NOW | RTLD_LOCAL});
    void *h = nullptr;
    std::string last;
    for (const auto &t : tries) {
        h = dlopen(t.first.c_str(), t.second);
        if (h) {
            g_rccl.origin = t.first + ((t.second & RTLD_NOLOAD) ? " (already mapped)" : "");
            break;
        }
        if (!(t.second & RTLD_NOLOAD))
            if (const char *m = dlerror()) last = m;
    }
    if (!h) {
        err = "RCCL not found (tried FLUX_RCCL_LIB, librccl.so.1, /opt/rocm/lib/librccl.so.1): " + last;
        return false;
    }
    Rccl r;
    r.handle = h;
    r.origin = g_rccl.origin;
#define FLUX_RCCL_SYM(field, name)                                          \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));          \
    if (!r.field) {                                                         \
        err = std::string("RCCL (") + r.origin + ") lacks the symbol " name; \
        dlclose(h);                                                         \
        return false;                                                       \
    }
    FLUX_RCCL_SYM(GetVersion, "ncclGetVersion")
    FLUX_RCCL_SYM(GetErrorString, "ncclGetErrorString")
    FLUX_RCCL_SYM(CommInitAll, "ncclCommInitAll")
    FLUX_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    FLUX_RCCL_SYM(AllGather, "ncclAllGather")
    FLUX_RCCL_SYM(GroupStart, "ncclGroupStart")
    FLUX_RCCL_SYM(GroupEnd, "ncclGroupEnd")
#undef FLUX_RCCL_SYM
    g_rccl = r;
    return true;
}

// Communicators per device list, kept for the life of the process (ncclCommInitAll costs far more than a frame; a front-end that
// schedules job after job -- flux/src/main.rs:247,304,313 -- creates a flux_multi per job): under g_rccl_mu.
std::map<std::vector<int>, std::vector<ncclComm_t>> g_comms;

double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// ---- reassembly: the counterpart of ImageBuilder placing rows (manager.rs:316-324), on the device --------------------
// sets: pixel (r, c) uses set s = rowperm[r][c]; rank s mod G rendered it as column s div G of its [H][cmax][3] share
__global__ void assemble_sets_kernel(const double *__restrict__ gathered, const int32_t *__restrict__ rowperm,
                                     double *__restrict__ frame, uint32_t H, uint32_t W, uint32_t G, uint32_t cmax) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)H * W) return;
    const uint32_t r = (uint32_t)(t / W);
    const uint32_t s = (uint32_t)rowperm[t];
    const uint32_t g = s % G, m = s / G;
    const double *src = gathered + (((size_t)g * H + r) * cmax + m) * 3;
    double *dst = frame + t * 3;
    dst[0] = src[0];
    dst[1] = src[1];
    dst[2] = src[2];
}
// rows: image row r is row r div G of rank r mod G's [rmax][W][3] share
__global__ void assemble_rows_kernel(const double *__restrict__ gathered, double *__restrict__ frame, uint32_t H, uint32_t W,
                                     uint32_t G, uint32_t rmax) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)H * W * 3) return;
    const size_t per_row = (size_t)W * 3;
    const uint32_t r = (uint32_t)(t / per_row);
    const size_t within = t - (size_t)r * per_row;
    frame[t] = gathered[((size_t)(r % G) * rmax + r / G) * per_row + within];
}

struct Rank {
    int device = 0;
    flux_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    uint64_t count = 0;           // sets (or rows) this rank renders
    double *d_render = nullptr;   // sets with count < cmax: the kernel's dense [H][count][3] output, copied into the padded share
    double *d_share = nullptr;    // what this rank contributes to the gather: [H][cmax][3] or [rmax][W][3], padding zero
    double *d_gathered = nullptr; // [G] x share
    ncclComm_t comm = nullptr;
    std::string error;            // of its creation thread
    int rc = FLUX_OK;
    double create_ms = 0;
};

}  // namespace

struct flux_multi {
    std::vector<Rank> ranks;
    int shard = FLUX_SHARD_SETS;
    bool loopback = false;       // FLUX_SHARD_LOOPBACK: ranks may share devices, the gather is device-to-device copies (no RCCL)
    uint32_t W = 0, H = 0, S = 0;
    uint64_t per_rank = 0;       // cmax (sets) or rmax (rows)
    size_t share_doubles = 0;    // doubles of one rank's share
    double *d_frame = nullptr;   // devices[0]: [H][W][3]
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // devices[0]'s stream: kernel end, gather end, reassembly end, copy end
    double timing[FLUX_MULTI_TIMING_WORDS] = {};
    bool comms_cached = false;
    int rccl_version = 0;
    uint64_t buffer_bytes = 0;
};

namespace {

void free_multi(flux_multi *m) {
    if (!m) return;
    for (Rank &rk : m->ranks) {
        DeviceGuard g(rk.device);
        if (rk.stream) (void)hipStreamSynchronize(rk.stream);
        flux_ctx_destroy(rk.ctx);
        (void)hipFree(rk.d_render);
        (void)hipFree(rk.d_share);
        (void)hipFree(rk.d_gathered);
        if (rk.stream) (void)hipStreamDestroy(rk.stream);
    }
    if (!m->ranks.empty()) {
        DeviceGuard g(m->ranks[0].device);
        (void)hipFree(m->d_frame);
        for (hipEvent_t e : m->ev)
            if (e) (void)hipEventDestroy(e);
    }
    delete m;  // (the communicators stay in the process-wide cache)
}

// the frame on devices[0]; when out_rgb != nullptr also copied to the host
int render_frame(flux_multi *m, double *out_rgb) {
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t G = (uint32_t)m->ranks.size();
    Rccl rccl;
    {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        rccl = g_rccl;
    }
    // fan-out: one launch per device, each on its own stream (asynchronous: this thread feeds all devices)
    for (uint32_t g = 0; g < G; g++) {
        Rank &rk = m->ranks[g];
        if (rk.count == 0) continue;
        int rc;
        if (m->shard == FLUX_SHARD_SETS) {
            double *dst = rk.d_render ? rk.d_render : rk.d_share;
            rc = flux_render_sets_device(rk.ctx, g, G, rk.count, dst, rk.stream);
            if (rc == FLUX_OK && rk.d_render) {
                DeviceGuard dg(rk.device);
                HIP_TRY(hipMemcpy2DAsync(rk.d_share, (size_t)m->per_rank * 24, rk.d_render, (size_t)rk.count * 24, (size_t)rk.count * 24,
                                         m->H, hipMemcpyDeviceToDevice, rk.stream));
            }
        } else {
            rc = flux_render_rows_device(rk.ctx, g, G, rk.count, rk.d_share, rk.stream);
        }
        if (rc != FLUX_OK) return rc;
    }
    Rank &root = m->ranks[0];
    {
        DeviceGuard dg(root.device);
        HIP_TRY(hipEventRecord(m->ev[0], root.stream));
    }
    if (m->loopback) {
        // the test hook's stand-in for the collective: every share copied into every rank's gather buffer, device to device
        for (uint32_t g = 0; g < G; g++) {
            DeviceGuard dg(m->ranks[g].device);
            HIP_TRY(hipStreamSynchronize(m->ranks[g].stream));
        }
        for (uint32_t dst = 0; dst < G; dst++) {
            Rank &rk = m->ranks[dst];
            DeviceGuard dg(rk.device);
            for (uint32_t src = 0; src < G; src++)
                HIP_TRY(hipMemcpyAsync(rk.d_gathered + (size_t)src * m->share_doubles, m->ranks[src].d_share, m->share_doubles * sizeof(double),
                                       hipMemcpyDeviceToDevice, rk.stream));
        }
    } else {
        // gather: ONE collective over all devices (a group, since one thread drives every communicator)
        ncclResult_t nr = rccl.GroupStart();
        for (uint32_t g = 0; g < G && nr == ncclSuccess; g++) {
            Rank &rk = m->ranks[g];
            DeviceGuard dg(rk.device);
            nr = rccl.AllGather(rk.d_share, rk.d_gathered, m->share_doubles, ncclDouble, rk.comm, rk.stream);
        }
        const ncclResult_t ne = rccl.GroupEnd();
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess) return fail(FLUX_E_DEVICE, "ncclAllGather over %u device(s): %s", G, rccl.GetErrorString(nr));
    }
    {
        DeviceGuard dg(root.device);
        HIP_TRY(hipEventRecord(m->ev[1], root.stream));
        const unsigned bs = 256;
        if (m->shard == FLUX_SHARD_SETS) {
            const size_t n = (size_t)m->H * m->W;
            assemble_sets_kernel<<<dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, root.stream>>>(
                root.d_gathered, root.ctx->d_rowperm, m->d_frame, m->H, m->W, G, (uint32_t)m->per_rank);
        } else {
            const size_t n = (size_t)m->H * m->W * 3;
            assemble_rows_kernel<<<dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, root.stream>>>(root.d_gathered, m->d_frame, m->H, m->W, G,
                                                                                                     (uint32_t)m->per_rank);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(m->ev[2], root.stream));
        if (out_rgb) HIP_TRY(hipMemcpyAsync(out_rgb, m->d_frame, (size_t)m->H * m->W * 3 * sizeof(double), hipMemcpyDeviceToHost, root.stream));
        HIP_TRY(hipEventRecord(m->ev[3], root.stream));
    }
    // every device has left the collective before the call returns (the next frame overwrites the shares)
    for (uint32_t g = 0; g < G; g++) {
        Rank &rk = m->ranks[g];
        DeviceGuard dg(rk.device);
        HIP_TRY(hipStreamSynchronize(rk.stream));
    }
    m->timing[3] = ms_since(t0);
    double kmax = 0;
    for (Rank &rk : m->ranks)
        if (rk.count) {
            const double k = flux_ctx_last_kernel_ms(rk.ctx);
            if (k > kmax) kmax = k;
        }
    m->timing[4] = kmax;
    {
        DeviceGuard dg(root.device);
        float f = 0.f;
        for (int k = 0; k < 3; k++) {
            HIP_TRY(hipEventElapsedTime(&f, m->ev[k], m->ev[k + 1]));
            m->timing[5 + k] = f;
        }
        if (!out_rgb) m->timing[7] = 0.0;
    }
    return FLUX_OK;
}

}  // namespace

extern "C" {

int flux_multi_create(const flux_scene_desc *scene, const flux_job_cfg *cfg, uint64_t seed, const int *devices, uint64_t num_devices,
                      int shard, flux_multi **out) {
    if (!scene || !cfg || !devices || !out) return fail(FLUX_E_INVALID, "flux_multi_create: null argument");
    *out = nullptr;
    if (num_devices < 1 || num_devices > 64) return fail(FLUX_E_INVALID, "flux_multi_create: 1..64 devices, got %llu", (unsigned long long)num_devices);
    const bool loopback = (shard & FLUX_SHARD_LOOPBACK) != 0;
    shard &= ~FLUX_SHARD_LOOPBACK;
    if (shard != FLUX_SHARD_AUTO && shard != FLUX_SHARD_SETS && shard != FLUX_SHARD_ROWS) return fail(FLUX_E_INVALID, "unknown shard mode %d", shard);
    const int ndev = flux_device_count();
    if (ndev < 1) return fail(FLUX_E_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    for (uint64_t a = 0; a < num_devices; a++) {
        if (devices[a] < 0 || devices[a] >= ndev) return fail(FLUX_E_INVALID, "device %d out of range [0,%d)", devices[a], ndev);
        for (uint64_t b = 0; b < a && !loopback; b++)
            if (devices[a] == devices[b]) return fail(FLUX_E_INVALID, "device %d is listed twice", devices[a]);
    }
    const uint64_t spp = cfg->sample_root * cfg->sample_root;
    if (shard == FLUX_SHARD_SETS && spp < 64) return fail(FLUX_E_INVALID, "FLUX_SHARD_SETS needs sample_root^2 >= 64 (use FLUX_SHARD_AUTO or FLUX_SHARD_ROWS)");
    if (shard == FLUX_SHARD_AUTO) shard = spp >= 64 ? FLUX_SHARD_SETS : FLUX_SHARD_ROWS;
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t G = (uint32_t)num_devices;

    std::string err;
    if (!loopback) {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        if (!rccl_load(err)) return fail(FLUX_E_DEVICE, "%s", err.c_str());
    }
    flux_multi *m = new (std::nothrow) flux_multi();
    if (!m) return fail(FLUX_E_NOMEM, "host allocation failed");
    m->shard = shard;
    m->loopback = loopback;
    m->ranks.resize(G);
    for (uint32_t g = 0; g < G; g++) m->ranks[g].device = devices[g];

    // communicators: from the cache, or created on a thread of their own while the contexts come up
    const std::vector<int> key(devices, devices + G);
    std::vector<ncclComm_t> comms;
    std::string comm_err;
    double comm_ms = 0;
    std::thread comm_thread;
    if (loopback) {
        comms.assign(G, nullptr);
    } else {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        auto it = g_comms.find(key);
        if (it != g_comms.end()) {
            comms = it->second;
            m->comms_cached = true;
        }
        (void)g_rccl.GetVersion(&m->rccl_version);
    }
    if (!m->comms_cached && !loopback)
        comm_thread = std::thread([&] {
            const auto tc = std::chrono::steady_clock::now();
            std::vector<ncclComm_t> cs(G, nullptr);
            const ncclResult_t r = g_rccl.CommInitAll(cs.data(), (int)G, key.data());
            if (r != ncclSuccess)
                comm_err = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r);
            else
                comms = cs;
            comm_ms = ms_since(tc);
        });
    // fan-out of the job (manager.rs:156-162): Scene::from_data + Camera::new per device, concurrently
    std::vector<std::thread> th;
    for (uint32_t g = 0; g < G; g++)
        th.emplace_back([&, g] {
            Rank &rk = m->ranks[g];
            const auto tc = std::chrono::steady_clock::now();
            rk.rc = shard == FLUX_SHARD_SETS ? flux_ctx_create_sets(scene, cfg, seed, rk.device, g, G, &rk.ctx)
                                             : flux_ctx_create(scene, cfg, seed, rk.device, &rk.ctx);
            if (rk.rc != FLUX_OK) rk.error = flux_last_error();
            rk.create_ms = ms_since(tc);
        });
    for (std::thread &t : th) t.join();
    if (comm_thread.joinable()) comm_thread.join();
    if (!m->comms_cached && !loopback && comm_err.empty()) {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        auto ins = g_comms.emplace(key, comms);
        if (!ins.second) {  // another thread created the same list meanwhile: keep the cached ones, drop ours
            for (ncclComm_t c : comms) (void)g_rccl.CommDestroy(c);
            comms = ins.first->second;
        }
    }
    for (uint32_t g = 0; g < G; g++)
        if (m->ranks[g].rc != FLUX_OK) {
            const int rc = fail(m->ranks[g].rc, "device %d: %s", m->ranks[g].device, m->ranks[g].error.c_str());
            free_multi(m);
            return rc;
        }
    if (!comm_err.empty()) {
        free_multi(m);
        return fail(FLUX_E_DEVICE, "%s", comm_err.c_str());
    }
    m->W = m->ranks[0].ctx->W;
    m->H = m->ranks[0].ctx->H;
    m->S = m->ranks[0].ctx->S;
    if (shard == FLUX_SHARD_SETS) {
        m->per_rank = (m->S + G - 1) / G;
        m->share_doubles = (size_t)m->H * m->per_rank * 3;
    } else {
        m->per_rank = (m->H + G - 1) / G;
        m->share_doubles = (size_t)m->per_rank * m->W * 3;
    }
    hipError_t e = hipSuccess;
    for (uint32_t g = 0; g < G && e == hipSuccess; g++) {
        Rank &rk = m->ranks[g];
        rk.comm = comms[g];
        const uint64_t total = shard == FLUX_SHARD_SETS ? m->S : m->H;
        rk.count = g < total ? (total - g + G - 1) / G : 0;
        DeviceGuard dg(rk.device);
        e = hipStreamCreateWithFlags(&rk.stream, hipStreamNonBlocking);
        const size_t share_bytes = m->share_doubles * sizeof(double);
        if (e == hipSuccess) e = hipMalloc((void **)&rk.d_share, share_bytes);
        if (e == hipSuccess) e = hipMemset(rk.d_share, 0, share_bytes);  // padding stays zero (image.rs:55-59 writes never-received rows as zeros)
        if (e == hipSuccess) e = hipMalloc((void **)&rk.d_gathered, share_bytes * G);
        m->buffer_bytes += share_bytes * (G + 1);
        if (e == hipSuccess && shard == FLUX_SHARD_SETS && rk.count && rk.count != m->per_rank) {
            e = hipMalloc((void **)&rk.d_render, (size_t)m->H * rk.count * 24);
            m->buffer_bytes += (size_t)m->H * rk.count * 24;
        }
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) {
        DeviceGuard dg(m->ranks[0].device);
        e = hipMalloc((void **)&m->d_frame, (size_t)m->H * m->W * 24);
        m->buffer_bytes += (size_t)m->H * m->W * 24;
        for (int k = 0; k < 4 && e == hipSuccess; k++) e = hipEventCreate(&m->ev[k]);
    }
    if (e != hipSuccess) {
        const int rc = fail(e == hipErrorOutOfMemory ? FLUX_E_NOMEM : FLUX_E_DEVICE, "flux_multi_create: %s", hipGetErrorString(e));
        free_multi(m);
        return rc;
    }
    m->timing[0] = ms_since(t0);
    for (const Rank &rk : m->ranks)
        if (rk.create_ms > m->timing[1]) m->timing[1] = rk.create_ms;
    m->timing[2] = comm_ms;
    *out = m;
    return FLUX_OK;
}

void flux_multi_destroy(flux_multi *m) { free_multi(m); }

int flux_multi_render_frame(flux_multi *m, double *out_rgb) {
    if (!m) return fail(FLUX_E_INVALID, "null flux_multi");
    if (!out_rgb) return fail(FLUX_E_INVALID, "null output pointer");
    return render_frame(m, out_rgb);
}

int flux_multi_render_frame_device(flux_multi *m, const void **d_frame_rgb) {
    if (!m || !d_frame_rgb) return fail(FLUX_E_INVALID, "null argument");
    *d_frame_rgb = nullptr;
    const int rc = render_frame(m, nullptr);
    if (rc == FLUX_OK) *d_frame_rgb = m->d_frame;
    return rc;
}

int flux_multi_set_kernel(flux_multi *m, int variant) {
    if (!m) return fail(FLUX_E_INVALID, "null flux_multi");
    for (Rank &rk : m->ranks)
        if (int rc = flux_ctx_set_kernel(rk.ctx, variant)) return rc;
    return FLUX_OK;
}

int flux_multi_set_math(flux_multi *m, int mode) {
    if (!m) return fail(FLUX_E_INVALID, "null flux_multi");
    for (Rank &rk : m->ranks)
        if (int rc = flux_ctx_set_math(rk.ctx, mode)) return rc;
    return FLUX_OK;
}

int flux_multi_ctx(flux_multi *m, uint64_t rank, flux_ctx **ctx) {
    if (!m || !ctx) return fail(FLUX_E_INVALID, "null argument");
    if (rank >= m->ranks.size()) return fail(FLUX_E_INVALID, "rank %llu outside the %zu devices", (unsigned long long)rank, m->ranks.size());
    *ctx = m->ranks[rank].ctx;
    return FLUX_OK;
}

int flux_multi_info(flux_multi *m, uint64_t out[FLUX_MULTI_INFO_WORDS]) {
    if (!m || !out) return fail(FLUX_E_INVALID, "null argument");
    uint64_t ctx_bytes = 0;
    for (Rank &rk : m->ranks) ctx_bytes += flux_ctx_device_bytes(rk.ctx);
    out[0] = m->ranks.size();
    out[1] = (uint64_t)m->shard;
    out[2] = (uint64_t)m->rccl_version;
    out[3] = m->share_doubles;
    out[4] = ctx_bytes;
    out[5] = m->buffer_bytes;
    out[6] = m->comms_cached ? 1 : 0;
    out[7] = 0;
    return FLUX_OK;
}

int flux_multi_timing(flux_multi *m, double out_ms[FLUX_MULTI_TIMING_WORDS]) {
    if (!m || !out_ms) return fail(FLUX_E_INVALID, "null argument");
    for (int k = 0; k < FLUX_MULTI_TIMING_WORDS; k++) out_ms[k] = m->timing[k];
    return FLUX_OK;
}

int flux_multi_release_comms(void) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    int n = 0;
    for (auto &kv : g_comms) {
        for (ncclComm_t c : kv.second)
            if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
        n++;
    }
    g_comms.clear();
    return n;
}

int flux_render_frame_multi(const flux_scene_desc *scene, const flux_job_cfg *cfg, uint64_t seed, const int *devices, uint64_t num_devices,
                            int shard, double *out_rgb) {
    if (!out_rgb) return fail(FLUX_E_INVALID, "null output pointer");
    std::vector<int> all;
    if (!devices) {
        const int ndev = flux_device_count();
        if (ndev < 1) return fail(FLUX_E_DEVICE, "no HIP device visible (this library has no CPU fallback)");
        if (num_devices == 0) num_devices = (uint64_t)ndev;
        if (num_devices > (uint64_t)ndev) return fail(FLUX_E_INVALID, "%llu devices requested, %d visible", (unsigned long long)num_devices, ndev);
        for (uint64_t d = 0; d < num_devices; d++) all.push_back((int)d);
        devices = all.data();
    }
    flux_multi *m = nullptr;
    int rc = flux_multi_create(scene, cfg, seed, devices, num_devices, shard, &m);
    if (rc != FLUX_OK) return rc;
    rc = flux_multi_render_frame(m, out_rgb);
    flux_multi_destroy(m);
    return rc;
}

}  // extern "C"
