// flux_host.cpp -- see flux_host.hpp.
#include "flux_host.hpp"

#include <random>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "yaml_lite.hpp"

namespace flux_host {
namespace {

using yaml_lite::Node;

[[noreturn]] void bad(const std::string &m) { throw FluxError(FLUX_E_INVALID, m); }

const Node &req(const Node &m, const char *key, const std::string &what) {
    if (m.kind != Node::Map) bad(what + ": expected a map");
    const Node *n = m.find(key);
    if (!n) bad(what + ": missing field `" + key + "`");
    return *n;
}

double num(const Node &n, const std::string &what) {
    if (n.kind != Node::Scalar) bad(what + ": expected a number");
    char *end = nullptr;
    double v = std::strtod(n.scalar.c_str(), &end);
    if (end == n.scalar.c_str() || *end != '\0') bad(what + ": expected a number, got `" + n.scalar + "`");
    return v;
}

size_t usize(const Node &n, const std::string &what) {
    double v = num(n, what);
    if (v < 0 || v != (double)(size_t)v) bad(what + ": expected an unsigned integer");
    return (size_t)v;
}

bool boolean(const Node &n, const std::string &what) {
    if (n.kind == Node::Scalar) {
        if (n.scalar == "true") return true;
        if (n.scalar == "false") return false;
    }
    bad(what + ": expected a boolean");
}

Vec3 vec3(const Node &n, const std::string &what) {
    if (n.kind != Node::Seq || n.seq.size() != 3) bad(what + ": expected a sequence of 3 numbers");
    return Vec3{num(n.seq[0], what), num(n.seq[1], what), num(n.seq[2], what)};
}
Color color(const Node &n, const std::string &what) {
    Vec3 v = vec3(n, what);
    return Color{v.x, v.y, v.z};
}

// externally tagged enum: a one-key map
std::pair<std::string, const Node *> variant(const Node &n, const std::string &what) {
    if (n.kind != Node::Map || n.map.size() != 1) bad(what + ": expected an externally tagged enum (one-key map)");
    return {n.map[0].first, &n.map[0].second};
}

MaterialData material(const Node &n, const std::string &what) {
    auto [tag, b] = variant(n, what);
    const std::string w = what + "." + tag;
    if (tag == "Matte")
        return MatteData{color(req(*b, "diffuse_color", w), w + ".diffuse_color"),
                         color(req(*b, "ambient_color", w), w + ".ambient_color"),
                         num(req(*b, "diffuse_coefficient", w), w + ".diffuse_coefficient")};
    if (tag == "Emissive") return EmissiveData{color(req(*b, "color", w), w + ".color"), num(req(*b, "power", w), w + ".power")};
    if (tag == "Reflective")
        return ReflectiveData{num(req(*b, "reflect_amount", w), w + ".reflect_amount"),
                              color(req(*b, "reflect_color", w), w + ".reflect_color")};
    if (tag == "GlossyReflective")
        return GlossyReflectiveData{num(req(*b, "reflect_amount", w), w + ".reflect_amount"),
                                    color(req(*b, "reflect_color", w), w + ".reflect_color"),
                                    num(req(*b, "reflect_exponent", w), w + ".reflect_exponent")};
    bad(what + ": unknown variant `" + tag + "`, expected one of `Matte`, `Emissive`, `Reflective`, `GlossyReflective`");
}

ShapeData shape(const Node &n, const std::string &what) {
    auto [tag, b] = variant(n, what);
    const std::string w = what + "." + tag;
    if (tag == "Sphere")
        return SphereData{vec3(req(*b, "center", w), w + ".center"), num(req(*b, "radius", w), w + ".radius"),
                          material(req(*b, "material", w), w + ".material"), boolean(req(*b, "invert", w), w + ".invert")};
    if (tag == "Plane")
        return PlaneData{vec3(req(*b, "point", w), w + ".point"), vec3(req(*b, "normal", w), w + ".normal"),
                         material(req(*b, "material", w), w + ".material")};
    bad(what + ": unknown variant `" + tag + "`, expected one of `Sphere`, `Plane`");
}

SceneData scene_from_node(const Node &d) {
    SceneData sd;
    const Node &name = req(d, "scene_name", "scene");
    if (name.kind != Node::Scalar) bad("scene.scene_name: expected a string");
    sd.scene_name = name.scalar;
    const Node &o = req(d, "output_settings", "scene");
    sd.output_settings.image_width = usize(req(o, "image_width", "output_settings"), "output_settings.image_width");
    sd.output_settings.image_height = usize(req(o, "image_height", "output_settings"), "output_settings.image_height");
    sd.output_settings.pixel_size = num(req(o, "pixel_size", "output_settings"), "output_settings.pixel_size");
    sd.background = color(req(d, "background", "scene"), "scene.background");
    const Node &shapes = req(d, "shapes", "scene");
    if (shapes.kind != Node::Seq && shapes.kind != Node::Null) bad("scene.shapes: expected a sequence");
    for (size_t i = 0; i < shapes.seq.size(); i++) sd.shapes.push_back(shape(shapes.seq[i], "shapes[" + std::to_string(i) + "]"));
    const Node &cs = req(d, "camera_settings", "scene");
    sd.camera_settings.eye = vec3(req(cs, "eye", "camera_settings"), "camera_settings.eye");
    sd.camera_settings.look_at = vec3(req(cs, "look_at", "camera_settings"), "camera_settings.look_at");
    sd.camera_settings.up = vec3(req(cs, "up", "camera_settings"), "camera_settings.up");
    const Node &cd = req(d, "camera_data", "scene");
    sd.camera_data.zoom_factor = num(req(cd, "zoom_factor", "camera_data"), "camera_data.zoom_factor");
    sd.camera_data.view_plane_distance = num(req(cd, "view_plane_distance", "camera_data"), "camera_data.view_plane_distance");
    sd.camera_data.focal_distance = num(req(cd, "focal_distance", "camera_data"), "camera_data.focal_distance");
    sd.camera_data.lens_radius = num(req(cd, "lens_radius", "camera_data"), "camera_data.lens_radius");
    return sd;
}

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

flux_material to_abi(const MaterialData &m) {
    flux_material o{};
    auto set3 = [](double *d, const Color &c) { d[0] = c.r; d[1] = c.g; d[2] = c.b; };
    if (auto p = std::get_if<MatteData>(&m)) {
        o.kind = FLUX_MAT_MATTE;
        set3(o.color, p->diffuse_color);
        set3(o.ambient, p->ambient_color);
        o.k = p->diffuse_coefficient;
    } else if (auto p = std::get_if<EmissiveData>(&m)) {
        o.kind = FLUX_MAT_EMISSIVE;
        set3(o.color, p->color);
        o.k = p->power;
    } else if (auto p = std::get_if<ReflectiveData>(&m)) {
        o.kind = FLUX_MAT_REFLECTIVE;
        set3(o.color, p->reflect_color);
        o.k = p->reflect_amount;
    } else if (auto p = std::get_if<GlossyReflectiveData>(&m)) {
        o.kind = FLUX_MAT_GLOSSY;
        set3(o.color, p->reflect_color);
        o.k = p->reflect_amount;
        o.exponent = p->reflect_exponent;
    }
    return o;
}

}  // namespace

SceneData scene_from_yaml_text(const std::string &text) {
    try {
        return scene_from_node(yaml_lite::parse(text));
    } catch (const yaml_lite::Error &e) {
        throw FluxError(FLUX_E_INVALID, e.what());
    }
}

SceneData scene_from_yaml_file(const std::string &path) {
    try {
        return scene_from_node(yaml_lite::parse_file(path));
    } catch (const yaml_lite::Error &e) {
        throw FluxError(FLUX_E_INVALID, path + ": " + e.what());
    }
}

AbiScene::AbiScene(const SceneData &sd) : name(sd.scene_name) {
    shapes.resize(sd.shapes.size());
    for (size_t i = 0; i < sd.shapes.size(); i++) {
        flux_shape &fs = shapes[i];
        fs = flux_shape{};
        if (auto s = std::get_if<SphereData>(&sd.shapes[i])) {
            fs.kind = FLUX_SHAPE_SPHERE;
            fs.p[0] = s->center.x; fs.p[1] = s->center.y; fs.p[2] = s->center.z;
            fs.radius = s->radius;
            fs.invert = s->invert ? 1 : 0;
            fs.material = to_abi(s->material);
        } else if (auto p = std::get_if<PlaneData>(&sd.shapes[i])) {
            fs.kind = FLUX_SHAPE_PLANE;
            fs.p[0] = p->point.x; fs.p[1] = p->point.y; fs.p[2] = p->point.z;
            fs.n[0] = p->normal.x; fs.n[1] = p->normal.y; fs.n[2] = p->normal.z;
            fs.material = to_abi(p->material);
        }
    }
    desc.scene_name = name.c_str();
    desc.image_width = sd.output_settings.image_width;
    desc.image_height = sd.output_settings.image_height;
    desc.pixel_size = sd.output_settings.pixel_size;
    desc.background[0] = sd.background.r; desc.background[1] = sd.background.g; desc.background[2] = sd.background.b;
    const CameraSettings &c = sd.camera_settings;
    desc.eye[0] = c.eye.x; desc.eye[1] = c.eye.y; desc.eye[2] = c.eye.z;
    desc.look_at[0] = c.look_at.x; desc.look_at[1] = c.look_at.y; desc.look_at[2] = c.look_at.z;
    desc.up[0] = c.up.x; desc.up[1] = c.up.y; desc.up[2] = c.up.z;
    desc.zoom_factor = sd.camera_data.zoom_factor;
    desc.view_plane_distance = sd.camera_data.view_plane_distance;
    desc.focal_distance = sd.camera_data.focal_distance;
    desc.lens_radius = sd.camera_data.lens_radius;
    desc.num_shapes = shapes.size();
    desc.shapes = shapes.empty() ? nullptr : shapes.data();
    desc.num_meshes = 0;
    desc.meshes = nullptr;
}

std::vector<WorkUnit> Job::work_units() const {
    const uint64_t h = scene_data.output_settings.image_height;
    int64_t n = flux_work_units(h, config.rows_per_work_unit, nullptr, 0);
    // the reference panics here: "Job row per work unit count invalid" (job.rs:67-70)
    if (n < 0) throw FluxError((int)n, flux_last_error());
    std::vector<flux_work_unit> raw((size_t)n);
    if (n > 0) flux_work_units(h, config.rows_per_work_unit, raw.data(), (uint64_t)n);
    std::vector<WorkUnit> us;
    for (const auto &u : raw) us.push_back(WorkUnit{(size_t)u.row_start, (size_t)u.row_end, id});
    return us;
}

// ---- GpuWorker: workers.rs:26-103 ---------------------------------------------------------------
GpuWorker::GpuWorker(int device, uint64_t seed)
    : device_(device), seed_(seed), sender_(std::make_shared<Channel<std::optional<WorkerRequest>>>()) {
    thread_ = std::thread([this] { run(); });
}

GpuWorker::~GpuWorker() { stop(); }

void GpuWorker::stop() {  // workers.rs:95-98: send(None), join
    if (stopped_) return;
    stopped_ = true;
    sender_->send(std::nullopt);
    if (thread_.joinable()) thread_.join();
}

std::atomic<int> GpuWorker::worker_failures_{0};
int GpuWorker::failures() { return worker_failures_.load(); }
void GpuWorker::note_failure() { worker_failures_.fetch_add(1); }

void GpuWorker::run() {
    // the worker's one-time set-up, before any job (LocalWorker::new builds its rayon pool here, workers.rs:27-38): the HIP runtime's
    // lazy initialisation on this device, so that a job's timer (manager.rs:145) sees the context's work and not the process's
    (void)flux_device_warmup(device_);
    // 'main: while let Ok(Some((job, recv_unit, send_result, wg))) = r.recv()   (workers.rs:43)
    for (;;) {
        auto msg = sender_->recv();
        if (!msg || !*msg) break;
        WorkerRequest req = std::move(**msg);
        // Scene::from_data + Camera::new (workers.rs:46-54)
        AbiScene abi(req.job->scene_data);
        flux_job_cfg cfg{req.job->config.sample_root, req.job->config.max_trace_depth,
                         req.job->config.rows_per_work_unit};
        flux_ctx *ctx = nullptr;
        if (flux_ctx_create(&abi.desc, &cfg, seed_, device_, &ctx) != FLUX_OK) {
            // the reference would panic in the worker thread (workers.rs:78); report and give the job up
            std::fprintf(stderr, "GpuWorker(device %d): %s\n", device_, flux_last_error());
            worker_failures_.fetch_add(1);
            req.wg->done();
            continue;
        }
        const size_t w = req.job->scene_data.output_settings.image_width;
        const size_t h = req.job->scene_data.output_settings.image_height;
        std::vector<double> buf;
        // while let Ok(unit) = recv_unit.recv()   (workers.rs:56)
        while (auto unit = req.recv_unit->recv()) {
            // A unit may come straight off a socket (flux_node): validate before sizing anything.  An inverted range is
            // what the reference renders as NO rows (`row_start..=row_end` is empty, trace.rs:62); rows past the image
            // would make its ImageBuilder index out of bounds (manager.rs:322-324) -- here the unit is refused.
            if (unit->row_end < unit->row_start) {
                RenderEvent ev;
                ev.kind = RenderEvent::RowsReady;
                ev.result.work_unit = *unit;
                req.send_result->send(std::move(ev));
                continue;
            }
            if (unit->row_end >= h) {
                std::fprintf(stderr, "GpuWorker(device %d): work unit rows [%zu,%zu] outside image height %zu\n", device_,
                             (size_t)unit->row_start, (size_t)unit->row_end, h);
                worker_failures_.fetch_add(1);
                break;
            }
            const size_t nrows = unit->row_end - unit->row_start + 1;
            buf.resize(nrows * w * 3);
            // camera.render(&scene, unit)   (workers.rs:60)
            if (flux_render_rows(ctx, unit->row_start, unit->row_end, buf.data()) != FLUX_OK) {
                // the reference would panic here and take the process down (workers.rs:78); a C++ worker thread records
                // the failure so that the front-end exits non-zero instead of reporting a frame with missing rows
                std::fprintf(stderr, "GpuWorker(device %d): %s\n", device_, flux_last_error());
                worker_failures_.fetch_add(1);
                break;
            }
            RenderEvent ev;
            ev.kind = RenderEvent::RowsReady;
            ev.result.work_unit = *unit;
            ev.result.rows.resize(nrows);
            for (size_t r = 0; r < nrows; r++) {
                auto &row = ev.result.rows[r];
                row.resize(w);
                const double *p = buf.data() + r * w * 3;
                for (size_t c = 0; c < w; c++) row[c] = Color{p[3 * c], p[3 * c + 1], p[3 * c + 2]};
            }
            req.send_result->send(std::move(ev));
        }
        flux_ctx_destroy(ctx);
        req.wg->done();  // drop(wg)   (workers.rs:74)
    }
}

// ---- MultiGpuWorker: the node's GPUs as one worker (flux_multi_*) ----------------------------------
MultiGpuWorker::MultiGpuWorker(std::vector<int> devices, uint64_t seed, int shard)
    : devices_(std::move(devices)), seed_(seed), shard_(shard), sender_(std::make_shared<Channel<std::optional<WorkerRequest>>>()) {
    thread_ = std::thread([this] { run(); });
}

MultiGpuWorker::~MultiGpuWorker() { stop(); }

void MultiGpuWorker::stop() {
    if (stopped_) return;
    stopped_ = true;
    sender_->send(std::nullopt);
    if (thread_.joinable()) thread_.join();
}

std::vector<double> MultiGpuWorker::last_timing() const {
    std::lock_guard<std::mutex> g(mu_);
    return timing_;
}

void MultiGpuWorker::run() {
    {   // one-time set-up per device, concurrently (see GpuWorker::run)
        std::vector<std::thread> th;
        for (int d : devices_) th.emplace_back([d] { (void)flux_device_warmup(d); });
        for (std::thread &t : th) t.join();
    }
    for (;;) {
        auto msg = sender_->recv();
        if (!msg || !*msg) break;
        WorkerRequest req = std::move(**msg);
        // Scene::from_data + Camera::new on every device (workers.rs:46-54), concurrently, + the communicators
        AbiScene abi(req.job->scene_data);
        flux_job_cfg cfg{req.job->config.sample_root, req.job->config.max_trace_depth, req.job->config.rows_per_work_unit};
        flux_multi *m = nullptr;
        if (flux_multi_create(&abi.desc, &cfg, seed_, devices_.data(), devices_.size(), shard_, &m) != FLUX_OK) {
            std::fprintf(stderr, "MultiGpuWorker(%zu devices): %s\n", devices_.size(), flux_last_error());
            GpuWorker::note_failure();
            req.wg->done();
            continue;
        }
        const size_t w = req.job->scene_data.output_settings.image_width;
        const size_t h = req.job->scene_data.output_settings.image_height;
        std::vector<double> frame;
        bool failed = false;
        while (auto unit = req.recv_unit->recv()) {
            if (unit->row_end < unit->row_start) {  // renders as no rows in the reference (trace.rs:62)
                RenderEvent ev;
                ev.kind = RenderEvent::RowsReady;
                ev.result.work_unit = *unit;
                req.send_result->send(std::move(ev));
                continue;
            }
            if (unit->row_end >= h) {
                std::fprintf(stderr, "MultiGpuWorker: work unit rows [%zu,%zu] outside image height %zu\n", (size_t)unit->row_start,
                             (size_t)unit->row_end, h);
                GpuWorker::note_failure();
                break;
            }
            if (frame.empty() && !failed) {  // the first unit: camera.render for the whole image, on all devices (workers.rs:60)
                frame.resize(w * h * 3);
                if (flux_multi_render_frame(m, frame.data()) != FLUX_OK) {
                    std::fprintf(stderr, "MultiGpuWorker(%zu devices): %s\n", devices_.size(), flux_last_error());
                    GpuWorker::note_failure();
                    failed = true;
                }
                double t[FLUX_MULTI_TIMING_WORDS];
                if (!failed && flux_multi_timing(m, t) == FLUX_OK) {
                    std::lock_guard<std::mutex> g(mu_);
                    timing_.assign(t, t + FLUX_MULTI_TIMING_WORDS);
                }
            }
            if (failed) break;
            const size_t nrows = unit->row_end - unit->row_start + 1;
            RenderEvent ev;
            ev.kind = RenderEvent::RowsReady;
            ev.result.work_unit = *unit;
            ev.result.rows.resize(nrows);
            for (size_t r = 0; r < nrows; r++) {
                auto &row = ev.result.rows[r];
                row.resize(w);
                const double *p = frame.data() + (unit->row_start + r) * w * 3;
                for (size_t c = 0; c < w; c++) row[c] = Color{p[3 * c], p[3 * c + 1], p[3 * c + 2]};
            }
            req.send_result->send(std::move(ev));
        }
        flux_multi_destroy(m);
        req.wg->done();
    }
}

// ---- ImageBuilder: manager.rs:278-363 -----------------------------------------------------------
ImageBuilder::ImageBuilder() : sender_(std::make_shared<Channel<std::optional<RenderEvent>>>()) {
    thread_ = std::thread([this] { run(); });
}
ImageBuilder::~ImageBuilder() { stop(); }
void ImageBuilder::stop() {
    if (stopped_) return;
    stopped_ = true;
    sender_->send(std::nullopt);
    if (thread_.joinable()) thread_.join();
}

void ImageBuilder::run() {
    auto first = sender_->recv();
    if (!first || !*first || (*first)->kind != RenderEvent::ImageInfo) return;  // manager.rs:291-297
    const std::string scene_name = (*first)->scene_name;
    const size_t width = (*first)->width, height = (*first)->height;
    auto second = sender_->recv();
    if (!second || !*second || (*second)->kind != RenderEvent::RenderingStarted) return;  // manager.rs:301-307
    const double start_time = (*second)->time_s;
    std::vector<double> img(width * height * 3, 0.0);
    std::vector<uint8_t> present(height, 0);
    for (;;) {
        auto m = sender_->recv();
        if (!m || !*m) break;
        RenderEvent &ev = **m;
        if (ev.kind == RenderEvent::RowsReady) {  // manager.rs:316-324
            for (size_t i = 0; i < ev.result.rows.size(); i++) {
                const size_t r = i + ev.result.work_unit.row_start;
                if (r >= height) continue;
                present[r] = 1;
                const auto &row = ev.result.rows[i];
                for (size_t c = 0; c < row.size() && c < width; c++) {
                    double *p = &img[(r * width + c) * 3];
                    p[0] = row[c].r; p[1] = row[c].g; p[2] = row[c].b;
                }
            }
        } else if (ev.kind == RenderEvent::RenderingFinished) {  // manager.rs:326-335
            total_time_s = ev.time_s - start_time;
            std::printf("rendering finished, total time %.6fs\n", total_time_s);
            written_path = output_dir + "/" + scene_name + ".ppm";
            if (flux_write_ppm(written_path.c_str(), img.data(), width, height, present.data()) != FLUX_OK)
                std::fprintf(stderr, "ImageBuilder: %s\n", flux_last_error());
        } else {
            return;  // unexpected message (manager.rs:336-339)
        }
    }
}

// ---- the slice of RenderManager the CLI needs: manager.rs:83-186 --------------------------------
void run_job(const Job &job, const std::vector<WorkerHandle> &workers,
             const std::shared_ptr<Channel<std::optional<RenderEvent>>> &events) {
    if (workers.empty()) throw FluxError(FLUX_E_INVALID, "RenderManager::new: must provide at least one worker handle");
    RenderEvent info;
    info.kind = RenderEvent::ImageInfo;  // manager.rs:86-98
    info.scene_name = job.scene_data.scene_name;
    info.width = job.scene_data.output_settings.image_width;
    info.height = job.scene_data.output_settings.image_height;
    events->send(info);
    auto units = std::make_shared<Channel<WorkUnit>>();  // manager.rs:100 (shared by every worker)
    for (const WorkUnit &u : job.work_units()) units->send(u);
    units->close();
    RenderEvent started;
    started.kind = RenderEvent::RenderingStarted;  // manager.rs:145-154: before the job reaches the workers
    started.job_id = job.id;
    started.time_s = now_s();
    events->send(started);
    auto wg = std::make_shared<WaitGroup>();
    auto shared_job = std::make_shared<Job>(job);
    for (const WorkerHandle &w : workers) {  // manager.rs:156-162
        wg->add();
        w.send(shared_job, units, events, wg);
    }
    wg->wait();  // manager.rs:166
    RenderEvent fin;
    fin.kind = RenderEvent::RenderingFinished;  // manager.rs:170-185
    fin.time_s = now_s();
    events->send(fin);
}

// ---- JobIDAllocator / RenderManager: job.rs:14-34, manager.rs:72-219 ------------------------------
JobIDAllocator::JobIDAllocator() {
    std::random_device rd;  // rand::thread_rng().gen() (job.rs:21-24)
    allocator_id_ = ((size_t)rd() << 32) ^ (size_t)rd();
}

RenderManager::RenderManager(std::vector<WorkerHandle> workers)
    : workers_(std::move(workers)), queue_(std::make_shared<Channel<std::optional<ScheduledJob>>>()) {
    if (workers_.empty())
        throw FluxError(FLUX_E_INVALID, "RenderManager::new: must provide at least one worker handle");
    thread_ = std::thread([this] { run(); });
}

RenderManager::~RenderManager() { stop(); }

void RenderManager::stop() {  // manager.rs:215-218
    if (stopped_) return;
    stopped_ = true;
    queue_->send(std::nullopt);
    if (thread_.joinable()) thread_.join();
}

JobHandle RenderManager::schedule_job(const SceneData &scene_data, const JobConfiguration &config,
                                      std::shared_ptr<Channel<std::optional<RenderEvent>>> result_sender) {
    ScheduledJob sj;  // manager.rs:198-213
    sj.job.id = ids_.next_id();
    sj.job.scene_data = scene_data;
    sj.job.config = config;
    sj.notify_done = std::make_shared<Channel<int>>();
    sj.notify_cancel = std::make_shared<Channel<int>>();
    sj.result_sender = std::move(result_sender);
    JobHandle h;
    h.job_id = sj.job.id;
    h.waiter_ = sj.notify_done;
    h.canceller_ = sj.notify_cancel;
    queue_->send(std::move(sj));
    return h;
}

void RenderManager::run() {
    // while let Ok(Some((job, notify_done, notify_cancel, result_sender))) = r.recv()   (manager.rs:83)
    for (;;) {
        auto msg = queue_->recv();
        if (!msg || !*msg) break;
        ScheduledJob sj = std::move(**msg);
        const Job &job = sj.job;
        RenderEvent info;
        info.kind = RenderEvent::ImageInfo;  // manager.rs:86-98
        info.scene_name = job.scene_data.scene_name;
        info.width = job.scene_data.output_settings.image_width;
        info.height = job.scene_data.output_settings.image_height;
        if (!sj.result_sender->send(info)) continue;

        auto ws = std::make_shared<Channel<WorkUnit>>(1);  // bounded(1), manager.rs:100
        auto wg = std::make_shared<WaitGroup>();
        std::vector<WorkUnit> units;
        try {
            units = job.work_units();  // the reference panics on rows_per_work_unit == 0 (job.rs:67-70)
        } catch (const FluxError &e) {
            std::fprintf(stderr, "RenderManager: %s\n", e.what());
            sj.notify_done->send(0);
            continue;
        }
        auto wu_queue = std::make_shared<CancellableWorkUnits>(std::move(units));
        // cancel listener (manager.rs:105-116): ends when a cancel arrives or the job's handle side closes
        auto cancel_ch = sj.notify_cancel;
        std::thread cancel_listener([cancel_ch, wu_queue] {
            if (cancel_ch->recv()) wu_queue->cancel();
        });
        // work-unit producer (manager.rs:118-141): blocks in send() until a worker takes the unit
        std::thread producer([ws, wu_queue] {
            while (auto u = wu_queue->next())
                if (!ws->send(*u)) break;
            ws->close();  // dropping the Sender: workers' recv() then fails and they finish the job
        });

        RenderEvent started;
        started.kind = RenderEvent::RenderingStarted;  // manager.rs:145-154
        started.job_id = job.id;
        started.time_s = now_s();
        const bool started_ok = sj.result_sender->send(started);
        if (started_ok) {
            auto shared_job = std::make_shared<Job>(job);
            for (const WorkerHandle &w : workers_) {  // manager.rs:156-162
                wg->add();
                w.send(shared_job, ws, sj.result_sender, wg);
            }
            wg->wait();  // manager.rs:166
            // every worker has dropped its Receiver: a producer still blocked in send() must fail
            // (crossbeam: send on a channel with no receivers errors, manager.rs:131-136)
            wu_queue->cancel();
            ws->close();
        } else {
            wu_queue->cancel();
            ws->close();
        }
        producer.join();
        cancel_ch->close();  // the reference leaves the listener blocked until the JobHandle drops
        cancel_listener.join();
        if (!started_ok) continue;
        RenderEvent fin;
        fin.kind = RenderEvent::RenderingFinished;  // manager.rs:170-177
        fin.time_s = now_s();
        if (!sj.result_sender->send(fin)) continue;
        sj.notify_done->send(0);  // manager.rs:179-185
    }
}

}  // namespace flux_host
