// flux_host.hpp -- C++ host side above the C ABI, mirroring the reference's plain-data types and
// its worker interface (same names, same argument meaning), so a GPU sits where a LocalWorker does:
//
//   SceneData & friends        fluxcore/src/scene.rs:12-74, shapes.rs:15-81, color.rs:8-16
//   JobConfiguration, WorkUnit, Job::work_units      fluxcore/src/job.rs:40-88
//   RenderEvent, WorkUnitResult, WorkerInfo, trait Worker, WorkerHandle
//                                                    fluxcore/src/manager.rs:16-52,221-236
//   LocalWorker job loop -> GpuWorker                fluxcore/src/workers.rs:26-103
//   ImageBuilder, Image::write                       fluxcore/src/manager.rs:278-363, image.rs:43-61
//
// The reference is Rust; this image has no Rust toolchain, so the mirror is C++ (std::thread and a
// small channel instead of crossbeam).  All rendering goes through include/flux_abi.h.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <variant>
#include <vector>

#include "../../include/flux_abi.h"

namespace flux_host {

struct FluxError : std::runtime_error {
    int code;
    FluxError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// ---- scene description (serde schema of the reference) ---------------------------------------
struct Color { double r = 0, g = 0, b = 0; };
struct Vec3 { double x = 0, y = 0, z = 0; };

struct MatteData { Color diffuse_color, ambient_color; double diffuse_coefficient = 0; };
struct EmissiveData { Color color; double power = 0; };
struct ReflectiveData { double reflect_amount = 0; Color reflect_color; };
struct GlossyReflectiveData { double reflect_amount = 0; Color reflect_color; double reflect_exponent = 0; };
using MaterialData = std::variant<MatteData, EmissiveData, ReflectiveData, GlossyReflectiveData>;

struct SphereData { Vec3 center; double radius = 0; MaterialData material; bool invert = false; };
struct PlaneData { Vec3 point, normal; MaterialData material; };
using ShapeData = std::variant<SphereData, PlaneData>;

struct CameraSettings { Vec3 eye, look_at, up; };
struct CameraData { double zoom_factor = 1, view_plane_distance = 0, focal_distance = 0, lens_radius = 0; };
struct OutputSettings { size_t image_width = 0, image_height = 0; double pixel_size = 0; };

struct SceneData {
    std::string scene_name;
    OutputSettings output_settings;
    Color background;
    std::vector<ShapeData> shapes;
    CameraSettings camera_settings;
    CameraData camera_data;
};

// serde_yaml::from_reader (flux/src/main.rs:27-29); throws FluxError(FLUX_E_INVALID) with a
// serde-style message on missing fields / unknown variants.
SceneData scene_from_yaml_file(const std::string &path);
SceneData scene_from_yaml_text(const std::string &text);

// ---- jobs ---------------------------------------------------------------------------------------
struct JobID { size_t allocator_id = 0, id = 0; };
struct JobConfiguration { size_t sample_root = 1, max_trace_depth = 5, rows_per_work_unit = 50; };
struct WorkUnit { size_t row_start = 0, row_end = 0; JobID job_id; };

struct Job {
    JobID id;
    SceneData scene_data;
    JobConfiguration config;
    std::vector<WorkUnit> work_units() const;  // job.rs:65-88 (via flux_work_units)
};

struct WorkUnitResult {
    WorkUnit work_unit;
    std::vector<std::vector<Color>> rows;
};

struct RenderEvent {  // manager.rs:16-22
    enum Kind { RenderingStarted, ImageInfo, RowsReady, RenderingFinished } kind = RowsReady;
    JobID job_id;
    double time_s = 0;  // start_time / end_time (seconds on a steady clock)
    std::string scene_name;
    size_t width = 0, height = 0;
    WorkUnitResult result;
};

struct WorkerInfo { size_t num_threads = 0; };  // manager.rs:221-230

// unbounded MPMC channel with close(), enough of crossbeam::channel for the job loop
template <typename T>
class Channel {
public:
    // capacity 0 = unbounded; otherwise crossbeam's bounded(cap): send blocks while the queue is full
    explicit Channel(size_t capacity = 0) : cap_(capacity) {}
    // returns false if the channel was closed before the value could be queued (crossbeam: send() -> Err)
    bool send(T v) {
        {
            std::unique_lock<std::mutex> l(mu_);
            if (cap_) space_.wait(l, [&] { return q_.size() < cap_ || closed_; });
            if (closed_) return false;
            q_.push_back(std::move(v));
        }
        cv_.notify_one();
        return true;
    }
    // returns nullopt once closed and drained (crossbeam: recv() -> Err)
    std::optional<T> recv() {
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return std::nullopt;
        T v = std::move(q_.front());
        q_.pop_front();
        l.unlock();
        space_.notify_one();
        return v;
    }
    void close() {
        { std::lock_guard<std::mutex> g(mu_); closed_ = true; }
        cv_.notify_all();
        space_.notify_all();
    }
private:
    size_t cap_ = 0;
    std::mutex mu_;
    std::condition_variable cv_, space_;
    std::deque<T> q_;
    bool closed_ = false;
};

struct WaitGroup {  // crossbeam::sync::WaitGroup: done() once per clone, wait() for all
    void add() { std::lock_guard<std::mutex> g(mu); n++; }
    void done() { { std::lock_guard<std::mutex> g(mu); n--; } cv.notify_all(); }
    void wait() { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return n == 0; }); }
    std::mutex mu; std::condition_variable cv; int n = 0;
};

// WorkerRequest = Option<(Box<Job>, Receiver<WorkUnit>, Sender<Option<RenderEvent>>, WaitGroup)> (manager.rs:36)
struct WorkerRequest {
    std::shared_ptr<Job> job;
    std::shared_ptr<Channel<WorkUnit>> recv_unit;
    std::shared_ptr<Channel<std::optional<RenderEvent>>> send_result;
    std::shared_ptr<WaitGroup> wg;
};

class WorkerHandle {  // manager.rs:38-52
public:
    explicit WorkerHandle(std::shared_ptr<Channel<std::optional<WorkerRequest>>> s) : sender_(std::move(s)) {}
    void send(std::shared_ptr<Job> j, std::shared_ptr<Channel<WorkUnit>> r,
              std::shared_ptr<Channel<std::optional<RenderEvent>>> s, std::shared_ptr<WaitGroup> wg) const {
        sender_->send(WorkerRequest{std::move(j), std::move(r), std::move(s), std::move(wg)});
    }
private:
    std::shared_ptr<Channel<std::optional<WorkerRequest>>> sender_;
};

class Worker {  // trait Worker, manager.rs:232-236
public:
    virtual ~Worker() = default;
    virtual WorkerHandle handle() const = 0;
    virtual void stop() = 0;
    virtual WorkerInfo info() const = 0;
};

// The GPU sibling of LocalWorker (workers.rs:26-103): one thread; per job it builds the context
// (Scene::from_data + Camera::new) once, then renders the work units it pulls from the shared channel
// and emits RenderEvent::RowsReady.
class GpuWorker : public Worker {
public:
    GpuWorker(int device, uint64_t seed);
    ~GpuWorker() override;
    WorkerHandle handle() const override { return WorkerHandle(sender_); }
    void stop() override;
    WorkerInfo info() const override { return WorkerInfo{1}; }
    // jobs / work units any GpuWorker of this process had to give up (context creation or a render call failed, or a
    // unit was out of range): the reference panics there (workers.rs:78); the front-ends exit non-zero when this is > 0
    static int failures();
    static void note_failure();  // (MultiGpuWorker books its failures in the same counter)
private:
    void run();
    static std::atomic<int> worker_failures_;
    int device_;
    uint64_t seed_;
    std::shared_ptr<Channel<std::optional<WorkerRequest>>> sender_;
    std::thread thread_;
    bool stopped_ = false;
};

// The GPUs of this node as ONE worker behind the same trait: per job it creates a flux_multi (include/flux_abi.h: one context per
// device holding its share of the sample tables, created concurrently) and, when the first work unit arrives, renders the WHOLE frame
// with one launch per device and one RCCL all-gather; every unit it pulls from the shared channel (workers.rs:56) is then answered
// from that frame.  Against one GpuWorker per device -- each building ALL tables and pulling fifty-row units (12 units over 8 GPUs: two
// rounds, 75 % at best), each unit copied back through the host -- the frame costs one balanced launch and one device-side gather.
// It renders every row itself, so it is for a pool WITHOUT other workers (flux_cli.cpp selects it only then): rows a NetworkWorker
// renders as well would be work done twice.
class MultiGpuWorker : public Worker {
public:
    MultiGpuWorker(std::vector<int> devices, uint64_t seed, int shard = FLUX_SHARD_AUTO);
    ~MultiGpuWorker() override;
    WorkerHandle handle() const override { return WorkerHandle(sender_); }
    void stop() override;
    WorkerInfo info() const override { return WorkerInfo{devices_.size()}; }
    // ms of the most recent job: flux_multi_timing's words (create, slowest context, communicators, frame, kernel, gather, reassembly, copy)
    std::vector<double> last_timing() const;
private:
    void run();
    std::vector<int> devices_;
    uint64_t seed_;
    int shard_;
    std::shared_ptr<Channel<std::optional<WorkerRequest>>> sender_;
    std::thread thread_;
    bool stopped_ = false;
    mutable std::mutex mu_;
    std::vector<double> timing_;
};

// ImageBuilder (manager.rs:278-363): assembles RowsReady rows, prints the total time and writes
// "<scene_name>.ppm" on RenderingFinished; rows never received are written as zeros (image.rs:55-59).
class ImageBuilder {
public:
    ImageBuilder();
    ~ImageBuilder();
    std::shared_ptr<Channel<std::optional<RenderEvent>>> sender() const { return sender_; }
    void stop();
    std::string output_dir = ".";
    double total_time_s = 0;
    std::string written_path;
private:
    void run();
    std::shared_ptr<Channel<std::optional<RenderEvent>>> sender_;
    std::thread thread_;
    bool stopped_ = false;
};

// One job on the calling thread, no queue and no cancellation (what the CLI's single render needs):
// ImageInfo, RenderingStarted, all work units through one shared channel, wait, RenderingFinished.
void run_job(const Job &job, const std::vector<WorkerHandle> &workers,
             const std::shared_ptr<Channel<std::optional<RenderEvent>>> &events);

// JobIDAllocator (job.rs:14-34): allocator_id is random per allocator, ids count from 0.
class JobIDAllocator {
public:
    JobIDAllocator();
    JobID next_id() { return JobID{allocator_id_, next_++}; }
private:
    size_t allocator_id_ = 0, next_ = 0;
};

// CancellableIterator (manager.rs:365-393) over a job's work units: after cancel() next() yields nothing.
class CancellableWorkUnits {
public:
    explicit CancellableWorkUnits(std::vector<WorkUnit> items) : items_(std::move(items)) {}
    void cancel() { std::lock_guard<std::mutex> g(mu_); cancelled_ = true; }
    std::optional<WorkUnit> next() {
        std::lock_guard<std::mutex> g(mu_);
        if (cancelled_ || pos_ >= items_.size()) return std::nullopt;
        return items_[pos_++];
    }
private:
    std::mutex mu_;
    std::vector<WorkUnit> items_;
    size_t pos_ = 0;
    bool cancelled_ = false;
};

// JobHandle (manager.rs:54-70): wait() blocks until the manager has sent RenderingFinished for the job;
// cancel() stops the hand-out of further work units (units already pulled by a worker complete).
class JobHandle {
public:
    JobID job_id;
    void wait() const { waiter_->recv(); }
    void cancel() const { canceller_->send(0); }
private:
    friend class RenderManager;
    std::shared_ptr<Channel<int>> waiter_, canceller_;
};

// RenderManager (manager.rs:30-219): a thread that takes scheduled jobs one at a time; per job it sends
// ImageInfo, starts a producer thread that feeds the job's work units into ONE bounded(1) channel shared by
// all workers (dynamic load balancing: a worker pulls its next unit when it is free) and a cancel-listener
// thread, sends RenderingStarted (its timestamp precedes the fan-out, manager.rs:145), hands the job to
// every worker, waits for the WaitGroup, sends RenderingFinished and wakes JobHandle::wait().
class RenderManager {
public:
    explicit RenderManager(std::vector<WorkerHandle> workers);  // throws on an empty vector (manager.rs:74-76)
    ~RenderManager();
    JobHandle schedule_job(const SceneData &scene_data, const JobConfiguration &config,
                           std::shared_ptr<Channel<std::optional<RenderEvent>>> result_sender);
    void stop();
private:
    struct ScheduledJob {
        Job job;
        std::shared_ptr<Channel<int>> notify_done, notify_cancel;
        std::shared_ptr<Channel<std::optional<RenderEvent>>> result_sender;
    };
    void run();
    std::vector<WorkerHandle> workers_;
    JobIDAllocator ids_;
    std::shared_ptr<Channel<std::optional<ScheduledJob>>> queue_;
    std::thread thread_;
    bool stopped_ = false;
};

// SceneData -> flux_scene_desc (+ the shape storage it points to)
struct AbiScene {
    std::vector<flux_shape> shapes;
    std::string name;
    flux_scene_desc desc{};
    explicit AbiScene(const SceneData &sd);
};

}  // namespace flux_host
