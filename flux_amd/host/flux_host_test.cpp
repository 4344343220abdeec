// flux_host_test.cpp -- CPU-only checks of the C++ host layer (run by tests/test_host_cpp.py).
// Prints one "ok <name>" line per check; exits non-zero on the first failure.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <string>

#include "flux_host.hpp"
#include "yaml_lite.hpp"

using namespace flux_host;

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

#define CHECK_VOID(cond)                                                         \
    do {                                                                         \
        if (!(cond)) std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
    } while (0)

static bool throws(const std::string &yaml, const char *needle) {
    try {
        scene_from_yaml_text(yaml);
    } catch (const FluxError &e) {
        return std::string(e.what()).find(needle) != std::string::npos;
    }
    return false;
}

// A Worker that renders nothing: per work unit it sleeps `ms_per_unit` and reports rows filled with its id
// (the scheduler under test never looks inside the rows).  Same job loop shape as workers.rs:43-75.
class FakeWorker : public Worker {
public:
    FakeWorker(int id, int ms_per_unit)
        : id_(id), ms_(ms_per_unit), sender_(std::make_shared<Channel<std::optional<WorkerRequest>>>()) {
        thread_ = std::thread([this] {
            for (;;) {
                auto msg = sender_->recv();
                if (!msg || !*msg) break;
                WorkerRequest req = std::move(**msg);
                jobs_seen++;
                while (auto unit = req.recv_unit->recv()) {
                    std::this_thread::sleep_for(std::chrono::milliseconds(ms_));
                    RenderEvent ev;
                    ev.kind = RenderEvent::RowsReady;
                    ev.result.work_unit = *unit;
                    ev.result.rows.assign(unit->row_end - unit->row_start + 1,
                                          std::vector<Color>(req.job->scene_data.output_settings.image_width,
                                                             Color{(double)id_, 0, 0}));
                    units_done++;
                    req.send_result->send(std::move(ev));
                }
                req.wg->done();
            }
        });
    }
    ~FakeWorker() override { stop(); }
    WorkerHandle handle() const override { return WorkerHandle(sender_); }
    void stop() override {
        if (stopped_) return;
        stopped_ = true;
        sender_->send(std::nullopt);
        thread_.join();
    }
    WorkerInfo info() const override { return WorkerInfo{1}; }
    std::atomic<int> units_done{0}, jobs_seen{0};
private:
    int id_, ms_;
    std::shared_ptr<Channel<std::optional<WorkerRequest>>> sender_;
    std::thread thread_;
    bool stopped_ = false;
};

static std::vector<RenderEvent> drain(const std::shared_ptr<Channel<std::optional<RenderEvent>>> &ch) {
    ch->close();
    std::vector<RenderEvent> out;
    while (auto m = ch->recv())
        if (*m) out.push_back(**m);
    return out;
}

int main(int argc, char **argv) {
    const std::string scenes = argc > 1 ? argv[1] : "scenes";
    {
        SceneData d1 = scene_from_yaml_file(scenes + "/demo1.yml");
        CHECK(d1.scene_name == "demo1" && d1.shapes.size() == 6);
        CHECK(d1.output_settings.image_width == 800 && d1.output_settings.image_height == 600);
        CHECK(d1.camera_settings.eye.x == 2.5 && d1.camera_settings.eye.z == -9.0 && d1.camera_data.lens_radius == 0.0);
        auto *s0 = std::get_if<SphereData>(&d1.shapes[0]);
        CHECK(s0 && s0->invert && s0->radius == 100.0);
        auto *e0 = std::get_if<EmissiveData>(&s0->material);
        CHECK(e0 && e0->power == 1.0 && e0->color.g == 0.9686);
        auto *p5 = std::get_if<PlaneData>(&d1.shapes[5]);
        CHECK(p5 && p5->normal.y == 1.0);
        auto *g3 = std::get_if<GlossyReflectiveData>(&std::get<SphereData>(d1.shapes[3]).material);
        CHECK(g3 && g3->reflect_exponent == 100000.0 && g3->reflect_amount == 0.9);
        std::puts("ok demo1");
    }
    {
        SceneData d2 = scene_from_yaml_file(scenes + "/demo2.yml");  // anchors/aliases, unknown top-level keys
        CHECK(d2.scene_name == "demo2" && d2.shapes.size() == 13 && d2.camera_data.lens_radius == 0.09);
        auto *m2 = std::get_if<GlossyReflectiveData>(&std::get<SphereData>(d2.shapes[2]).material);
        auto *m5 = std::get_if<GlossyReflectiveData>(&std::get<SphereData>(d2.shapes[5]).material);
        CHECK(m2 && m5 && m2->reflect_exponent == 10000.0 && m5->reflect_exponent == 10000.0 && m2->reflect_color.b == 1.0);
        auto *m4 = std::get_if<GlossyReflectiveData>(&std::get<SphereData>(d2.shapes[4]).material);
        CHECK(m4 && m4->reflect_exponent == 10.0);
        CHECK(std::get_if<PlaneData>(&d2.shapes[12]) != nullptr);
        AbiScene abi(d2);
        CHECK(abi.desc.num_shapes == 13 && abi.shapes[1].radius == 5.0 && abi.shapes[1].material.k == 10.0);
        CHECK(abi.shapes[12].kind == FLUX_SHAPE_PLANE && abi.shapes[2].material.kind == FLUX_MAT_GLOSSY);
        std::puts("ok demo2");
    }
    {
        const std::string base =
            "scene_name: t\noutput_settings: {}\n";
        CHECK(throws(base, "flow mappings"));
        const std::string good =
            "scene_name: t\n"
            "camera_settings:\n  eye: [0, 0, -5]\n  look_at: [0, 0, 0]\n  up: [0, 1, 0]\n"
            "camera_data:\n  zoom_factor: 1.0\n  view_plane_distance: 500.0\n  focal_distance: 10.0\n  lens_radius: 0.0\n"
            "output_settings:\n  image_width: 8\n  image_height: 6\n  pixel_size: 0.5\n"
            "background: [0, 0, 0]\n"
            "shapes:\n"
            "- Sphere:   # sequence at the key's own indentation\n"
            "    center: [0, 0, 0]\n    radius: 1\n    invert: false\n"
            "    material:\n      Reflective:\n        reflect_amount: 0.5\n        reflect_color: [1, 1, 1]\n";
        SceneData sd = scene_from_yaml_text(good);
        CHECK(sd.shapes.size() == 1 && std::get_if<ReflectiveData>(&std::get<SphereData>(sd.shapes[0]).material));
        std::string missing = good;
        missing.replace(missing.find("    invert: false\n"), 18, "");
        CHECK(throws(missing, "missing field `invert`"));
        std::string unknown = good;
        unknown.replace(unknown.find("- Sphere:"), 9, "- Torus: ");
        CHECK(throws(unknown, "unknown variant `Torus`"));
        std::string badmat = good;
        badmat.replace(badmat.find("Reflective:"), 11, "Glass:     ");
        CHECK(throws(badmat, "unknown variant `Glass`"));
        std::string badnum = good;
        badnum.replace(badnum.find("radius: 1"), 9, "radius: x");
        CHECK(throws(badnum, "expected a number"));
        CHECK(throws("scene_name: t\n\tbad: 1\n", "tab"));
        CHECK(throws("a: *nope\n", "unknown alias"));
        std::puts("ok yaml errors");
    }
    {
        Job job;
        job.scene_data.output_settings.image_height = 600;
        job.config.rows_per_work_unit = 50;
        auto us = job.work_units();
        CHECK(us.size() == 12 && us[0].row_start == 0 && us[0].row_end == 49 && us[11].row_end == 599);
        job.config.rows_per_work_unit = 1;
        CHECK(job.work_units().size() == 599);  // job.rs:74 quirk
        job.config.rows_per_work_unit = 0;
        bool threw = false;
        try {
            job.work_units();
        } catch (const FluxError &) {
            threw = true;
        }
        CHECK(threw);
        std::puts("ok work_units");
    }
    {
        Channel<int> ch;
        ch.send(1);
        ch.send(2);
        ch.close();
        CHECK(*ch.recv() == 1 && *ch.recv() == 2 && !ch.recv());
        WaitGroup wg;
        wg.add();
        std::thread t([&] { wg.done(); });
        wg.wait();
        t.join();
        std::puts("ok channel");
    }
    {
        // ImageBuilder: missing rows are zero-padded (image.rs:55-59); events in reference order
        ImageBuilder ib;
        ib.output_dir = argc > 2 ? argv[2] : "/tmp";
        auto ev = ib.sender();
        RenderEvent a;
        a.kind = RenderEvent::ImageInfo;
        a.scene_name = "host_test";
        a.width = 2;
        a.height = 3;
        ev->send(a);
        RenderEvent b;
        b.kind = RenderEvent::RenderingStarted;
        b.time_s = 1.0;
        ev->send(b);
        RenderEvent c;
        c.kind = RenderEvent::RowsReady;
        c.result.work_unit = WorkUnit{2, 2, {}};
        c.result.rows = {{Color{1.0, 0.5, 0.0}, Color{0.25, 0.25, 0.25}}};
        ev->send(c);
        RenderEvent d;
        d.kind = RenderEvent::RenderingFinished;
        d.time_s = 3.5;
        ev->send(d);
        ib.stop();
        CHECK(std::fabs(ib.total_time_s - 2.5) < 1e-12);
        FILE *f = std::fopen(ib.written_path.c_str(), "r");
        CHECK(f != nullptr);
        char buf[256];
        std::string all;
        while (std::fgets(buf, sizeof buf, f)) all += buf;
        std::fclose(f);
        CHECK(all == "P3\n2 3\n65535\n0 0 0\n0 0 0\n0 0 0\n0 0 0\n65535 32767 0\n16383 16383 16383\n");
        std::puts("ok image_builder");
    }
    {
        // bounded(1): the second send blocks until the first value is taken; close() fails pending sends
        Channel<int> ch(1);
        CHECK(ch.send(1));
        std::atomic<bool> second_sent{false};
        std::thread t([&] { ch.send(2); second_sent = true; });
        std::this_thread::sleep_for(std::chrono::milliseconds(30));
        CHECK(!second_sent);
        CHECK(*ch.recv() == 1);
        t.join();
        CHECK(second_sent && *ch.recv() == 2);
        CHECK(ch.send(3));
        std::thread t2([&] { CHECK_VOID(!ch.send(4)); });
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
        ch.close();
        t2.join();
        JobIDAllocator a;
        JobID i0 = a.next_id(), i1 = a.next_id();
        CHECK(i0.id == 0 && i1.id == 1 && i0.allocator_id == i1.allocator_id);
        CancellableWorkUnits it({WorkUnit{0, 0, {}}, WorkUnit{1, 1, {}}, WorkUnit{2, 2, {}}});
        CHECK(it.next()->row_start == 0);
        it.cancel();
        CHECK(!it.next());
        std::puts("ok bounded channel");
    }
    {
        // RenderManager (manager.rs:72-219): event order, every unit exactly once across two workers,
        // jobs run one after another, ids count up within one allocator
        bool threw = false;
        try {
            RenderManager none({});
        } catch (const FluxError &) {
            threw = true;
        }
        CHECK(threw);
        FakeWorker w0(0, 1), w1(1, 2);
        RenderManager mgr({w0.handle(), w1.handle()});
        SceneData sd;
        sd.scene_name = "sched";
        sd.output_settings.image_width = 4;
        sd.output_settings.image_height = 600;
        auto ev1 = std::make_shared<Channel<std::optional<RenderEvent>>>();
        auto ev2 = std::make_shared<Channel<std::optional<RenderEvent>>>();
        JobHandle h1 = mgr.schedule_job(sd, JobConfiguration{1, 5, 50}, ev1);
        JobHandle h2 = mgr.schedule_job(sd, JobConfiguration{1, 5, 7}, ev2);
        h1.wait();
        h2.wait();
        CHECK(h1.job_id.id == 0 && h2.job_id.id == 1 && h1.job_id.allocator_id == h2.job_id.allocator_id);
        auto e1 = drain(ev1), e2 = drain(ev2);
        CHECK(e1.size() == 2 + 12 + 1);
        CHECK(e1[0].kind == RenderEvent::ImageInfo && e1[0].scene_name == "sched" && e1[0].height == 600);
        CHECK(e1[1].kind == RenderEvent::RenderingStarted && e1[1].job_id.id == 0);
        CHECK(e1.back().kind == RenderEvent::RenderingFinished && e1.back().time_s >= e1[1].time_s);
        std::set<size_t> starts;
        for (size_t k = 2; k + 1 < e1.size(); k++) {
            CHECK(e1[k].kind == RenderEvent::RowsReady);
            const WorkUnit &u = e1[k].result.work_unit;
            CHECK(u.row_end == u.row_start + 49 && e1[k].result.rows.size() == 50 && u.job_id.id == 0);
            CHECK(starts.insert(u.row_start).second);  // each unit once
        }
        CHECK(starts.size() == 12 && *starts.begin() == 0 && *starts.rbegin() == 550);
        CHECK(e2.size() == 2 + 86 + 1);  // 600 rows in units of 7: 85 full + 1 short (5 rows)
        CHECK(e2[1].time_s >= e1.back().time_s);  // job 2 starts after job 1 finished
        CHECK(w0.jobs_seen == 2 && w1.jobs_seen == 2 && w0.units_done + w1.units_done == 12 + 86);
        CHECK(w0.units_done > 0 && w1.units_done > 0);  // both pulled from the shared queue
        mgr.stop();
        std::puts("ok render_manager");
    }
    {
        // cancellation (manager.rs:105-116,365-393): no further units are handed out, units in flight
        // complete, RenderingFinished is still sent and wait() returns
        FakeWorker w(0, 15);
        RenderManager mgr({w.handle()});
        SceneData sd;
        sd.scene_name = "cancel";
        sd.output_settings.image_width = 2;
        sd.output_settings.image_height = 600;
        auto ev = std::make_shared<Channel<std::optional<RenderEvent>>>();
        JobHandle h = mgr.schedule_job(sd, JobConfiguration{1, 5, 10}, ev);  // 60 units x 15 ms
        std::this_thread::sleep_for(std::chrono::milliseconds(80));
        h.cancel();
        h.wait();
        auto e = drain(ev);
        CHECK(e.size() >= 3 && e.size() < 2 + 60 + 1);
        CHECK(e[0].kind == RenderEvent::ImageInfo && e[1].kind == RenderEvent::RenderingStarted);
        CHECK(e.back().kind == RenderEvent::RenderingFinished);
        CHECK(w.units_done >= 1 && w.units_done < 60);
        // a second job on the same manager still runs to completion
        auto ev2 = std::make_shared<Channel<std::optional<RenderEvent>>>();
        JobHandle h2 = mgr.schedule_job(sd, JobConfiguration{1, 5, 300}, ev2);
        h2.wait();
        CHECK(drain(ev2).size() == 2 + 2 + 1);
        mgr.stop();
        // a worker that gives the job up without pulling a unit (e.g. context creation failed): the producer
        // blocked on the bounded channel must be released, the job still finishes
        struct QuitterWorker : Worker {
            std::shared_ptr<Channel<std::optional<WorkerRequest>>> sender = std::make_shared<Channel<std::optional<WorkerRequest>>>();
            std::thread th{[this] {
                while (auto m = sender->recv()) {
                    if (!*m) break;
                    (**m).wg->done();
                }
            }};
            WorkerHandle handle() const override { return WorkerHandle(sender); }
            void stop() override { sender->send(std::nullopt); if (th.joinable()) th.join(); }
            WorkerInfo info() const override { return WorkerInfo{0}; }
        } q;
        RenderManager mgr2({q.handle()});
        auto ev3 = std::make_shared<Channel<std::optional<RenderEvent>>>();
        JobHandle h3 = mgr2.schedule_job(sd, JobConfiguration{1, 5, 10}, ev3);
        h3.wait();
        auto e3 = drain(ev3);
        CHECK(e3.size() == 3 && e3.back().kind == RenderEvent::RenderingFinished);
        mgr2.stop();
        q.stop();
        std::puts("ok cancel");
    }
    std::puts("all ok");
    return 0;
}
