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
#include "flux_net.hpp"
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
    {
        // CBOR codec against the RFC 8949 Appendix A examples (published test vectors of the format)
        auto hex = [](const std::string &b) {
            static const char *d = "0123456789abcdef";
            std::string h;
            for (unsigned char c : b) { h.push_back(d[c >> 4]); h.push_back(d[c & 15]); }
            return h;
        };
        auto unhex = [](const std::string &h) {
            std::string b;
            for (size_t k = 0; k + 1 < h.size(); k += 2) b.push_back((char)std::stoi(h.substr(k, 2), nullptr, 16));
            return b;
        };
        struct U { uint64_t v; const char *h; } us[] = {{0, "00"}, {1, "01"}, {10, "0a"}, {23, "17"}, {24, "1818"},
            {25, "1819"}, {100, "1864"}, {1000, "1903e8"}, {1000000, "1a000f4240"}, {1000000000000ull, "1b000000e8d4a51000"},
            {18446744073709551615ull, "1bffffffffffffffff"}};
        for (auto &u : us) {
            cbor::Encoder e;
            e.uint(u.v);
            CHECK(hex(e.out) == u.h);
            std::string raw = unhex(u.h);
            cbor::StringReader r(raw);
            cbor::Decoder d(r);
            uint64_t got = 1;
            CHECK(d.read_uint(got) && got == u.v && r.at_end());
        }
        struct I { int64_t v; const char *h; } is[] = {{-1, "20"}, {-10, "29"}, {-100, "3863"}, {-1000, "3903e7"}};
        for (auto &i : is) {
            cbor::Encoder e;
            e.integer(i.v);
            CHECK(hex(e.out) == i.h);
            std::string raw = unhex(i.h);
            cbor::StringReader r(raw);
            cbor::Decoder d(r);
            int64_t got = 0;
            CHECK(d.read_int(got) && got == i.v);
        }
        // floats: the RFC's examples are the shortest exact encodings, which is serde_cbor's f64 rule
        struct F { double v; const char *h; } fs[] = {{0.0, "f90000"}, {-0.0, "f98000"}, {1.0, "f93c00"}, {1.1, "fb3ff199999999999a"},
            {1.5, "f93e00"}, {65504.0, "f97bff"}, {100000.0, "fa47c35000"}, {3.4028234663852886e+38, "fa7f7fffff"},
            {1.0e+300, "fb7e37e43c8800759c"}, {5.960464477539063e-8, "f90001"}, {0.00006103515625, "f90400"}, {-4.0, "f9c400"},
            {-4.1, "fbc010666666666666"}, {INFINITY, "f97c00"}, {-INFINITY, "f9fc00"}};
        for (auto &f : fs) {
            cbor::Encoder e;
            e.real(f.v);
            CHECK(hex(e.out) == f.h);
            std::string raw = unhex(f.h);
            cbor::StringReader r(raw);
            cbor::Decoder d(r);
            double got = 7;
            CHECK(d.read_number(got) && got == f.v && std::signbit(got) == std::signbit(f.v));
        }
        {
            cbor::Encoder e;
            e.real(NAN);
            CHECK(hex(e.out) == "f97e00");
            for (const char *h : {"f97e00", "fa7fc00000", "fb7ff8000000000000", "fa7f800000", "fb7ff0000000000000"}) {
                std::string raw = unhex(h);
                cbor::StringReader r(raw);
                cbor::Decoder d(r);
                double got = 0;
                CHECK(d.read_number(got) && (std::isnan(got) || std::isinf(got)));
            }
        }
        {   // simple values, strings, arrays, maps: definite (encoder) and indefinite (decoder only)
            cbor::Encoder e;
            e.boolean(false); e.boolean(true); e.null(); e.text(""); e.text("a"); e.text("IETF"); e.text("\xc3\xbc");
            e.array(0); e.array(3); e.uint(1); e.uint(2); e.uint(3);
            e.map(2); e.text("a"); e.uint(1); e.text("b"); e.array(2); e.uint(2); e.uint(3);
            CHECK(hex(e.out) == "f4f5f660616164494554466" "2c3bc80830102" "03a26161016162820203");
            cbor::Encoder big;
            big.array(25);
            for (int k = 1; k <= 25; k++) big.uint((uint64_t)k);
            CHECK(hex(big.out) == "98190102030405060708090a0b0c0d0e0f101112131415161718181819");
            // {"a": 1, "b": [2, 3]} definite, then {_ "a": 1, "b": [_ 2, 3]} and (_ "strea", "ming") indefinite, tag 1
            std::string raw = unhex("a26161016162820203" "bf61610161629f0203ffff" "7f657374726561646d696e67ff" "c11a514b67b0" "f7");
            cbor::StringReader r(raw);
            cbor::Decoder d(r);
            for (int rep = 0; rep < 2; rep++) {
                uint64_t n, m, v;
                std::string k;
                CHECK(d.read_map(n) && (rep == 0 ? n == 2 : n == cbor::Decoder::kIndefinite));
                CHECK((rep == 0 || !d.at_break()) && d.read_text(k) && k == "a" && d.read_uint(v) && v == 1);
                CHECK((rep == 0 || !d.at_break()) && d.read_text(k) && k == "b" && d.read_array(m));
                CHECK((rep == 0 || !d.at_break()) && d.read_uint(v) && v == 2 && d.read_uint(v) && v == 3);
                if (rep == 1) CHECK(d.at_break() && d.at_break());
            }
            std::string st;
            CHECK(d.read_text(st) && st == "streaming");
            uint64_t tagged = 0;
            CHECK(d.read_uint(tagged) && tagged == 1363896240);  // tag 1 is skipped
            CHECK(d.read_null());                                // undefined reads as null
            CHECK(d.peek() == cbor::Type::End && !d.failed());
            std::string raw2 = unhex("a26161016162820203" "9f018202039f0405ffff" "5f42010243030405ff" "18");
            cbor::StringReader r2(raw2);
            cbor::Decoder d2(r2);
            CHECK(d2.skip() && d2.skip() && d2.skip());          // whole items of every container flavour
            CHECK(!d2.skip() && d2.failed());                    // truncated head
        }
        {   // hostile input is a decode error, not an allocation or a stack overflow
            std::string huge = unhex("7b00000000ffffffff");      // text string head claiming 4 GiB - 1, no payload
            cbor::StringReader r(huge);
            cbor::Decoder d(r);
            std::string st;
            CHECK(!d.read_text(st) && d.failed() && st.capacity() < (1u << 20));
            std::string chunked = "\x5f";                       // indefinite byte string of 17 MiB in 1 MiB chunks
            for (int k = 0; k < 17; k++) chunked += unhex("5a00100000") + std::string(1u << 20, 'x');
            chunked += "\xff";
            cbor::StringReader rc(chunked);
            cbor::Decoder dc(rc);
            CHECK(!dc.read_bytes(st) && dc.failed());            // total capped at Decoder::kMaxString (16 MiB)
            std::string deep(100000, '\x81');                    // 100000 nested one-element arrays
            deep += "\x01";
            cbor::StringReader rd(deep);
            cbor::Decoder dd(rd);
            CHECK(!dd.skip() && dd.failed());
            std::string ok(100, '\x81');
            ok += "\x01";
            cbor::StringReader ro(ok);
            cbor::Decoder dok(ro);
            CHECK(dok.skip() && !dok.failed());                  // 100 levels: inside serde_cbor's limit of 128
        }
        std::puts("ok cbor");
    }
    {
        // node-protocol messages: serde_cbor 0.9 layout on the way out, both enum layouts on the way in
        auto hex = [](const std::string &b) {
            static const char *d = "0123456789abcdef";
            std::string h;
            for (unsigned char c : b) { h.push_back(d[c >> 4]); h.push_back(d[c & 15]); }
            return h;
        };
        cbor::Encoder e0;
        encode_worker_info(e0, WorkerInfo{16});
        CHECK(hex(e0.out) == "a1" "6b6e756d5f74687265616473" "10");  // {"num_threads": 16}
        NetworkWorkerRequest done;
        cbor::Encoder e1;
        encode_request(e1, done);
        CHECK(hex(e1.out) == "64446f6e65");                           // "Done"
        NetworkWorkerRequest wu;
        wu.kind = NetworkWorkerRequest::WorkUnitMsg;
        wu.unit = WorkUnit{50, 99, JobID{7, 3}};
        cbor::Encoder e2;
        encode_request(e2, wu);
        // ["WorkUnit", {"row_start": 50, "row_end": 99, "job_id": [7, 3]}]
        CHECK(hex(e2.out) == "82" "68576f726b556e6974" "a3" "69726f775f7374617274" "1832" "67726f775f656e64" "1863"
                             "666a6f625f6964" "820703");
        SceneData sd = scene_from_yaml_file(scenes + "/demo2.yml");
        NetworkWorkerRequest set;
        set.kind = NetworkWorkerRequest::SetJob;
        set.job.id = JobID{123456789012345ull, 2};
        set.job.scene_data = sd;
        set.job.config = JobConfiguration{128, 5, 50};
        cbor::Encoder e3;
        encode_request(e3, set);
        {
            cbor::StringReader r(e3.out);
            cbor::Decoder d(r);
            NetworkWorkerRequest back;
            CHECK(decode_request(d, back) && r.at_end() && back.kind == NetworkWorkerRequest::SetJob);
            CHECK(back.job.id.allocator_id == 123456789012345ull && back.job.id.id == 2 && back.job.config.sample_root == 128);
            const SceneData &b = back.job.scene_data;
            CHECK(b.scene_name == "demo2" && b.shapes.size() == 13 && b.camera_data.lens_radius == 0.09);
            CHECK(b.output_settings.image_width == 800 && b.output_settings.pixel_size == 0.5 && b.camera_settings.eye.y == 5.5);
            for (size_t k = 0; k < 13; k++) CHECK(b.shapes[k].index() == sd.shapes[k].index());
            auto *s1 = std::get_if<SphereData>(&b.shapes[1]);
            CHECK(s1 && s1->center.x == -9.0 && s1->radius == 5.0 && !s1->invert && std::get<EmissiveData>(s1->material).power == 10.0);
            auto *g = std::get_if<GlossyReflectiveData>(&std::get<SphereData>(b.shapes[2]).material);
            CHECK(g && g->reflect_exponent == 10000.0 && g->reflect_color.g == 0.6 && g->reflect_amount == 0.5);
            auto *pl = std::get_if<PlaneData>(&b.shapes[12]);
            CHECK(pl && pl->normal.y == 1.0 && std::get<MatteData>(pl->material).diffuse_coefficient == 1.0);
            CHECK(std::get<SphereData>(b.shapes[0]).invert);
        }
        // the same request in the serde_cbor >= 0.10 layout ({"WorkUnit": {...}}) and with an unknown field
        {
            cbor::Encoder e;
            e.map(1); e.text("WorkUnit"); e.map(4); e.key("row_start"); e.uint(1); e.key("extra"); e.array(2); e.uint(1); e.uint(2);
            e.key("row_end"); e.uint(2); e.key("job_id"); e.array(2); e.uint(5); e.uint(6);
            cbor::StringReader r(e.out);
            cbor::Decoder d(r);
            NetworkWorkerRequest back;
            CHECK(decode_request(d, back) && back.kind == NetworkWorkerRequest::WorkUnitMsg && back.unit.row_end == 2 && back.unit.job_id.id == 6);
            std::string bad = "\x82\x65Other\x01";
            cbor::StringReader rb(bad);
            cbor::Decoder db(rb);
            CHECK(!decode_request(db, back));  // unknown variant
        }
        RenderEvent ev;
        ev.kind = RenderEvent::RowsReady;
        ev.result.work_unit = WorkUnit{3, 4, JobID{1, 0}};
        ev.result.rows = {{Color{1.0, 0.5, 0.0}, Color{0.1, 0.2, 0.3}}, {Color{0.0, 0.0, 0.0}, Color{1e-300, 0.25, 0.999}}};
        cbor::Encoder e4;
        encode_event(e4, ev);
        CHECK(hex(e4.out).substr(0, 22) == "8269526f77735265616479");  // ["RowsReady", ...
        RenderEvent st;
        st.kind = RenderEvent::RenderingStarted;
        st.job_id = JobID{9, 1};
        st.time_s = 1538352000.25;
        encode_event(e4, st);
        RenderEvent ii;
        ii.kind = RenderEvent::ImageInfo;
        ii.scene_name = "demo2";
        ii.width = 800;
        ii.height = 600;
        encode_event(e4, ii);
        RenderEvent fin;
        fin.kind = RenderEvent::RenderingFinished;
        fin.time_s = 1538352010.5;
        encode_event(e4, fin);
        {
            cbor::StringReader r(e4.out);  // a stream of concatenated values, as StreamDeserializer reads it
            cbor::Decoder d(r);
            RenderEvent a, b, c, f;
            CHECK(decode_event(d, a) && decode_event(d, b) && decode_event(d, c) && decode_event(d, f) && d.peek() == cbor::Type::End);
            CHECK(a.kind == RenderEvent::RowsReady && a.result.work_unit.row_end == 4 && a.result.rows.size() == 2);
            CHECK(a.result.rows[0][1].g == 0.2 && a.result.rows[1][1].r == 1e-300 && a.result.rows[1][1].b == 0.999);
            CHECK(b.kind == RenderEvent::RenderingStarted && b.job_id.allocator_id == 9 && std::fabs(b.time_s - 1538352000.25) < 1e-6);
            CHECK(c.kind == RenderEvent::ImageInfo && c.scene_name == "demo2" && c.width == 800 && c.height == 600);
            CHECK(f.kind == RenderEvent::RenderingFinished && std::fabs(f.time_s - 1538352010.5) < 1e-6);
        }
        std::puts("ok node messages");
    }
    {
        // loopback: RenderManager -> NetworkWorker -> TCP -> NodeServer -> (fake) worker and back
        FakeWorker backend(7, 1);
        NodeServer server("127.0.0.1", "0", backend.handle(), 3);
        std::thread srv([&] { server.serve_forever(); });
        const std::string endpoint = "127.0.0.1:" + std::to_string(server.port());
        {
            NetworkWorker nw(endpoint);
            CHECK(nw.info().num_threads == 3);
            FakeWorker local(1, 1);
            RenderManager mgr({nw.handle(), local.handle()});
            SceneData sd = scene_from_yaml_file(scenes + "/demo1.yml");
            sd.output_settings.image_width = 5;
            auto ev = std::make_shared<Channel<std::optional<RenderEvent>>>();
            JobHandle h = mgr.schedule_job(sd, JobConfiguration{2, 5, 25}, ev);
            h.wait();
            auto e = drain(ev);
            CHECK(e.size() == 2 + 24 + 1 && e.back().kind == RenderEvent::RenderingFinished);
            std::set<size_t> starts;
            size_t remote = 0;
            for (size_t k = 2; k + 1 < e.size(); k++) {
                CHECK(e[k].kind == RenderEvent::RowsReady && e[k].result.rows.size() == 25 && e[k].result.rows[0].size() == 5);
                CHECK(starts.insert(e[k].result.work_unit.row_start).second);
                CHECK(e[k].result.work_unit.job_id.id == h.job_id.id);
                if (e[k].result.rows[0][0].r == 7.0) remote++;   // rendered by the node's worker
                else CHECK(e[k].result.rows[0][0].r == 1.0);
            }
            CHECK(starts.size() == 24 && remote >= 2 && remote == (size_t)backend.units_done && backend.jobs_seen == 1);
            mgr.stop();
            nw.stop();
            local.stop();
        }
        while (server.clients_served() < 1) std::this_thread::sleep_for(std::chrono::milliseconds(5));
        bool refused = false;
        try {
            NetworkWorker nobody("127.0.0.1:1");
        } catch (const FluxError &) {
            refused = true;
        }
        CHECK(refused);
        server.stop();
        srv.join();
        backend.stop();
        std::puts("ok node loopback");
    }
    std::puts("all ok");
    return 0;
}
