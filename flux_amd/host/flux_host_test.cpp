// flux_host_test.cpp -- CPU-only checks of the C++ host layer (run by tests/test_host_cpp.py).
// Prints one "ok <name>" line per check; exits non-zero on the first failure.
#include <cmath>
#include <cstdio>
#include <cstdlib>
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

static bool throws(const std::string &yaml, const char *needle) {
    try {
        scene_from_yaml_text(yaml);
    } catch (const FluxError &e) {
        return std::string(e.what()).find(needle) != std::string::npos;
    }
    return false;
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
    std::puts("all ok");
    return 0;
}
