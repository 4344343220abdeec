// sampler_debug.cpp -- the reference's sampler-debug tool (sampler-debug/src/main.rs) on the device
// generators: plots the regular, jittered, multi-jittered and correlated multi-jittered sample sets and
// their to_hemisphere(.., 0.0) images into 100x100 PPMs
//   sampler-debug-{r,j,mj,cmj}.ppm, sampler-debug-{r,j,mj,cmj}-hemi.ppm
// with the reference's plotting rules (main.rs:12-23) and PPM writer (image.rs:43-61).
// Flags: -r/--root <n> (default 10, main.rs:74-79); extra: --seed <u64> (the reference seeds from entropy),
// --outdir <dir>, --device <i>.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/flux_abi.h"

namespace {

struct Img {  // Image::new(100, 100) + set_pixel; never-set pixels are written as zeros (image.rs:55-59)
    size_t w = 100, h = 100;
    std::vector<double> rgb = std::vector<double>(100 * 100 * 3, 0.0);
    void set_pixel(size_t row, size_t col, double r, double g, double b) {
        if (row >= h || col >= w) return;  // the reference panics (image.rs:29-35); cannot happen for [0,1) samples
        double *p = &rgb[(row * w + col) * 3];
        p[0] = r; p[1] = g; p[2] = b;
    }
};

int plot(int device, int kind, const char *basename, uint64_t root, uint64_t seed, const std::string &outdir) {
    const size_t N = (size_t)root * root;
    std::vector<double> xy(N * 2), hemi(N * 3);
    if (flux_sampler_grid(device, kind, root, seed, xy.data(), hemi.data()) != FLUX_OK) {
        std::fprintf(stderr, "error: %s\n", flux_last_error());
        return 1;
    }
    Img i1;
    for (size_t k = 0; k < N; k++) {  // plot_2d_sample, main.rs:12-16
        const size_t x = (size_t)(xy[2 * k] * ((double)i1.w - 0.01));
        const size_t y = (size_t)(xy[2 * k + 1] * ((double)i1.h - 0.01));
        i1.set_pixel(y, x, 1.0, 0.2, 0.2);
    }
    const std::string path1 = outdir + "/sampler-debug-" + basename + ".ppm";
    if (flux_write_ppm(path1.c_str(), i1.rgb.data(), i1.w, i1.h, nullptr) != FLUX_OK) {
        std::fprintf(stderr, "error: %s\n", flux_last_error());
        return 1;
    }
    std::printf("Wrote output to %s\n", path1.c_str());
    Img i2;
    for (size_t k = 0; k < N; k++) {  // plot_hemi_sample, main.rs:18-23
        const size_t x = (size_t)(((hemi[3 * k] / 2.0) + 0.5) * ((double)i2.w - 0.01));
        const size_t y = (size_t)(((hemi[3 * k + 1] / 2.0) + 0.5) * ((double)i2.h - 0.01));
        i2.set_pixel(y, x, hemi[3 * k + 2], 0.2, 0.2);
    }
    const std::string path2 = outdir + "/sampler-debug-" + basename + "-hemi.ppm";
    if (flux_write_ppm(path2.c_str(), i2.rgb.data(), i2.w, i2.h, nullptr) != FLUX_OK) {
        std::fprintf(stderr, "error: %s\n", flux_last_error());
        return 1;
    }
    std::printf("Wrote output to %s\n", path2.c_str());
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    uint64_t root = 10, seed = 1;
    int device = 0;
    std::string outdir = ".";
    for (int a = 1; a < argc; a++) {
        const std::string f = argv[a];
        auto need = [&](const char *what) -> const char * {
            if (a + 1 >= argc) {
                std::fprintf(stderr, "error: The argument '%s' requires a value\n", what);
                std::exit(2);
            }
            return argv[++a];
        };
        if (f == "-r" || f == "--root") {
            char *end = nullptr;
            const char *v = need("--root <sample_root>");
            root = std::strtoull(v, &end, 10);
            if (!*v || *end || root == 0) {
                std::fprintf(stderr, "error: invalid value '%s' for --root\n", v);
                return 2;
            }
        } else if (f == "--seed") {
            seed = std::strtoull(need("--seed <seed>"), nullptr, 10);
        } else if (f == "--outdir") {
            outdir = need("--outdir <dir>");
        } else if (f == "--device") {
            device = std::atoi(need("--device <i>"));
        } else if (f == "-h" || f == "--help") {
            std::puts("sampler-debug\nSampler debugging utility\n\nUSAGE:\n    sampler_debug [OPTIONS]\n\nOPTIONS:\n"
                      "    -r, --root <sample_root>    Sample root\n        --seed <seed>\n        --outdir <dir>\n        --device <i>");
            return 0;
        } else {
            std::fprintf(stderr, "error: Found argument '%s' which wasn't expected\n", f.c_str());
            return 2;
        }
    }
    if (flux_device_count() < 1) {
        std::fprintf(stderr, "error: no HIP device visible (the generators run on the device; no CPU fallback)\n");
        return 1;
    }
    // main.rs:53-56
    if (plot(device, FLUX_SAMPLER_REGULAR, "r", root, seed, outdir)) return 1;
    if (plot(device, FLUX_SAMPLER_JITTERED, "j", root, seed, outdir)) return 1;
    if (plot(device, FLUX_SAMPLER_MULTI_JITTERED, "mj", root, seed, outdir)) return 1;
    if (plot(device, FLUX_SAMPLER_CORRELATED_MULTI_JITTERED, "cmj", root, seed, outdir)) return 1;
    return 0;
}
