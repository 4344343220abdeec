// flux_cli.cpp -- flag-compatible `flux` front-end driving GpuWorkers.
//
// Mirrors flux/src/main.rs: config_from_args (:126-205; same flags, same defaults: root 1, depth 5,
// 50 rows per work unit), load YAML -> SceneData (:27-29), start workers (:37-70), schedule one job and
// wait (:92-94), ImageBuilder writes "<scene_name>.ppm" into the cwd and prints the total time
// (manager.rs:326-335).  -n/--node ADDRESS[:PORT] adds NetworkWorkers (flux_node processes, flux_net.hpp) and
// -L leaves the local GPUs out, as in main.rs:37-66.  Additions: --seed (the reference seeds from OS
// entropy), --gpus, --outdir, --split.  Not carried over: -g (SDL preview), rejected.
//
// --split sets|rows|units: how the local GPUs share a frame.  `units` is the reference's scheme -- one worker per device pulling
// WorkUnits from the shared channel (manager.rs:100) --, `sets` / `rows` hand the node's GPUs to ONE MultiGpuWorker (flux_multi_*:
// per-device contexts with 1/G of the tables, one launch per device, one RCCL all-gather).  Default: sets (rows below 64 spp) when
// two or more local GPUs are the whole pool, units otherwise (a pool with network nodes needs the shared channel's balancing).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "flux_host.hpp"
#include "flux_net.hpp"

using namespace flux_host;

namespace {

const size_t DEFAULT_SAMPLE_ROOT = 1;  // flux/src/main.rs:20
const size_t DEFAULT_DEPTH = 5;        // flux/src/main.rs:21

struct Config {  // flux/src/main.rs:115-124
    std::vector<std::string> network_workers;
    bool use_local_worker = true;
    size_t sample_root = DEFAULT_SAMPLE_ROOT;
    size_t max_depth = DEFAULT_DEPTH;
    size_t rows_per_work_unit = 50;
    std::string input_filename;
    bool show_live_preview = false;
    size_t num_threads = 0;
    // additions
    uint64_t seed = 1;
    int gpus = -1;
    std::string outdir = ".";
    std::string split = "auto";
};

void usage() {
    std::fprintf(stderr,
                 "flux (MI355X render path)\n\nUSAGE:\n    flux [FLAGS] [OPTIONS] <scene_file>\n\nFLAGS:\n"
                 "    -L               Do not use the local host (its GPUs) for rendering\n"
                 "    -g               Show a live graphical preview window during rendering (not supported)\n\nOPTIONS:\n"
                 "    -d, --depth <DEPTH>          Tracing depth [default 5]\n"
                 "    -n, --node <ADDRESS[:PORT]>  Render using the flux_node process at this address (repeatable)\n"
                 "    -R, --rows <COUNT>           Image rows per work unit [default 50]\n"
                 "    -r, --root <ROOT>            Sample root [default 1]\n"
                 "    -t, --threads <N>            CPU rendering threads (accepted, unused)\n"
                 "        --seed <SEED>            RNG seed [default 1]\n"
                 "        --gpus <N>               number of GPUs to use [default: all]\n"
                 "        --outdir <DIR>           where <scene_name>.ppm is written [default .]\n"
                 "        --split <sets|rows|units> how the local GPUs share a frame [default: sets for >= 2 GPUs without nodes, else units]\n");
}

size_t parse_usize(const char *flag, const char *v) {
    char *end = nullptr;
    unsigned long long x = std::strtoull(v, &end, 10);
    if (end == v || *end != '\0' || v[0] == '-') {
        // usize::from_str(..).unwrap() panics in the reference
        std::fprintf(stderr, "error: invalid value '%s' for '%s'\n", v, flag);
        std::exit(2);
    }
    return (size_t)x;
}

Config config_from_args(int argc, char **argv) {
    Config c;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](const char *flag) -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "error: The argument '%s' requires a value\n", flag);
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "-h" || a == "--help") {
            usage();
            std::exit(0);
        } else if (a == "-d" || a == "--depth") {
            c.max_depth = parse_usize("--depth", need("--depth"));
        } else if (a == "-R" || a == "--rows") {
            c.rows_per_work_unit = parse_usize("--rows", need("--rows"));
        } else if (a == "-r" || a == "--root") {
            c.sample_root = parse_usize("--root", need("--root"));
        } else if (a == "-t" || a == "--threads") {
            c.num_threads = parse_usize("--threads", need("--threads"));
        } else if (a == "-n" || a == "--node") {
            c.network_workers.push_back(need("--node"));
        } else if (a == "-L") {
            c.use_local_worker = false;
        } else if (a == "-g") {
            c.show_live_preview = true;
        } else if (a == "--seed") {
            c.seed = (uint64_t)parse_usize("--seed", need("--seed"));
        } else if (a == "--gpus") {
            c.gpus = (int)parse_usize("--gpus", need("--gpus"));
        } else if (a == "--outdir") {
            c.outdir = need("--outdir");
        } else if (a == "--split") {
            c.split = need("--split");
            if (c.split != "sets" && c.split != "rows" && c.split != "units" && c.split != "auto") {
                std::fprintf(stderr, "error: invalid value '%s' for '--split' (sets, rows, units)\n", c.split.c_str());
                std::exit(2);
            }
        } else if (!a.empty() && a[0] == '-') {
            std::fprintf(stderr, "error: Found argument '%s' which wasn't expected\n", a.c_str());
            std::exit(2);
        } else if (c.input_filename.empty()) {
            c.input_filename = a;
        } else {
            std::fprintf(stderr, "error: Found argument '%s' which wasn't expected\n", a.c_str());
            std::exit(2);
        }
    }
    if (c.input_filename.empty()) {
        std::fprintf(stderr, "error: The following required arguments were not provided:\n    <scene_file>\n");
        usage();
        std::exit(2);
    }
    return c;
}

}  // namespace

int main(int argc, char **argv) {
    Config config = config_from_args(argc, argv);
    if (config.show_live_preview) {
        std::fprintf(stderr, "error: -g: the SDL live preview is not part of the MI355X render path\n");
        return 2;
    }
    SceneData s;
    try {
        s = scene_from_yaml_file(config.input_filename);
    } catch (const FluxError &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    // main.rs:37-66: the local worker(s) unless -L, then one NetworkWorker per -n
    int ndev = config.use_local_worker ? flux_device_count() : 0;
    if (config.use_local_worker && ndev < 1) {
        std::fprintf(stderr, "error: no HIP device visible (this renderer has no CPU fallback)\n");
        return 1;
    }
    if (config.gpus > 0 && config.gpus < ndev) ndev = config.gpus;

    std::vector<std::unique_ptr<Worker>> workers;
    std::vector<WorkerHandle> handles;
    std::string split = config.split;
    if (split == "auto") split = (ndev >= 2 && config.network_workers.empty()) ? "sets" : "units";
    if (split != "units" && !config.network_workers.empty()) {
        std::fprintf(stderr, "error: --split %s renders whole frames on the local GPUs and cannot share a job with -n nodes; use --split units\n",
                     split.c_str());
        return 2;
    }
    MultiGpuWorker *multi = nullptr;
    if (split != "units" && ndev >= 1) {
        std::vector<int> devs;
        for (int d = 0; d < ndev; d++) devs.push_back(d);
        // (sets need 64 spp: below that the library's AUTO picks rows)
        const int shard = split == "rows" ? FLUX_SHARD_ROWS : (config.sample_root * config.sample_root >= 64 ? FLUX_SHARD_SETS : FLUX_SHARD_AUTO);
        multi = new MultiGpuWorker(devs, config.seed, shard);
        workers.emplace_back(multi);
        handles.push_back(workers.back()->handle());
    } else {
        for (int d = 0; d < ndev; d++) {
            workers.emplace_back(new GpuWorker(d, config.seed));
            handles.push_back(workers.back()->handle());
        }
    }
    for (const std::string &endpoint : config.network_workers) {
        std::printf("Connecting to worker %s\n", endpoint.c_str());
        try {
            workers.emplace_back(new NetworkWorker(endpoint));
        } catch (const FluxError &e) {
            std::fprintf(stderr, "Error connecting to %s: %s\n", endpoint.c_str(), e.what());  // main.rs:60-63
            return 1;
        }
        handles.push_back(workers.back()->handle());
    }
    if (handles.empty()) {
        std::fprintf(stderr, "No workers specified, exiting\n");  // main.rs:68-71
        return 1;
    }
    std::printf("Rendering %s with %d GPU worker(s) and %zu network node(s), sample root %zu, depth %zu, %zu rows per work unit\n",
                s.scene_name.c_str(), ndev, config.network_workers.size(), config.sample_root, config.max_depth, config.rows_per_work_unit);
    if (multi) std::printf("The %d local GPU(s) render whole frames as one worker (--split %s: one launch per device, one RCCL all-gather)\n", ndev, split.c_str());

    // flux/src/main.rs:70-111: manager, image builder, schedule one job, wait, shut everything down
    ImageBuilder image_builder;
    image_builder.output_dir = config.outdir;
    int rc = 0;
    if (config.rows_per_work_unit == 0) {  // the reference panics in Job::work_units (job.rs:67-70)
        std::fprintf(stderr, "error: Job row per work unit count invalid\n");
        rc = 1;
    } else {
        RenderManager manager(handles);
        JobHandle job = manager.schedule_job(
            s, JobConfiguration{config.sample_root, config.max_depth, config.rows_per_work_unit}, image_builder.sender());
        job.wait();
        manager.stop();
    }
    image_builder.stop();
    if (multi) {
        const std::vector<double> t = multi->last_timing();
        if (t.size() >= 8)
            std::printf("multi-GPU frame: create %.1f ms (slowest context %.1f, communicators %.1f), frame %.1f ms (kernel %.1f, all-gather %.2f, "
                        "reassembly %.2f, copy %.2f)\n", t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
    }
    for (auto &w : workers) w->stop();
    flux_multi_release_comms();
    if (GpuWorker::failures() > 0) {
        std::fprintf(stderr, "error: %d job(s) / work unit(s) were abandoned by a GPU worker; the image is incomplete\n",
                     GpuWorker::failures());
        rc = 1;
    }
    if (rc == 0 && !image_builder.written_path.empty()) {
        const double samples = (double)s.output_settings.image_width * s.output_settings.image_height *
                               config.sample_root * config.sample_root;
        std::printf("wrote %s (%.1f Msamples/s incl. context creation)\n", image_builder.written_path.c_str(),
                    image_builder.total_time_s > 0 ? samples / image_builder.total_time_s / 1e6 : 0.0);
    }
    return rc;
}
