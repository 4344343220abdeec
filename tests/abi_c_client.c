/* abi_c_client.c -- a plain-C client of include/flux_abi.h and nothing else: what a non-Python, non-C++ embedder
 * (the reference's Rust `GpuWorker`) does at the boundary that replaces
 * fluxcore/src/workers.rs:46-60.  Fills scenes/demo1.yml's SceneData by hand (values: scenes/demo1.yml:1-78, reduced to
 * `width` x `height`), creates a context, renders the frame as work units of `rows` rows (Job::work_units,
 * job.rs:65-88), and writes the raw f64 RGB frame to `out`.  tests/test_gpu_abi_client.py builds this with gcc, runs it
 * and compares the file bit for bit with the Python binding's render of the same scene.
 *
 *   abi_c_client <out.bin> <width> <height> <sample_root> <seed> <rows_per_unit> [multi <G> <shard>]
 *
 * With `multi G shard` the frame written to `out` comes from the multi-GPU entry instead (flux_multi_create on devices
 * 0..G-1 + flux_multi_render_frame: per-device contexts, one launch each, ONE ncclAllGather over RCCL, reassembly on
 * devices[0]) -- the counterpart of RenderManager's fan-out and ImageBuilder's gather (manager.rs:156-162, 316-324) --
 * and is compared HERE, with memcmp, against the whole frame from flux_render_rows and against the one-call
 * flux_render_frame_multi; a second flux_multi on the same devices must find the communicators cached.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "flux_abi.h"

static flux_material matte(double r, double g, double b, double kd) {
    flux_material m;
    memset(&m, 0, sizeof m);
    m.kind = FLUX_MAT_MATTE;
    m.color[0] = r; m.color[1] = g; m.color[2] = b;
    m.k = kd;
    return m;
}
static flux_material glossy(double amount, double r, double g, double b, double e) {
    flux_material m;
    memset(&m, 0, sizeof m);
    m.kind = FLUX_MAT_GLOSSY;
    m.color[0] = r; m.color[1] = g; m.color[2] = b;
    m.k = amount;
    m.exponent = e;
    return m;
}
static flux_shape sphere(double x, double y, double z, double radius, int invert, flux_material m) {
    flux_shape s;
    memset(&s, 0, sizeof s);
    s.kind = FLUX_SHAPE_SPHERE;
    s.invert = invert;
    s.p[0] = x; s.p[1] = y; s.p[2] = z;
    s.radius = radius;
    s.material = m;
    return s;
}

int main(int argc, char **argv) {
    if (argc != 7 && !(argc == 10 && strcmp(argv[7], "multi") == 0)) {
        fprintf(stderr, "usage: %s out.bin width height sample_root seed rows_per_unit [multi G shard]\n", argv[0]);
        return 2;
    }
    const uint64_t width = strtoull(argv[2], NULL, 10), height = strtoull(argv[3], NULL, 10);
    flux_job_cfg cfg;
    cfg.sample_root = strtoull(argv[4], NULL, 10);
    cfg.max_trace_depth = 5;
    cfg.rows_per_work_unit = strtoull(argv[6], NULL, 10);
    const uint64_t seed = strtoull(argv[5], NULL, 10);

    flux_shape shapes[6];
    flux_material env;
    memset(&env, 0, sizeof env);
    env.kind = FLUX_MAT_EMISSIVE;
    env.color[0] = 1.0; env.color[1] = 0.9686; env.color[2] = 0.8588;
    env.k = 1.0;
    shapes[0] = sphere(0, 0, 0, 100.0, 1, env);
    shapes[1] = sphere(0, 1, 0, 1.0, 0, matte(0.0, 0.7, 0.6, 1.0));
    shapes[2] = sphere(2, 1, 2, 1.0, 0, glossy(0.9, 0.9, 1.0, 0.9, 100.0));
    shapes[3] = sphere(4, 1, 4, 1.0, 0, glossy(0.9, 0.9, 1.0, 0.9, 100000.0));
    shapes[4] = sphere(6, 1, 2, 1.0, 0, matte(0.5, 0.3, 0.8, 1.0));
    memset(&shapes[5], 0, sizeof shapes[5]);
    shapes[5].kind = FLUX_SHAPE_PLANE;
    shapes[5].n[1] = 1.0;
    shapes[5].material = matte(0.5, 0.5, 0.5, 1.0);

    flux_scene_desc sd;
    memset(&sd, 0, sizeof sd);
    sd.scene_name = "demo1";
    sd.image_width = width;
    sd.image_height = height;
    sd.pixel_size = 0.5 * 800.0 / (double)width;   /* same field of view as the 800-wide original */
    sd.eye[0] = 2.5; sd.eye[1] = 1.5; sd.eye[2] = -9.0;
    sd.look_at[0] = 2.5; sd.look_at[1] = 1.0; sd.look_at[2] = 0.0;
    sd.up[1] = 1.0;
    sd.zoom_factor = 1.0;
    sd.view_plane_distance = 500.0;
    sd.focal_distance = 10.0;
    sd.lens_radius = 0.0;
    sd.num_shapes = 6;
    sd.shapes = shapes;
    sd.num_meshes = 0;
    sd.meshes = NULL;

    if (flux_abi_version() != FLUX_ABI_VERSION) {
        fprintf(stderr, "ABI version mismatch: library %u, header %u\n", flux_abi_version(), FLUX_ABI_VERSION);
        return 1;
    }
    flux_ctx *ctx = NULL;
    if (flux_ctx_create(&sd, &cfg, seed, 0, &ctx) != FLUX_OK) {
        fprintf(stderr, "flux_ctx_create: %s\n", flux_last_error());
        return 1;
    }
    double *frame = (double *)calloc((size_t)(width * height * 3), sizeof(double));
    flux_work_unit units[4096];
    const int64_t nunits = flux_work_units(height, cfg.rows_per_work_unit, units, 4096);
    if (!frame || nunits < 0 || nunits > 4096) {
        fprintf(stderr, "work units: %s\n", flux_last_error());
        return 1;
    }
    for (int64_t u = 0; u < nunits; u++) {   /* the job loop of workers.rs:56-60 */
        if (flux_render_rows(ctx, units[u].row_start, units[u].row_end, frame + units[u].row_start * width * 3) != FLUX_OK) {
            fprintf(stderr, "flux_render_rows: %s\n", flux_last_error());
            return 1;
        }
    }
    /* error behaviour of the boundary: bad rows are a code + message, never an abort (flux_abi.h) */
    if (flux_render_rows(ctx, height, height, frame) != FLUX_E_INVALID || strlen(flux_last_error()) == 0) {
        fprintf(stderr, "out-of-range work unit was not rejected\n");
        return 1;
    }
    if (argc == 10) {
        const uint64_t G = strtoull(argv[8], NULL, 10);
        const int shard = atoi(argv[9]);
        const size_t doubles = (size_t)(width * height * 3);
        int devices[64];
        double *whole = (double *)malloc(doubles * sizeof(double)), *multi = (double *)malloc(doubles * sizeof(double));
        double *once = (double *)malloc(doubles * sizeof(double));
        if (!whole || !multi || !once || G < 1 || G > 64) return 1;
        for (uint64_t g = 0; g < G; g++) devices[g] = (int)g;
        if (flux_render_rows(ctx, 0, height - 1, whole) != FLUX_OK) {   /* every row, also the one job.rs:74 never issues */
            fprintf(stderr, "flux_render_rows: %s\n", flux_last_error());
            return 1;
        }
        flux_multi *m = NULL, *m2 = NULL;
        if (flux_multi_create(&sd, &cfg, seed, devices, G, shard, &m) != FLUX_OK) {
            fprintf(stderr, "flux_multi_create: %s\n", flux_last_error());
            return 1;
        }
        for (int frame_no = 0; frame_no < 2; frame_no++) {   /* twice: the second frame re-uses every buffer */
            memset(multi, 0xff, doubles * sizeof(double));
            if (flux_multi_render_frame(m, multi) != FLUX_OK) {
                fprintf(stderr, "flux_multi_render_frame: %s\n", flux_last_error());
                return 1;
            }
            if (memcmp(multi, whole, doubles * sizeof(double)) != 0) {
                fprintf(stderr, "frame %d of flux_multi_render_frame differs from flux_render_rows\n", frame_no);
                return 1;
            }
        }
        uint64_t info[FLUX_MULTI_INFO_WORDS];
        double ms[FLUX_MULTI_TIMING_WORDS];
        if (flux_multi_info(m, info) != FLUX_OK || flux_multi_timing(m, ms) != FLUX_OK) return 1;
        printf("multi: devices %llu shard %llu rccl %llu share_doubles %llu ctx_bytes %llu buffer_bytes %llu cached %llu\n",
               (unsigned long long)info[0], (unsigned long long)info[1], (unsigned long long)info[2], (unsigned long long)info[3],
               (unsigned long long)info[4], (unsigned long long)info[5], (unsigned long long)info[6]);
        printf("multi_ms: create %.3f ctx %.3f comm %.3f frame %.3f kernel %.3f gather %.3f assemble %.3f d2h %.3f\n", ms[0], ms[1], ms[2],
               ms[3], ms[4], ms[5], ms[6], ms[7]);
        if (info[0] != G || info[2] == 0 || info[6] != 0) {
            fprintf(stderr, "flux_multi_info: unexpected contents\n");
            return 1;
        }
        flux_ctx *rank0 = NULL;
        int64_t plan[FLUX_PLAN_WORDS];
        if (flux_multi_ctx(m, 0, &rank0) != FLUX_OK || !rank0 || flux_multi_ctx(m, G, &rank0) != FLUX_E_INVALID) return 1;
        if (flux_multi_ctx(m, 0, &rank0) != FLUX_OK || flux_ctx_launch_plan(rank0, info[1] == FLUX_SHARD_ROWS ? (height + G - 1) / G : 0,
                                                                           info[1] == FLUX_SHARD_SETS ? (width + G - 1) / G : 0, plan) != FLUX_OK) {
            fprintf(stderr, "launch plan of rank 0: %s\n", flux_last_error());
            return 1;
        }
        printf("multi_plan: kernel %lld block %lld blocks %lld\n", (long long)plan[0], (long long)plan[1], (long long)plan[2]);
        /* a second job on the same devices (flux/src/main.rs:247,304,313 schedules job after job): communicators from the cache */
        if (flux_multi_create(&sd, &cfg, seed, devices, G, shard, &m2) != FLUX_OK || flux_multi_info(m2, info) != FLUX_OK || info[6] != 1) {
            fprintf(stderr, "second flux_multi_create: %s (cached %llu)\n", flux_last_error(), (unsigned long long)info[6]);
            return 1;
        }
        const void *d_frame = NULL;
        if (flux_multi_render_frame_device(m2, &d_frame) != FLUX_OK || !d_frame) {
            fprintf(stderr, "flux_multi_render_frame_device: %s\n", flux_last_error());
            return 1;
        }
        flux_multi_destroy(m2);
        flux_multi_destroy(m);
        /* the one-call form: a job's whole life */
        if (flux_render_frame_multi(&sd, &cfg, seed, NULL, G, shard, once) != FLUX_OK) {
            fprintf(stderr, "flux_render_frame_multi: %s\n", flux_last_error());
            return 1;
        }
        if (memcmp(once, whole, doubles * sizeof(double)) != 0) {
            fprintf(stderr, "flux_render_frame_multi differs from flux_render_rows\n");
            return 1;
        }
        /* error behaviour: codes + messages, never an abort */
        devices[0] = 0;
        devices[1] = 0;
        if (flux_multi_create(&sd, &cfg, seed, devices, 2, shard, &m) != FLUX_E_INVALID || strlen(flux_last_error()) == 0) return 1;
        if (flux_multi_release_comms() != 1) {
            fprintf(stderr, "expected ONE cached device list\n");
            return 1;
        }
        memcpy(frame, multi, doubles * sizeof(double));
        free(whole);
        free(multi);
        free(once);
    }
    flux_ctx_destroy(ctx);
    FILE *f = fopen(argv[1], "wb");
    if (!f || fwrite(frame, sizeof(double), (size_t)(width * height * 3), f) != (size_t)(width * height * 3) || fclose(f) != 0) {
        fprintf(stderr, "cannot write %s\n", argv[1]);
        return 1;
    }
    free(frame);
    printf("%lld work units, %llux%llu, %llu spp\n", (long long)nunits, (unsigned long long)width,
           (unsigned long long)height, (unsigned long long)(cfg.sample_root * cfg.sample_root));
    return 0;
}
