// bvh_selftest.cpp -- CPU-only invariants of flux_amd/csrc/bvh.cpp (extension: no reference counterpart), built and run by
// tests/test_bvh_builder.py.  For meshes of several shapes (a jittered grid, a triangle soup, degenerate and single-triangle
// inputs, a grid far from the origin):
//   * build_bvh: every triangle lies in exactly one leaf and inside every box on the way down to it (f32 boxes, padded);
//   * quantize_bvh: returns true and every 16-bit box contains its f32 box (grid values in exact arithmetic);
//   * build_wide: the 4-wide tree reaches every leaf record exactly once, every record's triangles are the DevTri operands bit
//     for bit (A = v0,e1,e2; B = v0,e2,e3), every DevTri slot appears exactly once, child boxes equal the binary tree's quantised
//     boxes, empty slots carry the inverted box, and BvhInfo::wide_stack bounds the pushes along every path.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "../flux_amd/csrc/flux_bvh.h"

using namespace flux;

static int fails = 0;
#define CHECK(c, ...)                                    \
    do {                                                 \
        if (!(c)) {                                      \
            if (fails++ < 20) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } \
        }                                                \
    } while (0)

static uint32_t rng_state = 12345u;
static double urand() { rng_state = rng_state * 1664525u + 1013904223u; return (rng_state >> 8) / 16777216.0; }

static DevTri make_tri(const double a[3], const double b[3], const double c[3], int id) {
    DevTri t;
    std::memset(&t, 0, sizeof t);
    t.v0x = a[0]; t.v0y = a[1]; t.v0z = a[2];
    t.e1x = b[0] - a[0]; t.e1y = b[1] - a[1]; t.e1z = b[2] - a[2];
    t.e2x = c[0] - a[0]; t.e2y = c[1] - a[1]; t.e2z = c[2] - a[2];
    t.id = id;
    return t;
}

static std::vector<DevTri> grid(int nx, int nz, const double off[3], double scale) {
    std::vector<double> V((size_t)(nx + 1) * (nz + 1) * 3);
    auto vid = [&](int i, int j) { return (size_t)i * (nz + 1) + j; };
    for (int i = 0; i <= nx; i++)
        for (int j = 0; j <= nz; j++) {
            double *p = &V[vid(i, j) * 3];
            p[0] = off[0] + scale * (-14.0 + 28.0 * i / nx);
            p[1] = off[1] + scale * (0.35 * std::sin(1.7 * i) * std::cos(1.3 * j) + 0.02 * (2 * urand() - 1));
            p[2] = off[2] + scale * (-10.0 + 30.0 * j / nz);
        }
    std::vector<DevTri> t;
    for (int i = 0; i < nx; i++)
        for (int j = 0; j < nz; j++) {
            t.push_back(make_tri(&V[vid(i, j) * 3], &V[vid(i, j + 1) * 3], &V[vid(i + 1, j + 1) * 3], (int)t.size() + 13));
            t.push_back(make_tri(&V[vid(i, j) * 3], &V[vid(i + 1, j + 1) * 3], &V[vid(i + 1, j) * 3], (int)t.size() + 13));
        }
    return t;
}

static std::vector<DevTri> soup(int n) {
    std::vector<DevTri> t;
    for (int k = 0; k < n; k++) {
        double a[3], b[3], c[3];
        for (int q = 0; q < 3; q++) {
            a[q] = 10 * urand() - 5;
            b[q] = a[q] + urand() - 0.5;
            c[q] = a[q] + urand() - 0.5;
        }
        t.push_back(make_tri(a, b, c, k + 3));
    }
    return t;
}

static void tri_box(const DevTri &t, double lo[3], double hi[3]) {
    const double v[3][3] = {{t.v0x, t.v0y, t.v0z}, {t.v0x + t.e1x, t.v0y + t.e1y, t.v0z + t.e1z}, {t.v0x + t.e2x, t.v0y + t.e2y, t.v0z + t.e2z}};
    for (int a = 0; a < 3; a++) {
        lo[a] = std::fmin(v[0][a], std::fmin(v[1][a], v[2][a]));
        hi[a] = std::fmax(v[0][a], std::fmax(v[1][a], v[2][a]));
    }
}

static void check_mesh(const char *name, std::vector<DevTri> tris) {
    const size_t n = tris.size();
    std::vector<int> ids_before;
    for (auto &t : tris) ids_before.push_back(t.id);
    std::vector<DevNode> nodes;
    BvhInfo info;
    build_bvh(tris, nodes, info);
    CHECK(tris.size() == n && info.tris == n, "%s: triangle count", name);
    if (n == 0) { std::printf("ok %s (empty)\n", name); return; }
    // binary tree: every triangle in exactly one leaf, inside every box above it
    std::vector<int> seen(n, 0);
    std::function<void(int32_t, int)> walk = [&](int32_t node, int depth) {
        CHECK(depth < (int)kBvhMaxDepth + 1, "%s: depth", name);
        const DevNode &N = nodes[(size_t)node];
        for (int side = 0; side < 2; side++) {
            const int32_t link = side ? N.child1 : N.child0;
            const float *lo = side ? N.lo1 : N.lo0, *hi = side ? N.hi1 : N.hi0;
            std::function<void(int32_t)> contains = [&](int32_t l) {
                if (l >= 0) { contains(nodes[(size_t)l].child0); contains(nodes[(size_t)l].child1); return; }
                const int ref = ~l, first = ref >> 3, cnt = ref & 7;
                for (int k = 0; k < cnt; k++) {
                    double tl[3], th[3];
                    tri_box(tris[(size_t)(first + k)], tl, th);
                    for (int a = 0; a < 3; a++) CHECK((double)lo[a] <= tl[a] && (double)hi[a] >= th[a], "%s: f32 box does not contain triangle %d", name, first + k);
                }
            };
            if (n <= 4096) contains(link);  // quadratic in depth: small meshes only
            if (link >= 0) walk(link, depth + 1);
            else {
                const int ref = ~link, first = ref >> 3, cnt = ref & 7;
                CHECK(cnt <= kBvhLeafSize, "%s: leaf size %d", name, cnt);
                for (int k = 0; k < cnt; k++) seen[(size_t)(first + k)]++;
            }
        }
    };
    walk(0, 0);
    for (size_t k = 0; k < n; k++) CHECK(seen[k] == 1, "%s: triangle slot %zu in %d leaves", name, k, seen[k]);
    // quantised boxes contain the f32 boxes
    std::vector<DevNodeQ> q;
    const bool ok = quantize_bvh(nodes, q, info);
    CHECK(ok, "%s: quantize_bvh reports lost containment", name);
    auto val = [&](uint16_t v, int a) { return (double)info.qmin[a] + (double)v * (double)info.qstep[a]; };
    for (size_t k = 0; k < nodes.size(); k++)
        for (int a = 0; a < 3; a++) {
            if (std::isfinite(nodes[k].lo0[a])) CHECK(val(q[k].lo0[a], a) <= nodes[k].lo0[a] && val(q[k].hi0[a], a) >= nodes[k].hi0[a], "%s: q box 0 of node %zu", name, k);
            if (std::isfinite(nodes[k].lo1[a])) CHECK(val(q[k].lo1[a], a) <= nodes[k].lo1[a] && val(q[k].hi1[a], a) >= nodes[k].hi1[a], "%s: q box 1 of node %zu", name, k);
        }
    // the 4-wide tree
    std::vector<DevNode4Q> w;
    std::vector<DevLeafRec> L;
    build_wide(nodes, q, tris, w, L, info);
    CHECK(info.wide_nodes == w.size() && info.leaf_records == L.size(), "%s: wide counts", name);
    std::vector<int> rec_seen(L.size(), 0), slot_seen(n, 0);
    uint64_t max_stack = 0;
    std::function<void(int32_t, uint64_t)> wwalk = [&](int32_t node, uint64_t stacked) {
        const DevNode4Q &W = w[(size_t)node];
        int kids = 0;
        for (int k = 0; k < 4; k++) kids += W.link[k] != (int32_t)0x80000000;
        CHECK(kids >= 1, "%s: wide node %d without children", name, node);
        const uint64_t here = stacked + (uint64_t)(kids > 0 ? kids - 1 : 0);
        if (here > max_stack) max_stack = here;
        for (int k = 0; k < 4; k++) {
            if (W.link[k] == (int32_t)0x80000000) {
                CHECK(W.bx[k] == 0x0000ffffu && W.by[k] == 0x0000ffffu && W.bz[k] == 0x0000ffffu, "%s: empty slot's box", name);
                continue;
            }
            CHECK((W.bx[k] & 0xffffu) <= (W.bx[k] >> 16) && (W.by[k] & 0xffffu) <= (W.by[k] >> 16) && (W.bz[k] & 0xffffu) <= (W.bz[k] >> 16), "%s: inverted child box", name);
            if (W.link[k] >= 0) { CHECK((size_t)W.link[k] < w.size() && W.link[k] > node, "%s: link order", name); wwalk(W.link[k], here); }
            else {
                const int ref = ~W.link[k], first = ref >> 3, cnt = ref & 7;
                CHECK(cnt >= 1 && (size_t)(first + cnt) <= L.size(), "%s: leaf reference", name);
                for (int r = 0; r < cnt; r++) {
                    rec_seen[(size_t)(first + r)]++;
                    const DevLeafRec &R = L[(size_t)(first + r)];
                    // the record's boxes: its triangles inside the child's quantised box
                    for (int half = 0; half < 2; half++) {
                        const int sl = R.slot[half];
                        if (sl < 0) { CHECK(half == 1, "%s: record without triangle A", name); continue; }
                        slot_seen[(size_t)sl]++;
                        const DevTri &T = tris[(size_t)sl];
                        CHECK(R.id[half] == T.id, "%s: record id", name);
                        const double *ea = half ? R.e2 : R.e1, *eb = half ? R.e3 : R.e2;
                        CHECK(R.v0[0] == T.v0x && R.v0[1] == T.v0y && R.v0[2] == T.v0z && ea[0] == T.e1x && ea[1] == T.e1y && ea[2] == T.e1z &&
                                  eb[0] == T.e2x && eb[1] == T.e2y && eb[2] == T.e2z, "%s: record %d half %d is not DevTri %d bit for bit", name, first + r, half, sl);
                        double tl[3], th[3];
                        tri_box(T, tl, th);
                        const uint32_t bw[3] = {W.bx[k], W.by[k], W.bz[k]};
                        for (int a = 0; a < 3; a++)
                            CHECK(val((uint16_t)(bw[a] & 0xffffu), a) <= tl[a] && val((uint16_t)(bw[a] >> 16), a) >= th[a], "%s: wide child box does not contain triangle %d", name, sl);
                    }
                }
            }
        }
    };
    wwalk(0, 0);
    for (size_t k = 0; k < L.size(); k++) CHECK(rec_seen[k] == 1, "%s: leaf record %zu reached %d times", name, k, rec_seen[k]);
    for (size_t k = 0; k < n; k++) CHECK(slot_seen[k] == 1, "%s: DevTri slot %zu in %d records", name, k, slot_seen[k]);
    CHECK(max_stack <= info.wide_stack, "%s: stack bound %llu < %llu", name, (unsigned long long)info.wide_stack, (unsigned long long)max_stack);
    // the arena layout (round 5): children contiguous, one record per leaf child, ONE stack entry per node
    {
        std::vector<DevNode4A> A;
        BvhInfo ia = info;
        build_wide_arena(nodes, q, tris, A, ia);
        CHECK(ia.arena_units == A.size() && A.size() >= 2, "%s: arena size", name);
        std::vector<int> unit_used(A.size(), 0), slot2(n, 0);
        uint64_t max_entries = 0, n_nodes = 0, n_recs = 0;
        std::function<void(uint32_t, uint64_t, int)> awalk = [&](uint32_t unit, uint64_t stacked, int depth) {
            CHECK(unit < A.size(), "%s: arena unit out of range", name);
            CHECK(depth <= 64, "%s: arena depth", name);
            if (unit >= A.size() || depth > 64) return;
            const DevNode4A &W = A[unit];
            unit_used[unit]++;
            n_nodes++;
            const uint32_t n_in = (W.meta >> 4) & 7u, first = (W.meta >> 6) & ~1u;
            CHECK((W.meta & 15u) == 0u, "%s: meta's mask bits", name);
            CHECK(W.kids >= 1 && W.kids <= 4 && n_in <= W.kids, "%s: arena node %u: kids %u inner %u", name, unit, W.kids, n_in);
            CHECK(first > unit && (first & 1u) == 0u, "%s: children block of %u at %u", name, unit, first);
            const uint64_t here = stacked + (W.kids >= 2 ? 1 : 0);
            if (here > max_entries) max_entries = here;
            for (uint32_t k = 0; k < 4; k++) {
                if (k >= W.kids) {
                    CHECK(W.bx[k] == 0x0000ffffu && W.by[k] == 0x0000ffffu && W.bz[k] == 0x0000ffffu, "%s: arena empty slot's box", name);
                    continue;
                }
                CHECK((W.bx[k] & 0xffffu) <= (W.bx[k] >> 16) && (W.by[k] & 0xffffu) <= (W.by[k] >> 16) && (W.bz[k] & 0xffffu) <= (W.bz[k] >> 16), "%s: arena inverted child box", name);
                // a stack entry with any pending mask yields the same links as the node's own meta word
                const int32_t link = wide_link(W.meta, k);
                CHECK(link == wide_link(W.meta | 15u, k) && link == wide_link(W.meta | (1u << k), k), "%s: wide_link depends on the mask", name);
                if (k < n_in) {
                    CHECK(link == (int32_t)(first + k), "%s: inner link", name);
                    awalk((uint32_t)link, here, depth + 1);
                } else {
                    CHECK(link < 0, "%s: leaf link", name);
                    const uint32_t ru = (uint32_t)~link;
                    CHECK(ru + 1 < A.size() && (ru & 1u) == 0u, "%s: leaf record unit %u", name, ru);
                    if (ru + 1 >= A.size()) continue;
                    unit_used[ru]++;
                    unit_used[ru + 1]++;
                    n_recs++;
                    DevLeafRec R;
                    std::memcpy(&R, &A[ru], sizeof(R));
                    for (int half = 0; half < 2; half++) {
                        const int sl = R.slot[half];
                        if (sl < 0) { CHECK(half == 1, "%s: arena record without triangle A", name); continue; }
                        CHECK((size_t)sl < n, "%s: arena record slot", name);
                        if ((size_t)sl >= n) continue;
                        slot2[(size_t)sl]++;
                        const DevTri &T = tris[(size_t)sl];
                        CHECK(R.id[half] == T.id, "%s: arena record id", name);
                        const double *ea = half ? R.e2 : R.e1, *eb = half ? R.e3 : R.e2;
                        CHECK(R.v0[0] == T.v0x && R.v0[1] == T.v0y && R.v0[2] == T.v0z && ea[0] == T.e1x && ea[1] == T.e1y && ea[2] == T.e1z &&
                                  eb[0] == T.e2x && eb[1] == T.e2y && eb[2] == T.e2z, "%s: arena record at %u half %d is not DevTri %d bit for bit", name, ru, half, sl);
                        double tl[3], th[3];
                        tri_box(T, tl, th);
                        const uint32_t bw[3] = {W.bx[k], W.by[k], W.bz[k]};
                        for (int a = 0; a < 3; a++)
                            CHECK(val((uint16_t)(bw[a] & 0xffffu), a) <= tl[a] && val((uint16_t)(bw[a] >> 16), a) >= th[a], "%s: arena child box does not contain triangle %d", name, sl);
                    }
                }
            }
        };
        awalk(0, 0, 0);
        for (size_t k = 0; k < n; k++) CHECK(slot2[k] == 1, "%s: DevTri slot %zu in %d arena records", name, k, slot2[k]);
        for (size_t u = 0; u < A.size(); u++) CHECK(unit_used[u] <= 1, "%s: arena unit %zu reached %d times", name, u, unit_used[u]);
        CHECK(n_nodes == ia.wide_nodes && n_recs == ia.leaf_records, "%s: arena counts", name);
        CHECK(max_entries <= ia.wide_stack, "%s: arena stack bound %llu < %llu", name, (unsigned long long)ia.wide_stack, (unsigned long long)max_entries);
        CHECK(ia.wide_stack <= info.wide_stack || info.wide_stack == 0, "%s: the per-node bound %llu exceeds the per-child bound %llu", name,
              (unsigned long long)ia.wide_stack, (unsigned long long)info.wide_stack);
        std::printf("   arena: %zu units, %llu nodes, %llu records (%llu leaves split), stack %llu entries (per child: %llu)\n", A.size(),
                    (unsigned long long)ia.wide_nodes, (unsigned long long)ia.leaf_records, (unsigned long long)ia.split_leaves,
                    (unsigned long long)ia.wide_stack, (unsigned long long)info.wide_stack);
    }
    // ids survive the reordering as a permutation

    std::vector<int> ids_after;
    for (auto &t : tris) ids_after.push_back(t.id);
    std::sort(ids_before.begin(), ids_before.end());
    std::sort(ids_after.begin(), ids_after.end());
    CHECK(ids_before == ids_after, "%s: ids are not a permutation", name);
    std::printf("ok %s: %zu triangles, %zu nodes (depth %llu), %zu wide nodes, %zu records (%llu quads), stack %llu\n", name, n, nodes.size(),
                (unsigned long long)info.max_depth, w.size(), L.size(), (unsigned long long)info.fused_leaves, (unsigned long long)info.wide_stack);
}

#include <algorithm>
int main() {
    const double o0[3] = {0, 0, 0}, far1[3] = {5.0e3, 0, 0}, far2[3] = {1.0e6, -1.0e6, 1.0e6};
    check_mesh("empty", {});
    check_mesh("one triangle", soup(1));
    check_mesh("two triangles", soup(2));
    check_mesh("three triangles", soup(3));
    check_mesh("soup 1000", soup(1000));
    check_mesh("grid 40x30", grid(40, 30, o0, 1.0));
    check_mesh("grid 200x100", grid(200, 100, o0, 1.0));
    check_mesh("grid 40x30 at 5e3", grid(40, 30, far1, 0.05));
    check_mesh("grid 40x30 at 1e6", grid(40, 30, far2, 0.05));
    {   // degenerate: repeated vertices (cleared edges, as abi.hip makes them) among ordinary triangles
        std::vector<DevTri> t = soup(50);
        for (int k = 0; k < 10; k++) { t[(size_t)k * 3].e1x = t[(size_t)k * 3].e1y = t[(size_t)k * 3].e1z = 0; t[(size_t)k * 3].e2x = t[(size_t)k * 3].e2y = t[(size_t)k * 3].e2z = 0; }
        check_mesh("soup with degenerate triangles", t);
    }
    {   // all triangles identical: SAH cannot separate them, the median split must bound the depth
        std::vector<DevTri> t(300, soup(1)[0]);
        for (size_t k = 0; k < t.size(); k++) t[k].id = (int)k + 1;
        check_mesh("300 coincident triangles", t);
    }
    {   // round 6: meshes of 65 536 triangles and more are built by worker threads below a serial top (bvh.cpp build_bvh): the tree,
        // the triangle order and everything derived from them must be THE SAME as the one-thread build's, bit for bit
        const std::vector<DevTri> mesh = grid(300, 200, o0, 1.0);  // 120 000 triangles
        std::vector<DevNode> nodes_ref;
        std::vector<DevTri> tris_ref;
        for (const char *threads : {"1", "2", "5", "16"}) {
            setenv("FLUX_BUILD_THREADS", threads, 1);
            std::vector<DevTri> t = mesh;
            std::vector<DevNode> nodes;
            BvhInfo info;
            build_bvh(t, nodes, info);
            if (nodes_ref.empty()) {
                nodes_ref = nodes;
                tris_ref = t;
                continue;
            }
            CHECK(nodes.size() == nodes_ref.size() && std::memcmp(nodes.data(), nodes_ref.data(), nodes.size() * sizeof(DevNode)) == 0,
                  "the %s-thread build's nodes differ from the one-thread build's", threads);
            CHECK(t.size() == tris_ref.size() && std::memcmp(t.data(), tris_ref.data(), t.size() * sizeof(DevTri)) == 0,
                  "the %s-thread build's triangle order differs from the one-thread build's", threads);
        }
        unsetenv("FLUX_BUILD_THREADS");
        check_mesh("grid 300x200 (threaded build)", mesh);
        if (!fails) std::printf("ok threaded build equals the serial build\n");
    }
    std::printf(fails ? "FAILED (%d)\n" : "all ok\n", fails);
    return fails ? 1 : 0;
}
