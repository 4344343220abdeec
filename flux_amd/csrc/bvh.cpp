// bvh.cpp -- host-side BVH builder for triangle meshes (extension; no reference counterpart).
//
// Top-down binned SAH (16 bins, all three axes), leaves of <= kBvhLeafSize triangles, depth bounded
// by falling back to median splits, children bounds stored in the parent as outward-rounded, padded
// f32 so that the device slab test never rejects a box whose triangle the f64 Moeller-Trumbore test
// would accept.  The traversal must return exactly the brute-force nearest hit (lowest id on ties).
#include "flux_bvh.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <functional>
#include <limits>
#include <thread>

#ifndef FLUX_BVH_BINS
#define FLUX_BVH_BINS 16
#endif
#ifndef FLUX_BVH_COLLAPSE_MODE
#define FLUX_BVH_COLLAPSE_MODE 0
#endif

namespace flux {
namespace {

struct Box {
    double lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::numeric_limits<double>::infinity();
            hi[a] = -std::numeric_limits<double>::infinity();
        }
    }
    void grow(const Box &b) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], b.lo[a]);
            hi[a] = std::max(hi[a], b.hi[a]);
        }
    }
    void grow(const double p[3]) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    double area() const {
        double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0;
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
};

// The builder partitions the primitives THEMSELVES, not an index array over them: a split then streams its range (and the lower
// levels, whose ranges fit a cache, never leave it), where `prims[order[k]]` was a cache miss per triangle per level -- the 1 M-triangle
// build was bound by exactly that (round 6).  The permutation is the same: std::partition / std::nth_element move elements by position,
// whatever their size.
struct Prim {
    Box box;
    double c[3];
    uint32_t id;  // the triangle's index in the caller's array
};

struct Builder {
    std::vector<Prim> &prims;  // shared by all builders of one build: each partitions its own, disjoint range
    std::vector<DevNode> &nodes;
    double pad;
    BvhInfo &info;

    static float down(double x) {
        float f = (float)x;
        if ((double)f > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
        return f;
    }
    static float up(double x) {
        float f = (float)x;
        if ((double)f < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
        return f;
    }
    void store_box(float lo[3], float hi[3], const Box &b) const {
        for (int a = 0; a < 3; a++) {
            lo[a] = down(b.lo[a] - pad);
            hi[a] = up(b.hi[a] + pad);
        }
    }
    static void empty_box(float lo[3], float hi[3]) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::numeric_limits<float>::infinity();
            hi[a] = -std::numeric_limits<float>::infinity();
        }
    }

    Box range_box(uint32_t b, uint32_t e) const {
        Box r;
        r.reset();
        for (uint32_t k = b; k < e; k++) r.grow(prims[k].box);
        return r;
    }

    // Chooses a split position in (b,e); partitions prims[b,e) accordingly.
    uint32_t split(uint32_t b, uint32_t e, int depth) {
        const uint32_t n = e - b;
        Box cb;
        cb.reset();
        for (uint32_t k = b; k < e; k++) cb.grow(prims[k].c);
        int best_axis = -1, best_bin = -1;
        double best_cost = std::numeric_limits<double>::infinity();
        constexpr int NB = FLUX_BVH_BINS;
        const bool allow_sah = depth < kBvhSahDepth;
        if (allow_sah) {
            for (int a = 0; a < 3; a++) {
                const double ext = cb.hi[a] - cb.lo[a];
                if (!(ext > 0.0)) continue;
                Box bb[NB];
                uint32_t cnt[NB];
                for (int i = 0; i < NB; i++) {
                    bb[i].reset();
                    cnt[i] = 0;
                }
                const double k1 = NB * (1.0 - 1e-12) / ext;
                for (uint32_t k = b; k < e; k++) {
                    const Prim &p = prims[k];
                    int bin = (int)((p.c[a] - cb.lo[a]) * k1);
                    bin = std::min(std::max(bin, 0), NB - 1);
                    bb[bin].grow(p.box);
                    cnt[bin]++;
                }
                double la[NB], ra[NB];
                uint32_t lc[NB], rc[NB];
                Box acc;
                acc.reset();
                uint32_t c = 0;
                for (int i = 0; i < NB; i++) {
                    acc.grow(bb[i]);
                    c += cnt[i];
                    la[i] = acc.area();
                    lc[i] = c;
                }
                acc.reset();
                c = 0;
                for (int i = NB - 1; i >= 0; i--) {
                    acc.grow(bb[i]);
                    c += cnt[i];
                    ra[i] = acc.area();
                    rc[i] = c;
                }
                for (int i = 0; i < NB - 1; i++) {
                    if (lc[i] == 0 || rc[i + 1] == 0) continue;
                    double cost = la[i] * lc[i] + ra[i + 1] * rc[i + 1];
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = a;
                        best_bin = i;
                    }
                }
            }
        }
        if (best_axis >= 0) {
            const int a = best_axis;
            const double ext = cb.hi[a] - cb.lo[a];
            const double k1 = NB * (1.0 - 1e-12) / ext;
            auto mid = std::partition(prims.begin() + b, prims.begin() + e, [&](const Prim &p) {
                int bin = (int)((p.c[a] - cb.lo[a]) * k1);
                bin = std::min(std::max(bin, 0), NB - 1);
                return bin <= best_bin;
            });
            uint32_t m = (uint32_t)(mid - prims.begin());
            if (m > b && m < e) return m;
        }
        // median split on the widest centroid axis (also the depth-bounding fallback)
        int a = 0;
        for (int k = 1; k < 3; k++)
            if (cb.hi[k] - cb.lo[k] > cb.hi[a] - cb.lo[a]) a = k;
        const uint32_t m = b + n / 2;
        std::nth_element(prims.begin() + b, prims.begin() + m, prims.begin() + e, [&](const Prim &x, const Prim &y) {
            return x.c[a] < y.c[a] || (x.c[a] == y.c[a] && x.id < y.id);
        });
        return m;
    }

    // Builds the subtree over prims[b,e) (more than one leaf's worth); returns its node index.
    // fork > 0 (large meshes, build_bvh): the LEFT child's subtree is built by a thread of its own -- with its own node vector and
    // statistics, appended afterwards -- while this thread goes on with the right one, both with fork - 1: level k of the tree is then
    // 2^k concurrent builders over disjoint ranges of `prims`, and the serial part of the build is the root's split, not the top five
    // or six levels.  The tree does not depend on it (a split sees its own range only; build_bvh renumbers the nodes breadth-first).
    int32_t build_inner(uint32_t b, uint32_t e, int depth, int fork = 0) {
        const int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        info.max_depth = std::max<uint64_t>(info.max_depth, (uint64_t)depth + 1);
        const uint32_t m = split(b, e, depth);
        const uint32_t rb[2] = {b, m}, re[2] = {m, e};
        std::thread left;
        std::vector<DevNode> left_nodes;
        BvhInfo left_info;
        Box left_box;
        const bool forked = fork > 0 && (m - b) >= 4096u && (m - b) > (uint32_t)kBvhLeafSize && (e - m) > (uint32_t)kBvhLeafSize;
        if (forked)
            left = std::thread([&, this] {
                Builder L{prims, left_nodes, pad, left_info};
                left_box = L.range_box(b, m);
                left_nodes.reserve((m - b) / 2 + 4);
                L.build_inner(b, m, depth + 1, fork - 1);
            });
        for (int side = forked ? 1 : 0; side < 2; side++) {
            const uint32_t cnt = re[side] - rb[side];
            const Box bx = range_box(rb[side], re[side]);
            int32_t link, count = 0;
            if (cnt <= (uint32_t)kBvhLeafSize) {
                link = ~(int32_t)((rb[side] << 3) | cnt);  // leaf reference: ~((first << 3) | count)
                count = (int32_t)cnt;
                info.max_leaf = std::max<uint64_t>(info.max_leaf, cnt);
            } else {
                link = build_inner(rb[side], re[side], depth + 1, forked ? fork - 1 : fork);
            }
            DevNode &N = nodes[me];  // re-fetch: the vector may have grown
            if (side == 0) {
                store_box(N.lo0, N.hi0, bx);
                N.child0 = link;
                N.count0 = count;
            } else {
                store_box(N.lo1, N.hi1, bx);
                N.child1 = link;
                N.count1 = count;
            }
        }
        if (forked) {
            left.join();
            const int32_t off = (int32_t)nodes.size();
            for (DevNode N : left_nodes) {
                if (N.child0 >= 0) N.child0 += off;
                if (N.child1 >= 0) N.child1 += off;
                nodes.push_back(N);
            }
            DevNode &N = nodes[me];
            store_box(N.lo0, N.hi0, left_box);
            N.child0 = off;  // the subtree's root is its first node
            N.count0 = 0;
            info.max_depth = std::max(info.max_depth, left_info.max_depth);
            info.max_leaf = std::max(info.max_leaf, left_info.max_leaf);
        }
        return me;
    }
};

}  // namespace

void build_bvh(std::vector<DevTri> &tris, std::vector<DevNode> &nodes, BvhInfo &info) {
    const auto t0 = std::chrono::steady_clock::now();
    nodes.clear();
    info = BvhInfo();
    info.tris = tris.size();
    if (tris.empty()) return;
    std::vector<Prim> prims(tris.size());
    Box all;
    all.reset();
    for (size_t k = 0; k < tris.size(); k++) {
        const DevTri &t = tris[k];
        const double v[3][3] = {{t.v0x, t.v0y, t.v0z},
                                {t.v0x + t.e1x, t.v0y + t.e1y, t.v0z + t.e1z},
                                {t.v0x + t.e2x, t.v0y + t.e2y, t.v0z + t.e2z}};
        Prim &p = prims[k];
        p.box.reset();
        for (int j = 0; j < 3; j++) p.box.grow(v[j]);
        for (int a = 0; a < 3; a++) p.c[a] = 0.5 * (p.box.lo[a] + p.box.hi[a]);
        p.id = (uint32_t)k;
        all.grow(p.box);
    }
    double diag = 0.0;
    for (int a = 0; a < 3; a++) diag = std::max(diag, all.hi[a] - all.lo[a]);
    double mag = 0.0;
    for (int a = 0; a < 3; a++) mag = std::max(mag, std::max(std::fabs(all.lo[a]), std::fabs(all.hi[a])));
    // absolute padding: far above the f64 rounding of the triangle test and the v0+e reconstruction,
    // far below anything visible in traversal cost
    const double pad = 1e-7 * std::max(std::max(diag, mag), 1e-30);

    Builder B{prims, nodes, pad, info};
    nodes.reserve(tris.size() / 2 + 4);
    // Large meshes are built by several threads (build_inner's fork; FLUX_BUILD_THREADS, default: the hardware's, at most 16): the tree,
    // hence everything that follows from it, is THE SAME as the one-thread build's (tests/bvh_selftest.cpp compares them bit for bit).
    // 1 M triangles on the GPU box's host: 420 ms with one thread (DESIGN.md "Context creation").
    unsigned threads = std::thread::hardware_concurrency();
    if (const char *env = std::getenv("FLUX_BUILD_THREADS")) threads = (unsigned)std::max(1, std::atoi(env));
    threads = std::min(std::max(threads, 1u), 16u);
    int fork = 0;
    if (tris.size() >= 65536)
        while ((1u << fork) < 2u * threads && fork < 6) fork++;  // about two leaf builders per thread (splits are not exactly even)
    if (threads <= 1) fork = 0;
    if (tris.size() <= (size_t)kBvhLeafSize) {
        nodes.emplace_back();
        DevNode &N = nodes[0];
        B.store_box(N.lo0, N.hi0, all);
        N.child0 = ~(int32_t)((0u << 3) | (uint32_t)tris.size());
        N.count0 = (int32_t)tris.size();
        Builder::empty_box(N.lo1, N.hi1);
        N.child1 = ~0;  // empty leaf, count 0.  Its inverted (+inf, -inf) box does NOT fail the symmetric min/max slab test
                        // (tn = -inf, tf = +inf): the leaf is visited and its zero triangles are tested -- harmless
        N.count1 = 0;
        info.max_depth = 1;
        info.max_leaf = tris.size();
    } else {
        B.build_inner(0, (uint32_t)tris.size(), 0, fork);
    }
    // Relabel the nodes breadth-first with siblings adjacent (built depth-first: child0 = parent + 1, child1
    // far away): the two children of a node then share one 128-B line, and the top of the tree -- touched by
    // every ray -- is one contiguous, cache-resident block.
    if (nodes.size() > 1) {
        std::vector<int32_t> order_bfs;
        order_bfs.reserve(nodes.size());
        order_bfs.push_back(0);
        for (size_t head = 0; head < order_bfs.size(); head++) {
            const DevNode &N = nodes[(size_t)order_bfs[head]];
            if (N.child0 >= 0) order_bfs.push_back(N.child0);
            if (N.child1 >= 0) order_bfs.push_back(N.child1);
        }
        std::vector<int32_t> new_index(nodes.size(), -1);
        for (size_t k = 0; k < order_bfs.size(); k++) new_index[(size_t)order_bfs[k]] = (int32_t)k;
        std::vector<DevNode> re(nodes.size());
        for (size_t k = 0; k < order_bfs.size(); k++) {
            DevNode N = nodes[(size_t)order_bfs[k]];
            if (N.child0 >= 0) N.child0 = new_index[(size_t)N.child0];
            if (N.child1 >= 0) N.child1 = new_index[(size_t)N.child1];
            re[k] = N;
        }
        nodes.swap(re);
    }
    std::vector<DevTri> sorted(tris.size());
    for (size_t k = 0; k < tris.size(); k++) sorted[k] = tris[prims[k].id];
    tris.swap(sorted);
    info.nodes = nodes.size();
    info.mag = mag;
    info.pad = pad;
    info.build_us = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(
                        std::chrono::steady_clock::now() - t0).count();
}

namespace {
uint16_t grid_lo(const BvhInfo &info, float v, int a) {  // a grid point at least one quantum below v
    if (!std::isfinite(v)) return v > 0 ? 65535 : 0;
    double q = std::floor(((double)v - (double)info.qmin[a]) / (double)info.qstep[a]) - 1.0;
    return (uint16_t)std::min(std::max(q, 0.0), 65535.0);
}
uint16_t grid_hi(const BvhInfo &info, float v, int a) {  // a grid point at least one quantum above v
    if (!std::isfinite(v)) return v > 0 ? 65535 : 0;
    double q = std::ceil(((double)v - (double)info.qmin[a]) / (double)info.qstep[a]) + 1.0;
    return (uint16_t)std::min(std::max(q, 0.0), 65535.0);
}
}  // namespace

bool quantize_bvh(const std::vector<DevNode> &nodes, std::vector<DevNodeQ> &out, BvhInfo &info) {
    out.clear();
    if (nodes.empty()) return true;
    // grid over all FINITE box corners (the one-leaf tree's empty slot is an inverted infinite box)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    auto grow = [&](const float *l, const float *h) {
        for (int a = 0; a < 3; a++) {
            if (std::isfinite(l[a])) lo[a] = std::min(lo[a], l[a]);
            if (std::isfinite(h[a])) hi[a] = std::max(hi[a], h[a]);
        }
    };
    for (const DevNode &N : nodes) {
        grow(N.lo0, N.hi0);
        grow(N.lo1, N.hi1);
    }
    // grid: value(q) = qmin + q * qstep with qmin AT LEAST ONE QUANTUM BELOW the lowest corner (so q >= 1 there) and the
    // step sized from (highest corner - qmin), so that the highest corner lands at q <= 65533 and 0 / 65534.. remain for the
    // extra quantum every box is widened by.  qmin is fixed FIRST: for a mesh far from the origin relative to its extent
    // (ulp_f32(lo) > step) rounding qmin down moves it by many quanta, and a step sized from (hi - lo) alone would push
    // the top corners past 65535, where the clamp breaks containment.
    for (int a = 0; a < 3; a++) {
        if (!(hi[a] >= lo[a])) lo[a] = hi[a] = 0.0f;
        const double L = (double)lo[a], H = (double)hi[a];
        auto upf = [](double x) {
            float f = (float)x;
            if ((double)f < x) f = std::nextafterf(f, INFINITY);
            return f;
        };
        float st = upf(std::max(H - L, 1e-30) / 65530.0);
        if (!(st > 0.0f)) st = 1e-30f;
        float org = (float)(L - (double)st);
        for (int it = 0; it < 64; ++it) {
            while ((double)org + (double)st > L) org = std::nextafterf(org, -INFINITY);
            const float need = upf((H - (double)org) / 65532.0);
            if (need <= st) break;
            st = need;  // a larger step: re-check that qmin is still a quantum below the lowest corner
        }
        info.qstep[a] = st;
        info.qmin[a] = org;
    }
    auto qlo = [&](float v, int a) { return grid_lo(info, v, a); };
    auto qhi = [&](float v, int a) { return grid_hi(info, v, a); };
    out.resize(nodes.size());
    for (size_t k = 0; k < nodes.size(); k++) {
        const DevNode &N = nodes[k];
        DevNodeQ &Q = out[k];
        for (int a = 0; a < 3; a++) {
            Q.lo0[a] = qlo(N.lo0[a], a);
            Q.hi0[a] = qhi(N.hi0[a], a);
            Q.lo1[a] = qlo(N.lo1[a], a);
            Q.hi1[a] = qhi(N.hi1[a], a);
        }
        Q.child0 = N.child0;
        Q.child1 = N.child1;
    }
    // the invariant the traversal relies on, verified: every quantised box contains its f32 box (grid values in exact
    // arithmetic; the slab test's padding covers the device's f32 evaluation, render_body.inc)
    auto val = [&](uint16_t q, int a) { return (double)info.qmin[a] + (double)q * (double)info.qstep[a]; };
    for (size_t k = 0; k < nodes.size(); k++) {
        const DevNode &N = nodes[k];
        const DevNodeQ &Q = out[k];
        for (int a = 0; a < 3; a++) {
            const bool ok0 = !std::isfinite(N.lo0[a]) || !std::isfinite(N.hi0[a]) ||
                             (val(Q.lo0[a], a) <= (double)N.lo0[a] && val(Q.hi0[a], a) >= (double)N.hi0[a]);
            const bool ok1 = !std::isfinite(N.lo1[a]) || !std::isfinite(N.hi1[a]) ||
                             (val(Q.lo1[a], a) <= (double)N.lo1[a] && val(Q.hi1[a], a) >= (double)N.hi1[a]);
            if (!ok0 || !ok1) return false;
        }
    }
    return true;
}

void build_wide(const std::vector<DevNode> &nodes, const std::vector<DevNodeQ> &nodesq, const std::vector<DevTri> &tris,
                std::vector<DevNode4Q> &wide, std::vector<DevLeafRec> &leaves, BvhInfo &info) {
    wide.clear();
    leaves.clear();
    info.wide_nodes = info.leaf_records = info.fused_leaves = info.wide_stack = 0;
    if (nodes.empty()) return;
    struct Slot {
        int32_t link;        // binary link: >= 0 inner node, < 0 leaf reference
        uint16_t lo[3], hi[3];
        float area;
    };
    auto slot_of = [&](int32_t parent, int side) {
        const DevNode &N = nodes[(size_t)parent];
        const DevNodeQ &Q = nodesq[(size_t)parent];
        Slot s;
        s.link = side ? N.child1 : N.child0;
        const float *lo = side ? N.lo1 : N.lo0, *hi = side ? N.hi1 : N.hi0;
        for (int a = 0; a < 3; a++) {
            s.lo[a] = side ? Q.lo1[a] : Q.lo0[a];
            s.hi[a] = side ? Q.hi1[a] : Q.hi0[a];
        }
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        s.area = (dx >= 0 && dy >= 0 && dz >= 0) ? 2.0f * (dx * dy + dy * dz + dz * dx) : -1.0f;
        return s;
    };
    // leaf reference of the binary tree -> records (a fused pair where the two triangles are the halves of a quad)
    auto leaf_ref = [&](int32_t link) -> int32_t {
        const int ref = ~link;
        const int first = ref >> 3, cnt = ref & 7;
        const size_t rec_first = leaves.size();
        auto single = [&](int k) {
            const DevTri &T = tris[(size_t)k];
            DevLeafRec R;
            std::memset(&R, 0, sizeof(R));
            R.v0[0] = T.v0x; R.v0[1] = T.v0y; R.v0[2] = T.v0z;
            R.e1[0] = T.e1x; R.e1[1] = T.e1y; R.e1[2] = T.e1z;
            R.e2[0] = T.e2x; R.e2[1] = T.e2y; R.e2[2] = T.e2z;
            R.id[0] = T.id; R.id[1] = -1;
            R.slot[0] = k; R.slot[1] = -1;
            leaves.push_back(R);
        };
        auto same3 = [](double ax, double ay, double az, double bx, double by, double bz) { return ax == bx && ay == by && az == bz; };
        int k = 0;
        while (k < cnt) {
            bool fused = false;
            if (k + 1 < cnt) {
                const DevTri &A = tris[(size_t)(first + k)], &B = tris[(size_t)(first + k + 1)];
                if (same3(A.v0x, A.v0y, A.v0z, B.v0x, B.v0y, B.v0z)) {
                    const DevTri *X = nullptr, *Y = nullptr;  // X = (v0, e1, e2), Y = (v0, e2, e3)
                    int sx = 0, sy = 0;
                    if (same3(A.e2x, A.e2y, A.e2z, B.e1x, B.e1y, B.e1z)) { X = &A; Y = &B; sx = first + k; sy = first + k + 1; }
                    else if (same3(B.e2x, B.e2y, B.e2z, A.e1x, A.e1y, A.e1z)) { X = &B; Y = &A; sx = first + k + 1; sy = first + k; }
                    if (X) {
                        DevLeafRec R;
                        std::memset(&R, 0, sizeof(R));
                        R.v0[0] = X->v0x; R.v0[1] = X->v0y; R.v0[2] = X->v0z;
                        R.e1[0] = X->e1x; R.e1[1] = X->e1y; R.e1[2] = X->e1z;
                        R.e2[0] = X->e2x; R.e2[1] = X->e2y; R.e2[2] = X->e2z;
                        R.e3[0] = Y->e2x; R.e3[1] = Y->e2y; R.e3[2] = Y->e2z;
                        R.id[0] = X->id; R.id[1] = Y->id;
                        R.slot[0] = sx; R.slot[1] = sy;
                        leaves.push_back(R);
                        info.fused_leaves++;
                        fused = true;
                        k += 2;
                    }
                }
            }
            if (!fused) {
                single(first + k);
                k++;
            }
        }
        const size_t nrec = leaves.size() - rec_first;  // <= cnt <= 7
        return ~(int32_t)(((uint32_t)rec_first << 3) | (uint32_t)nrec);
    };
    // leaves below every binary node (children have larger indices after the breadth-first relabelling: bottom-up pass)
    std::vector<uint32_t> leaves_below(nodes.size(), 0);
    for (size_t k = nodes.size(); k-- > 0;) {
        const DevNode &N = nodes[k];
        leaves_below[k] = (N.child0 >= 0 ? leaves_below[(size_t)N.child0] : 1u) + (N.child1 >= 0 ? leaves_below[(size_t)N.child1] : 1u);
    }
    // breadth-first over the binary nodes that survive as 4-wide nodes
    std::vector<int32_t> queue;      // binary index of wide node k
    std::vector<int32_t> wide_of(nodes.size(), -1);
    queue.push_back(0);
    wide_of[0] = 0;
    std::vector<std::array<Slot, 4>> slots_of;
    std::vector<int> count_of;
    for (size_t head = 0; head < queue.size(); head++) {
        const int32_t b = queue[head];
        Slot sl[4];
        int n = 2;
        sl[0] = slot_of(b, 0);
        sl[1] = slot_of(b, 1);
        while (n < 4) {  // open the inner child with the largest surface
            int pick = -1;
#if FLUX_BVH_COLLAPSE_MODE >= 1
            // ... but with the LAST free slot rather absorb a child whose own children are both leaves (a whole node less to visit)
            if (n == 3)
                for (int k = 0; k < n; k++)
                    if (sl[k].link >= 0 && nodes[(size_t)sl[k].link].child0 < 0 && nodes[(size_t)sl[k].link].child1 < 0 &&
                        (pick < 0 || sl[k].area > sl[pick].area))
                        pick = k;
#endif
            if (pick < 0)
                for (int k = 0; k < n; k++)
                    if (sl[k].link >= 0 && (pick < 0 || sl[k].area > sl[pick].area)) pick = k;
            if (pick < 0) break;
            const int32_t c = sl[pick].link;
            sl[pick] = slot_of(c, 0);
            sl[n++] = slot_of(c, 1);
        }
        std::array<Slot, 4> arr;
        for (int k = 0; k < 4; k++) arr[(size_t)k] = sl[k < n ? k : 0];
        slots_of.push_back(arr);
        count_of.push_back(n);
        for (int k = 0; k < n; k++)
            if (sl[k].link >= 0) {
                wide_of[(size_t)sl[k].link] = (int32_t)queue.size();
                queue.push_back(sl[k].link);
            }
    }
    wide.resize(queue.size());
    for (size_t w = 0; w < queue.size(); w++) {
        DevNode4Q &W = wide[w];
        const int n = count_of[w];
        for (int k = 0; k < 4; k++) {
            if (k < n) {
                const Slot &S = slots_of[w][(size_t)k];
                W.bx[k] = (uint32_t)S.lo[0] | ((uint32_t)S.hi[0] << 16);
                W.by[k] = (uint32_t)S.lo[1] | ((uint32_t)S.hi[1] << 16);
                W.bz[k] = (uint32_t)S.lo[2] | ((uint32_t)S.hi[2] << 16);
                // an empty leaf of the binary tree (the one-leaf tree's second slot: count 0) stays an empty slot
                if (S.link < 0 && ((~S.link) & 7) == 0) {
                    W.bx[k] = W.by[k] = W.bz[k] = 0x0000ffffu;
                    W.link[k] = (int32_t)0x80000000;
                } else {
                    W.link[k] = S.link >= 0 ? wide_of[(size_t)S.link] : leaf_ref(S.link);
                }
            } else {
                W.bx[k] = W.by[k] = W.bz[k] = 0x0000ffffu;  // lo 65535, hi 0: never hit
                W.link[k] = (int32_t)0x80000000;
            }
        }
    }
    // most stack entries at once: a visit of a node with k children leaves at most k - 1 behind
    std::vector<uint32_t> need(wide.size(), 0);
    for (size_t w = wide.size(); w-- > 0;) {  // children have larger indices (breadth-first): bottom-up
        const DevNode4Q &W = wide[w];
        int kids = 0;
        uint32_t deepest = 0;
        for (int k = 0; k < 4; k++) {
            if (W.link[k] == (int32_t)0x80000000) continue;
            kids++;
            if (W.link[k] >= 0) deepest = std::max(deepest, need[(size_t)W.link[k]]);
        }
        need[w] = (uint32_t)(kids > 0 ? kids - 1 : 0) + deepest;
    }
    info.wide_nodes = wide.size();
    info.leaf_records = leaves.size();
    info.wide_stack = need[0];
}


// ---- the arena layout (flux_bvh.h DevNode4A): children contiguous, ONE stack entry per node ------------------------------
void build_wide_arena(const std::vector<DevNode> &nodes, const std::vector<DevNodeQ> &nodesq, const std::vector<DevTri> &tris,
                      std::vector<DevNode4A> &arena, BvhInfo &info) {
    arena.clear();
    info.wide_nodes = info.leaf_records = info.fused_leaves = info.wide_stack = info.arena_units = info.split_leaves = 0;
    if (nodes.empty()) return;
    constexpr int32_t kEmpty = INT32_MIN;
    // (1) the binary tree with its leaves rewritten: a leaf link becomes ~record (ONE record: a quad's two halves or a single
    //     triangle); a leaf whose triangles need more than one record becomes a small subtree of one-record leaves, boxes from
    //     the triangles themselves (f32, rounded outward and padded like every box; on the same 16-bit grid)
    std::vector<DevNode> bn = nodes;
    std::vector<DevNodeQ> bq = nodesq;
    std::vector<DevLeafRec> recs;
    struct FBox { float lo[3], hi[3]; };
    auto rec_box = [&](const DevLeafRec &R) {
        double lo[3], hi[3];
        for (int a = 0; a < 3; a++) {
            const double v0 = R.v0[a], v1 = R.v0[a] + R.e1[a], v2 = R.v0[a] + R.e2[a];
            lo[a] = std::min(v0, std::min(v1, v2));
            hi[a] = std::max(v0, std::max(v1, v2));
            if (R.slot[1] >= 0) {
                const double v3 = R.v0[a] + R.e3[a];
                lo[a] = std::min(lo[a], v3);
                hi[a] = std::max(hi[a], v3);
            }
        }
        FBox b;
        for (int a = 0; a < 3; a++) {
            b.lo[a] = Builder::down(lo[a] - info.pad);
            b.hi[a] = Builder::up(hi[a] + info.pad);
        }
        return b;
    };
    auto same3 = [](double ax, double ay, double az, double bx, double by, double bz) { return ax == bx && ay == by && az == bz; };
    auto single = [&](int k) {
        const DevTri &T = tris[(size_t)k];
        DevLeafRec R;
        std::memset(&R, 0, sizeof(R));
        R.v0[0] = T.v0x; R.v0[1] = T.v0y; R.v0[2] = T.v0z;
        R.e1[0] = T.e1x; R.e1[1] = T.e1y; R.e1[2] = T.e1z;
        R.e2[0] = T.e2x; R.e2[1] = T.e2y; R.e2[2] = T.e2z;
        R.id[0] = T.id; R.id[1] = -1;
        R.slot[0] = k; R.slot[1] = -1;
        recs.push_back(R);
    };
    auto leaf_records = [&](int first, int cnt) {  // the records of triangles [first, first + cnt): quads where two fuse
        int k = 0;
        while (k < cnt) {
            bool fused = false;
            if (k + 1 < cnt) {
                const DevTri &A = tris[(size_t)(first + k)], &B = tris[(size_t)(first + k + 1)];
                if (same3(A.v0x, A.v0y, A.v0z, B.v0x, B.v0y, B.v0z)) {
                    const DevTri *X = nullptr, *Y = nullptr;  // X = (v0, e1, e2), Y = (v0, e2, e3)
                    int sx = 0, sy = 0;
                    if (same3(A.e2x, A.e2y, A.e2z, B.e1x, B.e1y, B.e1z)) { X = &A; Y = &B; sx = first + k; sy = first + k + 1; }
                    else if (same3(B.e2x, B.e2y, B.e2z, A.e1x, A.e1y, A.e1z)) { X = &B; Y = &A; sx = first + k + 1; sy = first + k; }
                    if (X) {
                        DevLeafRec R;
                        std::memset(&R, 0, sizeof(R));
                        R.v0[0] = X->v0x; R.v0[1] = X->v0y; R.v0[2] = X->v0z;
                        R.e1[0] = X->e1x; R.e1[1] = X->e1y; R.e1[2] = X->e1z;
                        R.e2[0] = X->e2x; R.e2[1] = X->e2y; R.e2[2] = X->e2z;
                        R.e3[0] = Y->e2x; R.e3[1] = Y->e2y; R.e3[2] = Y->e2z;
                        R.id[0] = X->id; R.id[1] = Y->id;
                        R.slot[0] = sx; R.slot[1] = sy;
                        recs.push_back(R);
                        info.fused_leaves++;
                        fused = true;
                        k += 2;
                    }
                }
            }
            if (!fused) {
                single(first + k);
                k++;
            }
        }
    };
    // subtree over records [lo, hi), hi - lo >= 2: halves; returns the new node's index
    std::function<int32_t(int, int)> subtree = [&](int lo, int hi) -> int32_t {
        const int32_t me = (int32_t)bn.size();
        bn.emplace_back();
        bq.emplace_back();
        const int mid = lo + (hi - lo) / 2;
        const int rb[2] = {lo, mid}, re[2] = {mid, hi};
        for (int side = 0; side < 2; side++) {
            FBox b;
            for (int a = 0; a < 3; a++) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
            for (int r = rb[side]; r < re[side]; r++) {
                const FBox t = rec_box(recs[(size_t)r]);
                for (int a = 0; a < 3; a++) { b.lo[a] = std::min(b.lo[a], t.lo[a]); b.hi[a] = std::max(b.hi[a], t.hi[a]); }
            }
            const int32_t link = re[side] - rb[side] == 1 ? ~(int32_t)rb[side] : subtree(rb[side], re[side]);
            DevNode &N = bn[(size_t)me];  // re-fetch: the vectors may have grown
            DevNodeQ &Q = bq[(size_t)me];
            float *nlo = side ? N.lo1 : N.lo0, *nhi = side ? N.hi1 : N.hi0;
            uint16_t *qlo = side ? Q.lo1 : Q.lo0, *qhi = side ? Q.hi1 : Q.hi0;
            for (int a = 0; a < 3; a++) {
                nlo[a] = b.lo[a];
                nhi[a] = b.hi[a];
                qlo[a] = grid_lo(info, b.lo[a], a);
                qhi[a] = grid_hi(info, b.hi[a], a);
            }
            (side ? N.child1 : N.child0) = link;
            (side ? Q.child1 : Q.child0) = link;
            (side ? N.count1 : N.count0) = 0;
        }
        return me;
    };
    const size_t n_orig = nodes.size();
    for (size_t k = 0; k < n_orig; k++)
        for (int side = 0; side < 2; side++) {
            const int32_t link = side ? nodes[k].child1 : nodes[k].child0;
            if (link >= 0) continue;
            const int ref = ~link, first = ref >> 3, cnt = ref & 7;
            int32_t nl = kEmpty;  // (the one-leaf tree's second slot: an empty leaf)
            if (cnt > 0) {
                const int r0 = (int)recs.size();
                leaf_records(first, cnt);
                const int r1 = (int)recs.size();
                if (r1 - r0 == 1) nl = ~(int32_t)r0;
                else {
                    nl = subtree(r0, r1);
                    info.split_leaves++;
                }
            }
            (side ? bn[k].child1 : bn[k].child0) = nl;
            (side ? bq[k].child1 : bq[k].child0) = nl;
        }
    // (2) collapse by surface area as build_wide does, breadth-first; (3) lay the children of every node out contiguously
    struct Slot {
        int32_t link;  // >= 0 binary node, < 0 ~record
        uint16_t lo[3], hi[3];
        float area;
    };
    auto slot_of = [&](int32_t parent, int side) {
        const DevNode &N = bn[(size_t)parent];
        const DevNodeQ &Q = bq[(size_t)parent];
        Slot s;
        s.link = side ? N.child1 : N.child0;
        const float *lo = side ? N.lo1 : N.lo0, *hi = side ? N.hi1 : N.hi0;
        for (int a = 0; a < 3; a++) {
            s.lo[a] = side ? Q.lo1[a] : Q.lo0[a];
            s.hi[a] = side ? Q.hi1[a] : Q.hi0[a];
        }
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        s.area = (dx >= 0 && dy >= 0 && dz >= 0) ? 2.0f * (dx * dy + dy * dz + dz * dx) : -1.0f;
        return s;
    };
    struct Item { int32_t b; uint32_t unit; };
    std::vector<Item> queue;
    std::vector<std::array<uint32_t, 4>> kid_items;  // per queue entry: queue index of each inner child (for the stack bound)
    std::vector<int> kid_count, inner_count;
    queue.push_back(Item{0, 0u});
    arena.resize(2);  // the root and one unit of padding: blocks start on even units
    std::memset(arena.data(), 0, 2 * sizeof(DevNode4A));
    bool too_large = false;
    for (size_t head = 0; head < queue.size() && !too_large; head++) {
        const int32_t b = queue[head].b;
        Slot sl[4];
        int n = 0;
        for (int side = 0; side < 2; side++) {
            const Slot s = slot_of(b, side);
            if (s.link != kEmpty) sl[n++] = s;
        }
        while (n < 4) {  // open the inner child with the largest surface
            int pick = -1;
            for (int k = 0; k < n; k++)
                if (sl[k].link >= 0 && (pick < 0 || sl[k].area > sl[pick].area)) pick = k;
            if (pick < 0) break;
            const int32_t c = sl[pick].link;
            const Slot s0 = slot_of(c, 0), s1 = slot_of(c, 1);
            n--;
            for (int k = pick; k < n; k++) sl[k] = sl[k + 1];
            if (s0.link != kEmpty) sl[n++] = s0;
            if (s1.link != kEmpty) sl[n++] = s1;
        }
        std::stable_sort(sl, sl + n, [](const Slot &x, const Slot &y) { return (x.link >= 0) > (y.link >= 0); });
        int n_in = 0;
        while (n_in < n && sl[n_in].link >= 0) n_in++;
        const size_t start = (arena.size() + 1) & ~(size_t)1;
        const size_t leaf0 = start + (size_t)((n_in + 1) & ~1);
        const size_t end = n > n_in ? leaf0 + 2 * (size_t)(n - n_in) : start + (size_t)n_in;
        if (end >= ((size_t)1 << 26)) { too_large = true; break; }
        arena.resize(end);
        for (size_t u = start; u < end; u++) std::memset(&arena[u], 0, sizeof(DevNode4A));
        DevNode4A W;
        std::memset(&W, 0, sizeof(W));
        std::array<uint32_t, 4> kids_q = {0, 0, 0, 0};
        for (int k = 0; k < 4; k++) {
            if (k < n) {
                const Slot &S = sl[k];
                W.bx[k] = (uint32_t)S.lo[0] | ((uint32_t)S.hi[0] << 16);
                W.by[k] = (uint32_t)S.lo[1] | ((uint32_t)S.hi[1] << 16);
                W.bz[k] = (uint32_t)S.lo[2] | ((uint32_t)S.hi[2] << 16);
                if (S.link >= 0) {
                    kids_q[(size_t)k] = (uint32_t)queue.size();
                    queue.push_back(Item{S.link, (uint32_t)(start + (size_t)k)});
                } else {
                    std::memcpy(&arena[leaf0 + 2 * (size_t)(k - n_in)], &recs[(size_t)~S.link], sizeof(DevLeafRec));
                    info.leaf_records++;
                }
            } else {
                W.bx[k] = W.by[k] = W.bz[k] = 0x0000ffffu;  // lo 65535, hi 0: never hit
            }
        }
        W.meta = ((uint32_t)n_in << 4) | ((uint32_t)(start >> 1) << 7);
        W.kids = (uint32_t)n;
        arena[queue[head].unit] = W;
        kid_items.push_back(kids_q);
        kid_count.push_back(n);
        inner_count.push_back(n_in);
    }
    if (too_large) {  // beyond the 26-bit unit index of a stack entry: the caller falls back to the binary tree's kernel
        arena.clear();
        info.leaf_records = 0;
        return;
    }
    // most stack entries at once: a visit of a node with two children or more leaves at most ONE entry behind
    std::vector<uint32_t> need(queue.size(), 0);
    for (size_t w = queue.size(); w-- > 0;) {  // children come later in the queue: bottom-up
        uint32_t deepest = 0;
        for (int k = 0; k < inner_count[w]; k++) deepest = std::max(deepest, need[(size_t)kid_items[w][(size_t)k]]);
        need[w] = (kid_count[w] >= 2 ? 1u : 0u) + deepest;
    }
    info.wide_nodes = queue.size();
    info.arena_units = arena.size();
    info.wide_stack = need[0];
}

}  // namespace flux
