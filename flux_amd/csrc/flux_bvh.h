// flux_bvh.h -- triangle records and the flattened BVH (extension; the reference has neither
// triangles nor a BVH: fluxcore/src/scene.rs:71-74, 156-160).  DESIGN.md "Triangles and the BVH".
#pragma once
#include <stdint.h>

#include <vector>

namespace flux {

// One triangle = one 128-B line: everything Moeller-Trumbore needs plus the stored normal.
struct DevTri {
    double v0x, v0y, v0z;
    double e1x, e1y, e1z;  // v1 - v0
    double e2x, e2y, e2z;  // v2 - v0
    int32_t id;            // index in hit order: num_shapes + position in the input meshes.  Right behind the edges: the
    int32_t mat;           // intersection test then reads bytes 0..79 = five 16-B gathers.  mat: index into the materials
    double nx, ny, nz;     // normalize(e1 x e2), never flipped (read when shading only)
    double pad[3];
};
static_assert(sizeof(DevTri) == 128, "DevTri layout");

// Binary BVH node, 64 B: the bounds of BOTH children (f32, rounded outward and padded, so the
// slab test is conservative w.r.t. the f64 triangle test) and their links.
// child >= 0: inner node index; child < 0: leaf reference ~((first << 3) | count) holding triangles
// [first, first + count), count <= 7, first < 2^27.
struct DevNode {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    int32_t child0, child1;
    int32_t count0, count1;  // leaf triangle counts (0 for inner children; a leaf reference with count 0 is an empty leaf)
};
static_assert(sizeof(DevNode) == 64, "DevNode layout");

// The same node in 32 B = TWO 16-B gathers instead of four: the child boxes as 16-bit fixed point on a grid over the
// mesh's bounding box (BvhInfo::qmin / qstep; lo rounded down and hi up, then one more quantum outward), links as in
// DevNode.  The traversal state machine (render_bvh_kernel) is bound by the NUMBER of divergent gathers per node
// visit -- measured: every extra 16-B gather of the same line adds 65 ms to a 254-ms frame --
// not by their latency or their bytes, so halving them is what counts.  Quantisation costs nothing in the slab test:
// lo = qmin + q step folds into the ray's own constants (t = q (step/d) + (qmin/d - (o+delta)/d)).
struct DevNodeQ {
    uint16_t lo0[3], hi0[3];
    uint16_t lo1[3], hi1[3];
    int32_t child0, child1;
};
static_assert(sizeof(DevNodeQ) == 32, "DevNodeQ layout");

// ---- the FAST traversal kernel's own layout (round 3) -------------------------------------------------------------------
// Measured (scripts/micro/gather_rate.hip, profiles/r03_experiments): a divergent gather costs the CU's vector-memory
// pipeline per ACTIVE LANE and per 128-B LINE it has to fill -- ~2.4 cycles per lane-line served by the L2, ~9 when the
// line comes from beyond it -- while further 16-B pieces of the SAME line cost ~0.4.  The state machine is bound by exactly
// that, so the layout minimises LINES per ray segment:
//   * 4-wide nodes of 64 B (two per line, never straddling): the binary SAH tree collapsed by surface area, child boxes on
//     the same 16-bit grid as DevNodeQ -- half the visits of the binary tree, one line each;
//   * leaf records of 128 B that hold TWO triangles when they share v0 and an edge (tri A = v0,e1,e2; tri B = v0,e2,e3:
//     the two halves of a quad, as every leaf of a triangulated grid is) -- one line instead of two, the same operands
//     bit for bit as the DevTri records, so the exact f64 test is unchanged.
struct DevNode4Q {
    uint32_t bx[4], by[4], bz[4];  // per child and axis: lo | hi << 16 on the grid (an empty slot: lo 65535, hi 0)
    int32_t link[4];               // >= 0: DevNode4Q index; < 0: ~((first_record << 3) | records), 0x80000000 = empty slot
};
static_assert(sizeof(DevNode4Q) == 64, "DevNode4Q layout");
struct DevLeafRec {
    double v0[3], e1[3], e2[3], e3[3];  // triangle A = (v0, e1, e2), triangle B = (v0, e2, e3)
    int32_t id[2];                      // hit-order indices (DevTri::id)
    int32_t slot[2];                    // DevTri indices (normal + material at shading time); slot[1] < 0: no triangle B
    double pad[2];
};
static_assert(sizeof(DevLeafRec) == 128, "DevLeafRec layout");

// ---- round 5: ONE stack entry per NODE instead of one per deferred child ------------------------------------
// The per-lane LDS stack of the link layout above holds one entry per hit-but-deferred CHILD, so its bound is the sum of
// (children - 1) along a path: 32 entries = 8 KiB per wave on the 1 M-triangle field, 7 LDS granules of 1 280 B, 18 waves per
// CU instead of the 20 the kernel's registers allow.  Here a node's children are CONTIGUOUS in one arena of 64-B units --
// inner children first (one unit each), then the leaf records (two units each, 128-B aligned) -- so a child's link is
// arithmetic on (first unit, slot) and ONE 32-bit entry per node, `first | number of inner children | mask of the pending
// slots`, replaces up to three: the bound drops to the number of nodes with >= 2 children on a path (the wide tree's depth).
//   unit 0: the root; its children block starts at unit 2 (blocks start on even units).
//   leaf child = exactly ONE record: a binary leaf whose two triangles do not fuse into a quad becomes a node of its own
//   with two one-triangle leaves (boxes from the triangles, on the same grid).
struct DevNode4A {
    uint32_t bx[4], by[4], bz[4];  // as DevNode4Q; slots sorted: inner children, leaf children, empty slots (lo 65535, hi 0)
    uint32_t meta;                 // bits 0-3: 0 (the pending mask of a stack entry), 4-6: inner children, 7-31: first child unit >> 1
    uint32_t kids;                 // occupied slots (host-side checks; the device reads `meta` only)
    uint32_t pad[2];
};
static_assert(sizeof(DevNode4A) == 64, "DevNode4A layout");
// link of slot k of a node / stack entry e: >= 0 the unit of an inner node, < 0: ~(unit of a leaf record)
#ifdef __HIP__  // (hipcc compiles bvh.cpp as HIP too, without hip_runtime.h: attributes spelled out)
#define FLUX_BVH_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define FLUX_BVH_HD inline
#endif
// (written without a branch: inner child f + k, leaf record ~(f + 2k - (n & 6)): the leaf's extra term and the complement are
// applied under an all-ones mask -- the device compiler turned the plain ternary into two exec-masked blocks per visit)
FLUX_BVH_HD int32_t wide_link(uint32_t e, uint32_t k) {
    const uint32_t t = e >> 4, n = t & 7u, f = (e >> 6) & ~1u;
    const uint32_t leaf = k >= n ? 0xffffffffu : 0u;
    return (int32_t)((f + k + ((k - (t & 6u)) & leaf)) ^ leaf);
}

constexpr int kBvhSahDepth = 32;    // below this depth the builder uses SAH, deeper: median splits
constexpr int kBvhMaxDepth = 64;    // hard limit; the per-lane LDS stack is sized max_depth entries
#ifndef FLUX_BVH_LEAF
#define FLUX_BVH_LEAF 2
#endif
constexpr int kBvhLeafSize = FLUX_BVH_LEAF;  // <= 7: leaf references carry the count in 3 bits

struct BvhInfo {
    uint64_t nodes = 0, tris = 0, max_depth = 0, max_leaf = 0, build_us = 0;
    double mag = 0.0;  // largest |coordinate| of any vertex (scale of the f32 slab test's padding)
    float qmin[3] = {0, 0, 0}, qstep[3] = {1, 1, 1};  // the 16-bit grid of DevNodeQ: coordinate = qmin + q * qstep
    uint64_t wide_nodes = 0, leaf_records = 0, fused_leaves = 0;
    uint64_t wide_stack = 0;  // most entries the 4-wide traversal can have stacked at once (link layout: sum of children - 1
                              // along a path; arena layout: nodes with two children or more along a path)
    double pad = 0.0;         // absolute padding of every f32 box (build_bvh)
    uint64_t arena_units = 0, split_leaves = 0;  // arena layout: 64-B units; binary leaves that became a node of one-triangle leaves
};

// Binned-SAH top-down build.  `tris` is reordered into leaf order (ids keep the original order).
void build_bvh(std::vector<DevTri> &tris, std::vector<DevNode> &nodes, BvhInfo &info);

// `nodes` on the 16-bit grid (fills info.qmin / qstep).  Every quantised box CONTAINS the f32 box it comes from; the
// function verifies that and returns false if it does not hold (the caller refuses the mesh).
bool quantize_bvh(const std::vector<DevNode> &nodes, std::vector<DevNodeQ> &out, BvhInfo &info);

// The 4-wide tree + leaf records of the FAST traversal kernel from the binary tree and its quantised boxes.
void build_wide(const std::vector<DevNode> &nodes, const std::vector<DevNodeQ> &nodesq, const std::vector<DevTri> &tris,
                std::vector<DevNode4Q> &wide, std::vector<DevLeafRec> &leaves, BvhInfo &info);

// The same tree in the arena layout (DevNode4A above): `arena` holds nodes (1 unit) and leaf records (2 units) in 64-B units.
void build_wide_arena(const std::vector<DevNode> &nodes, const std::vector<DevNodeQ> &nodesq, const std::vector<DevTri> &tris,
                      std::vector<DevNode4A> &arena, BvhInfo &info);

}  // namespace flux
