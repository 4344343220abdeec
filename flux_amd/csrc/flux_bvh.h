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
    uint64_t wide_stack = 0;  // most entries the 4-wide traversal can have stacked at once (sum of children - 1 along a path)
};

// Binned-SAH top-down build.  `tris` is reordered into leaf order (ids keep the original order).
void build_bvh(std::vector<DevTri> &tris, std::vector<DevNode> &nodes, BvhInfo &info);

// `nodes` on the 16-bit grid (fills info.qmin / qstep).  Every quantised box CONTAINS the f32 box it comes from; the
// function verifies that and returns false if it does not hold (the caller refuses the mesh).
bool quantize_bvh(const std::vector<DevNode> &nodes, std::vector<DevNodeQ> &out, BvhInfo &info);

// The 4-wide tree + leaf records of the FAST traversal kernel from the binary tree and its quantised boxes.
void build_wide(const std::vector<DevNode> &nodes, const std::vector<DevNodeQ> &nodesq, const std::vector<DevTri> &tris,
                std::vector<DevNode4Q> &wide, std::vector<DevLeafRec> &leaves, BvhInfo &info);

}  // namespace flux
