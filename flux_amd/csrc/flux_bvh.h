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
};

// Binned-SAH top-down build.  `tris` is reordered into leaf order (ids keep the original order).
void build_bvh(std::vector<DevTri> &tris, std::vector<DevNode> &nodes, BvhInfo &info);

// `nodes` on the 16-bit grid (fills info.qmin / qstep).  Every quantised box CONTAINS the f32 box it comes from; the
// function verifies that and returns false if it does not hold (the caller refuses the mesh).
bool quantize_bvh(const std::vector<DevNode> &nodes, std::vector<DevNodeQ> &out, BvhInfo &info);

}  // namespace flux
