// bvh_gpu.h -- the hierarchy of bvh.h built / refitted / widened by kernels (bvh_gpu.cpp); what a Scene keeps of it.
#pragma once
#include "bvh.h"
#include <memory>
#include <vector>

namespace rt {

struct BvhBuildParams { int leaf_max = 4, bins = 64; float trav_cost = 0.3f; };     // bvh.cpp: Builder's knobs

struct BvhDev {
    // device arrays.  A refitted tree owns nodes / tris / wide / area and shares the rest with the tree it was refitted from.
    Node *nodes = nullptr; float *tris = nullptr; int *ids = nullptr; Node4 *wide = nullptr;
    int *wide_src = nullptr;        // 4 per wide record: the binary records its child boxes are copies of (-1: no child)
    int *prim_ids = nullptr;        // {shape, triangle} per primitive, in the order the build numbered them
    double *area = nullptr;         // sum of the inner half-areas as of the last build / refit
    const void *shapes = nullptr;   // the ShapeRef table the triangle records were gathered from
    int num_nodes = 0, num_slots = 0, depth = 0, num_wide = 0, wide_stack_need = 0;
    std::vector<int> level_first;   // node index at which each level starts, then num_nodes
    double inner_area = 0;          // ... when the topology was built
    std::shared_ptr<const BvhDev> parent;
    std::vector<void *> owned;
    BvhDev() = default;
    BvhDev(const BvhDev &) = delete;
    BvhDev &operator=(const BvhDev &) = delete;
    ~BvhDev();
    BvhD view() const {
        BvhD v{nodes, tris, ids, num_nodes, num_slots, depth + 2};
        v.wide = wide; v.num_wide = num_wide; v.wide_stack_need = wide_stack_need;
        return v;
    }
};

// d_shapes: device table of {const float *vertices; const int *indices;} per shape; h_prim_ids: {shape, triangle} per primitive
void build_tri_bvh_device(const void *d_shapes, const int *h_prim_ids, int n, const BvhBuildParams &prm, BvhDev &out);
void build_box_bvh_device(const float *d_boxes, int n, const BvhBuildParams &prm, BvhDev &out);
void refit_tri_bvh_device(const BvhDev &src, const void *d_shapes, BvhDev &out);
void refit_box_bvh_device(const BvhDev &src, const float *d_boxes, BvhDev &out);      // boxes indexed by src.ids[2 slot + 1]

}
