// edges.h -- mesh edges and the two edge hierarchies used by the edge-sampling estimators.
//
// Behavioural spec: Edge + silhouette predicates src/edge.h:13-241; edge list construction,
// primary PMF/CDF src/edge.cpp:43-334; EdgeTree (3-D hierarchy over camera-silhouette edges, 6-D
// position x Hough-normal hierarchy over the rest) src/edge_tree.cpp:14-882, src/aabb.h.
//
// Layout: both hierarchies use one 128-byte node record (the 3-D tree leaves d_* unused) in a
// single array [internal nodes | leaves]; links are array indices instead of pointers, so the
// structure can be uploaded to HBM verbatim.  A node reference is an int: bit 30 selects the
// tree (0 = camera-silhouette, 1 = the rest), the low bits index that tree's array.
//
// Which edge a sample lands on depends on the exact order of the edge list and on the exact
// topology of the trees (Morton ties, treelet restructuring), so edges.cpp reproduces the
// reference's build step by step, including the behaviour of the sequential sort it runs on CPU.
#pragma once
#include "surface.h"
#include "bvh.h"
#include "bvh_gpu.h"
#include <vector>

namespace rdr {

struct EdgeD { int shape_id, v0, v1, f0, f1; };

struct EdgeNode {
    V3 p_min, p_max;     // spatial bounds
    V3 d_min, d_max;     // Hough-space bounds (6-D tree only)
    double wlen;         // sum of length * exterior dihedral angle below this node
    double cost;         // SAH cost used by the treelet optimiser
    int parent, child0, child1, edge_id;   // edge_id >= 0 marks a leaf
};

// The arithmetic of the hierarchy build that decides its topology, shared by the host builder (edges.cpp) and the gfx950
// kernels (edges_gpu.cpp) so that both round alike: minima / maxima with std::min / std::max's operand order, the surface
// area the treelet pass prices boxes with (src/edge_tree.cpp:14-23, src/aabb.h).
RDR_FN double dmin_std(double a, double b) { return (b < a) ? b : a; }
RDR_FN double dmax_std(double a, double b) { return (a < b) ? b : a; }
RDR_FN float fminf_std(float a, float b) { return (b < a) ? b : a; }
RDR_FN float fmaxf_std(float a, float b) { return (a < b) ? b : a; }
RDR_FN double edge_node_area(const EdgeNode &nd, bool is3d) {
    V3 dp = nd.p_max - nd.p_min;
    if (is3d) return 2 * (dp.x * dp.y + dp.x * dp.z + dp.y * dp.z);
    V3 dd = nd.d_max - nd.d_min;
    return 2 * ((dp.x * dp.y + dp.x * dp.z + dp.y * dp.z) + (dd.x * dd.y + dd.x * dd.z + dd.y * dd.z));
}
RDR_FN void edge_node_merge(EdgeNode &dst, const EdgeNode &a, const EdgeNode &b, bool is3d) {
    const V3 pl = V3{dmin_std(a.p_min.x, b.p_min.x), dmin_std(a.p_min.y, b.p_min.y), dmin_std(a.p_min.z, b.p_min.z)};
    const V3 ph = V3{dmax_std(a.p_max.x, b.p_max.x), dmax_std(a.p_max.y, b.p_max.y), dmax_std(a.p_max.z, b.p_max.z)};
    dst.p_min = pl; dst.p_max = ph;
    if (!is3d) {
        const V3 dl = V3{dmin_std(a.d_min.x, b.d_min.x), dmin_std(a.d_min.y, b.d_min.y), dmin_std(a.d_min.z, b.d_min.z)};
        const V3 dh = V3{dmax_std(a.d_max.x, b.d_max.x), dmax_std(a.d_max.y, b.d_max.y), dmax_std(a.d_max.z, b.d_max.z)};
        dst.d_min = dl; dst.d_max = dh;
    }
}

// What the samplers read: one 128-byte line per INTERIOR node holding everything a traversal step needs about
// both children -- their spatial bounds, Hough x-interval, weight and reference -- plus the node's own bounds
// (for the "shading point inside this node" test).  Leaves are not nodes here: a leaf child is a negative
// reference, ~edge_id, and its bounds/weight sit in its parent like any child's.  One dependent fetch per level
// instead of three (self, child 0, child 1 with the 128-byte EdgeNode), and no fetch at all for leaves.
//  * spatial bounds are unions of fp32 vertex coordinates, hence exact in fp32;
//  * of the Hough-space bounds only the x interval can ever decide the reference's sphere/box test: its loop
//    returns at the first axis whose partial distance is within the radius, and partial sums only grow, so the
//    verdict is the x term's (src/aabb.h:158-170).  [quirk]
struct EdgeNodeP {
    float p_min[3], p_max[3];            // own bounds
    float c_pmin[2][3], c_pmax[2][3];    // children
    double c_dx_min[2], c_dx_max[2];
    double c_wlen[2];
    int c_ref[2];                        // >= 0: interior node index (tree bit NOT included); < 0: ~edge_id
};
static_assert(sizeof(EdgeNodeP) == 128, "EdgeNodeP must stay one cache line");
RDR_FN V3 v3_of(const float *p) { return V3{(double)p[0], (double)p[1], (double)p[2]}; }

constexpr int kEdgeTreeBit = 1 << 30;

// Everything the NEE-mode gather needs about one edge (80 B, stored in the leaf-slot order of the billboard hierarchy so
// that a leaf's 1..4 candidates arrive with one contiguous fetch and nothing else has to be chased): the Hough x-interval
// of the edge's leaf in the reference's 6-D tree ((-inf, +inf) for edges of the 3-D tree, which is never Hough-tested),
// the edge's geometry record (end points -- its spatial bounds are their component-wise min / max -- and the third corner
// of each adjacent face, as in EdgeGeom), its id and its position in the reference's leaf order.
struct GatherLeaf {
    double dx_lo, dx_hi;
    float v0[3], v1[3], o0[3], o1[3];
    int eid, rank;
    short f0, f1;            // -1: no such face, else 0 (only the sign is used)
    int has_normals;
};
static_assert(sizeof(GatherLeaf) == 80, "GatherLeaf must stay 80 bytes");
// The record of a leaf slot whose (canonical) edge is not in the current edge list: its Hough interval is empty, so the test
// every candidate passes first (sphere_box_x against the edge's own leaf interval) rejects it for every query; its end points
// lie where no NEE segment reaches.
RDR_FN GatherLeaf dead_gather_leaf() {
    GatherLeaf gl;
    gl.dx_lo = 1e300 * 1e300; gl.dx_hi = -gl.dx_lo;            // (+inf, -inf)
    for (int k = 0; k < 3; ++k) { gl.v0[k] = gl.v1[k] = gl.o0[k] = gl.o1[k] = 1e30f; }
    gl.eid = -1; gl.rank = 0x7fffffff; gl.f0 = gl.f1 = -1; gl.has_normals = 0;
    return gl;
}

// One positive-weight leaf found by the NEE-mode gather (stages_edge.h: SecEdgeGatherN), replayed in `rank` order.
struct GatherCand { int rank, eid; double w; };
constexpr int kGatherCands = 8;          // per slot; a slot that finds more falls back to the reference-order walk

struct EdgeGeom;
struct EdgeSceneD {
    const EdgeD *edges;
    const EdgeGeom *geom;                    // one per edge, same order
    int num_edges;
    const double *primary_pmf, *primary_cdf;
    const EdgeNodeP *cs_nodes, *ncs_nodes;   // interior nodes; may be null for a one-edge tree
    int cs_root, ncs_root;                   // root references; kNoEdgeTree when the tree is empty
    double edge_bounds_expand;
    int max_stack;           // entries the NEE-mode traversal can need: deepest leaf level + 1
    V3 cam_org;
    const float *ltc;                         // tabM, 128 x 128 x 9
    // NEE-mode pick as an order-free gather (see SecEdgeGatherN): a spatial hierarchy over the billboard boxes of ALL
    // edges (rt::Node records; leaf slot s holds edge id gather.ids[2 s + 1]), the Hough x-interval of each edge's own
    // leaf in the reference's 6-D tree ((-inf, +inf) for edges of the 3-D tree, which is never Hough-tested), and each
    // edge's position in the order the reference's traversal reaches the leaves (for the reservoir replay).
    rt::BvhD gather;
    const GatherLeaf *gleaf;                  // one per leaf slot of `gather`, in slot order
};

constexpr int kNoEdgeTree = 0x7fffffff;
// Interior references carry the tree in bit 30 (set = the 6-D tree of non-camera-silhouette edges).
RDR_FN const EdgeNodeP &edge_node(const EdgeSceneD &es, int ref) {
    return (ref & kEdgeTreeBit) ? es.ncs_nodes[ref & (kEdgeTreeBit - 1)] : es.cs_nodes[ref];
}

// ---- fp32 vertex helpers (comparisons and lengths are done in float, like the reference) --------
struct F3 { float x, y, z; };
RDR_FN F3 f3_vertex(const ShapeD &sh, int i) { return F3{sh.vertices[3 * i], sh.vertices[3 * i + 1], sh.vertices[3 * i + 2]}; }
RDR_FN bool f3_ne(F3 a, F3 b) { return a.x != b.x || a.y != b.y || a.z != b.z; }
RDR_FN bool f3_eq(F3 a, F3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
RDR_FN V3 to_v3(F3 a) { return V3{(double)a.x, (double)a.y, (double)a.z}; }
RDR_FN float f3_distance(F3 a, F3 b) {
    float dx = b.x - a.x, dy = b.y - a.y, dz = b.z - a.z;
    return sqrtf(dx * dx + dy * dy + dz * dz);
}

RDR_FN F3 edge_v0f(const ShapeD *shapes, const EdgeD &e) { return f3_vertex(shapes[e.shape_id], e.v0); }
RDR_FN F3 edge_v1f(const ShapeD *shapes, const EdgeD &e) { return f3_vertex(shapes[e.shape_id], e.v1); }
RDR_FN V3 edge_v0(const ShapeD *shapes, const EdgeD &e) { return to_v3(edge_v0f(shapes, e)); }
RDR_FN V3 edge_v1(const ShapeD *shapes, const EdgeD &e) { return to_v3(edge_v1f(shapes, e)); }

// third corner of face f0 (by index) / of face f1 (by position, because merged duplicates may
// reference different vertex ids; src/edge.h:88-126)
RDR_FN F3 edge_opp0f(const ShapeD *shapes, const EdgeD &e) {
    const ShapeD &sh = shapes[e.shape_id];
    for (int i = 0; i < 3; ++i) {
        int vi = sh.indices[3 * e.f0 + i];
        if (vi != e.v0 && vi != e.v1) return f3_vertex(sh, vi);
    }
    return edge_v0f(shapes, e);
}
RDR_FN F3 edge_opp1f(const ShapeD *shapes, const EdgeD &e) {
    const ShapeD &sh = shapes[e.shape_id];
    F3 a = edge_v0f(shapes, e), b = edge_v1f(shapes, e);
    for (int i = 0; i < 3; ++i) {
        F3 v = f3_vertex(sh, sh.indices[3 * e.f1 + i]);
        if (f3_ne(v, a) && f3_ne(v, b)) return v;
    }
    return b;
}

RDR_FN V3 edge_n0(const ShapeD *shapes, const EdgeD &e) {
    V3 a = edge_v0(shapes, e), b = edge_v1(shapes, e), o = to_v3(edge_opp0f(shapes, e));
    V3 n = cross(a - o, b - o);
    double l2 = len_sq(n);
    if (l2 < 1e-20) return v3(0);
    return n / sqrt(l2);
}
RDR_FN V3 edge_n1(const ShapeD *shapes, const EdgeD &e) {
    V3 a = edge_v0(shapes, e), b = edge_v1(shapes, e), o = to_v3(edge_opp1f(shapes, e));
    V3 n = cross(b - o, a - o);
    double l2 = len_sq(n);
    if (l2 < 1e-20) return v3(0);
    return n / sqrt(l2);
}

// Is the edge a silhouette as seen from point p?  (src/edge.h:155-204)
RDR_FN bool edge_is_silhouette(const ShapeD *shapes, V3 p, const EdgeD &e) {
    V3 a = edge_v0(shapes, e), b = edge_v1(shapes, e);
    if (e.f0 == -1 || e.f1 == -1) {
        if (e.f0 != -1) {
            V3 o = to_v3(edge_opp0f(shapes, e));
            if (len_sq(cross(a - o, b - o)) < 1e-20) return false;
        }
        if (e.f1 != -1) {
            V3 o = to_v3(edge_opp1f(shapes, e));
            if (len_sq(cross(b - o, a - o)) < 1e-20) return false;
        }
        return true;
    }
    V3 o0 = to_v3(edge_opp0f(shapes, e)), o1 = to_v3(edge_opp1f(shapes, e));
    V3 n0 = cross(a - o0, b - o0), n1 = cross(b - o1, a - o1);
    double l0 = len_sq(n0), l1 = len_sq(n1);
    if (l0 < 1e-20 || l1 < 1e-20) return false;
    n0 = n0 / sqrt(l0); n1 = n1 / sqrt(l1);
    if (!shapes[e.shape_id].normals) return !(dot(n0, n1) >= 1 - 1e-6f);
    bool ff0 = dot(p - o0, n0) > 0.f, ff1 = dot(p - o1, n1) > 0.f;
    return (ff0 && !ff1) || (!ff0 && ff1);
}

// Everything the samplers need about one edge in one 64-byte record (built per Scene from the same fp32 vertex
// data): the two end points and the third corner of each adjacent face.  Replaces the chain
// EdgeD -> ShapeD -> index buffer -> vertex buffer (four dependent fetches) in the edge-pick leaf tests.
struct EdgeGeom {
    float v0[3], v1[3], o0[3], o1[3];
    int f0, f1;              // -1: no such face
    int has_normals;         // the shape has shading normals (silhouette = front/back-facing flip)
    int pad;
};
static_assert(sizeof(EdgeGeom) == 64, "EdgeGeom must stay 64 bytes");

// edge_is_silhouette on an EdgeGeom (same arithmetic, src/edge.h:155-204)
RDR_FN bool edge_is_silhouette_g(const EdgeGeom &g, V3 p) {
    V3 a = V3{(double)g.v0[0], (double)g.v0[1], (double)g.v0[2]}, b = V3{(double)g.v1[0], (double)g.v1[1], (double)g.v1[2]};
    V3 o0 = V3{(double)g.o0[0], (double)g.o0[1], (double)g.o0[2]}, o1 = V3{(double)g.o1[0], (double)g.o1[1], (double)g.o1[2]};
    if (g.f0 == -1 || g.f1 == -1) {
        if (g.f0 != -1) { if (len_sq(cross(a - o0, b - o0)) < 1e-20) return false; }
        if (g.f1 != -1) { if (len_sq(cross(b - o1, a - o1)) < 1e-20) return false; }
        return true;
    }
    V3 n0 = cross(a - o0, b - o0), n1 = cross(b - o1, a - o1);
    double l0 = len_sq(n0), l1 = len_sq(n1);
    if (l0 < 1e-20 || l1 < 1e-20) return false;
    n0 = n0 / sqrt(l0); n1 = n1 / sqrt(l1);
    if (!g.has_normals) return !(dot(n0, n1) >= 1 - 1e-6f);
    bool ff0 = dot(p - o0, n0) > 0.f, ff1 = dot(p - o1, n1) > 0.f;
    return (ff0 && !ff1) || (!ff0 && ff1);
}

RDR_FN double edge_exterior_dihedral(const ShapeD *shapes, const EdgeD &e) {
    double a = double(M_PI);
    if (e.f1 != -1) {
        double c = dot(edge_n0(shapes, e), edge_n1(shapes, e));
        c = c < -1.0 ? -1.0 : (c > 1.0 ? 1.0 : c);
        a = acos(c);
    }
    return a;
}

// ---- host-side container -------------------------------------------------------------------------
struct Scene;
struct EdgeData {
    std::vector<EdgeD> edges;
    std::vector<double> primary_pmf, primary_cdf;
    std::vector<EdgeNode> cs_nodes, ncs_nodes;   // [internal | leaves]
    int cs_leaves = 0, ncs_leaves = 0;
    int max_stack = 2;             // see EdgeSceneD::max_stack
    double edge_bounds_expand = 0;
    rt::BvhHost gather;            // see EdgeSceneD::gather (CPU debugging harness; the GPU build: gather_dev)
    // GPU build: the billboard hierarchy is built / refitted by kernels (bvh_gpu.cpp) from these boxes of the CANONICAL edges;
    // gather_cur_of: canonical edge -> current edge id (-1: not in the list)
    std::vector<float> gather_boxes; std::vector<int> gather_cur_of; std::vector<EdgeD> gather_canon;
    std::shared_ptr<rt::BvhDev> gather_dev;
    bool gather_refit_allowed = true;
    std::vector<GatherLeaf> gleaf;
    std::vector<EdgeGeom> geom;                  // per edge, what EdgeSceneD::geom will hold
    std::vector<EdgeNodeP> cs_fat, ncs_fat;      // the samplers' interior-node records (EdgeSceneD::cs_nodes / ncs_nodes)
    // GPU build of the two hierarchies (edges_gpu.cpp): the host prepares the edge ids of each tree (ascending) and the
    // per-edge weight; the kernels leave the node arrays on the device (dev_nodes, [interior | leaves] like cs_nodes)
    bool device_trees = false;
    std::vector<int> cs_ids, ncs_ids;
    std::vector<double> wlen;
    EdgeNode *dev_nodes[2] = {nullptr, nullptr};
    int dev_n[2] = {0, 0};
    std::vector<void *> owned;                   // device allocations (pool blocks), released by delete_edge_data
    EdgeSceneD d;            // device view (pointers valid after publish_edge_data)
};
// The build in two steps, both run by the edge-builder thread beside the caller (scene.cpp): everything computed on the
// host, reading only the Scene's host mirrors; then the device copies and the hierarchy kernels on that thread's stream.
EdgeData *compute_edge_data(const Scene &scene);
void publish_edge_data(EdgeData &ed);
// edges_gpu.cpp (not part of the CPU debugging harness): both hierarchies, leaf order, sampler and gather records on the
// calling thread's stream; and the node arrays back on the host for rdr_debug_dump_edges.
void build_edge_trees_device(EdgeData &ed);
void drop_gather_cache();                        // the billboard hierarchy kept for the next Scene's refit (rdr_trim_cache)
void gather_hierarchy_device(EdgeData &ed);      // edges_gpu.cpp: EdgeSceneD::gather from ed.gather_boxes (build or refit, by kernels)
void download_edge_trees(EdgeData &ed);
void delete_edge_data(EdgeData *e);

} // namespace rdr
