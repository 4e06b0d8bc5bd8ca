// bvh.h -- the triangle hierarchy that replaces Embree / OptiX Prime
// (reference call sites: src/scene.cpp:128-154 build, :503-597 intersect(), :629-690 occluded()).
//
// Layout (HBM, built once per Scene on the host, see bvh.cpp):
//   nodes : 32-byte records {lo.xyz, a | hi.xyz, b}.  Inner node: a = index of its left child,
//           the right child is a+1 (siblings are adjacent, so both children arrive in one 64-byte
//           fetch); b = 0.  Leaf: a = first triangle slot, b = triangle count (1..4).
//   tris  : 36-byte records, 9 fp32 (three corners), stored in leaf order.
//   ids   : per triangle slot {shape id, triangle id} -- read only for accepted hits.
// These are the 32 B / 36 B units of the traversal kernel's algorithmic-byte count
// (SURVEY.md section 8d).  All boxes are padded with rt::pad_box so traversal is conservative with
// respect to the shared hit predicate in raytri.h; the result equals a brute-force scan.
#pragma once
#include "raytri.h"
#include <stdint.h>

namespace rt {

struct Node {
    float lo[3]; int a;
    float hi[3]; int b;
};
static_assert(sizeof(Node) == 32, "node must be 32 bytes");

struct BvhD {              // device/host view
    const Node *nodes;
    const float *tris;     // 9 floats per slot
    const int *ids;        // 2 ints per slot
    int num_nodes, num_tris;
    int stack_need;        // entries a traversal can need (hierarchy depth + 2); picks the kernel's LDS stack size
};

struct Counters { unsigned long long nodes, tris; };

// Ray queue records exchanged between the shading stages and the traversal kernels (HBM, dense by
// queue slot).  A dead slot has tmax < 0 and always reports a miss.
struct alignas(16) RayRec { float ox, oy, oz, tmin, dx, dy, dz, tmax; };   // 32 B
struct alignas(8) HitRec { int shape, prim; };                            // 8 B; shape < 0 = miss
static_assert(sizeof(RayRec) == 32 && sizeof(HitRec) == 8, "queue record sizes");

// One ray against the hierarchy.  ANY = stop at the first accepted hit (occlusion query).
// `cnt` (optional) tallies node records loaded and triangle records tested.
// `stack` holds kTraverseStack ints spaced `stride` apart (a private array on the host, a column of
// an LDS tile on the GPU); the builder rejects hierarchies deeper than that.
constexpr int kTraverseStack = 40;
// IDX: the stack's element type -- unsigned short when the hierarchy has < 65536 nodes (halves the LDS column).
// `fetch(i)` returns node record i by value.  The GPU kernels pass a functor that serves the first records from LDS (they
// are in breadth-first order, so those are the upper levels of the hierarchy, staged once per workgroup: every ray starts
// there and spends about half of its steps there) and the rest from global memory; the default reads bvh.nodes.
struct FetchGlobal {
    const Node *nodes;
    RT_HD Node operator()(int i) const { return nodes[i]; }
};
template <bool ANY, class IDX, class Fetch>
RT_HD inline Hit traverse_with(const BvhD &bvh, const float o[3], const float d[3], float tnear, float tfar,
                               IDX *stack, int stride, Counters *cnt, const Fetch &fetch) {
    Hit best{tfar, -1, -1};
    if (bvh.num_nodes == 0) return best;
#define RT_NODE_AT(i) fetch(i)
    const float inv[3] = {1.f / d[0], 1.f / d[1], 1.f / d[2]};
    int sp = 0;
    float tn;
    unsigned long long nn = 1, nt = 0;
    const Node root = RT_NODE_AT(0);
    if (!ray_box(o, inv, tnear, tfar, root.lo, root.hi, &tn)) { if (cnt) { cnt->nodes += nn; } return best; }
    int cur = 0;
    for (;;) {
        const Node n = RT_NODE_AT(cur);
        if (n.b > 0) {
            for (int k = 0; k < n.b; ++k) {
                int slot = n.a + k;
                const float *t = bvh.tris + 9 * slot;
                float th;
                ++nt;
                if (ray_triangle(o, d, tnear, tfar, t, t + 3, t + 6, &th)) {
                    int s = bvh.ids[2 * slot], p = bvh.ids[2 * slot + 1];
                    if (ANY) { if (cnt) { cnt->nodes += nn; cnt->tris += nt; } return Hit{th, s, p}; }
                    if (closer(th, s, p, best)) best = Hit{th, s, p};
                }
            }
        } else {
            const Node l = RT_NODE_AT(n.a), r = RT_NODE_AT(n.a + 1);
            nn += 2;
            // keep the window closed at best.t so equal-t candidates are still visited (tie-break)
            float lim = best.shape < 0 ? tfar : best.t * 1.0000004f + 1e-30f;
            float tl, tr;
            bool hl = ray_box_once(o, inv, tnear, lim, l.lo, l.hi, &tl);
            bool hr = ray_box_once(o, inv, tnear, lim, r.lo, r.hi, &tr);
            if (hl && hr) {
                int near = n.a, far = n.a + 1;
                if (tr < tl) { near = n.a + 1; far = n.a; }
                stack[sp * stride] = (IDX)far; ++sp;
                cur = near;
                continue;
            } else if (hl) { cur = n.a; continue; }
            else if (hr) { cur = n.a + 1; continue; }
        }
        // pop; entries whose box entry lies beyond the current best are re-tested lazily by their
        // children's box tests (lim shrinks), which keeps the stack to one int per entry.
        if (sp == 0) break;
        --sp;
        cur = (int)stack[sp * stride];
    }
    if (cnt) { cnt->nodes += nn; cnt->tris += nt; }
    return best;
#undef RT_NODE_AT
}
template <bool ANY, class IDX = int>
RT_HD inline Hit traverse(const BvhD &bvh, const float o[3], const float d[3], float tnear, float tfar,
                          IDX *stack, int stride, Counters *cnt = nullptr) {
    return traverse_with<ANY, IDX>(bvh, o, d, tnear, tfar, stack, stride, cnt, FetchGlobal{bvh.nodes});
}

} // namespace rt

#include <vector>
namespace rt {
// Host-side build product.
struct BvhHost {
    double inner_area = 0;            // sum of the inner nodes' half-areas: what a refit compares its result with
    std::vector<Node> nodes;
    std::vector<float> tris;
    std::vector<int> ids;
    int depth = 0;
};
struct MeshView { const float *vertices; const int *indices; int num_triangles; };
// Binned-SAH top-down build over all triangles of all shapes (shape id = position in `meshes`).
BvhHost build_bvh(const std::vector<MeshView> &meshes);
// Same builder over boxes (lo.xyz, hi.xyz per box); leaf slots hold {0, box index} in `ids`, no triangle records.
BvhHost build_box_bvh(const float *boxes, int n);
// Refit: the topology and the leaf assignment of `h` are kept, triangle records and every box are recomputed from the
// current vertex positions / boxes (children carry larger indices than their parents, so one backward sweep does it).
// Hits do not depend on the hierarchy (raytri.h), only the work per ray does: the return value is the sum of the inner
// nodes' half-areas over what it was when the hierarchy was built -- the caller rebuilds when that ratio drifts.
double refit_bvh(BvhHost &h, const std::vector<MeshView> &meshes);
double refit_box_bvh(BvhHost &h, const float *boxes);
}
