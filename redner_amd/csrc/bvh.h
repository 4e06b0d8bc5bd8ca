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

// The 4-wide form of the same hierarchy (bvh.cpp: collapse_wide), what the gfx950 ray kernels walk: one 128-byte record
// = one cache line per interior node, holding the boxes of up to four children (one axis of the four per 16-byte load) and a
// link per child.  A ray then takes half as many dependent steps as on the binary records -- one fetch per step instead of
// the node's link words + its two children -- and a step tests four boxes with all of a step's loads in flight together.
//   link >= 0          : index of the child's own record
//   link <  0          : a leaf: first triangle slot (link & 0x7fffffff) >> 2, triangle count (link & 3) + 1
//   link == kEmptyLink : no child in this place (the places are filled from 0 up)
// Boxes are the padded boxes of the binary records, so the walk is conservative with respect to raytri.h all the same.
struct alignas(16) Node4 {
    float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
    int link[4];
    int aux[4];            // aux[0]: number of children; the rest unused (pads the record to a cache line)
};
static_assert(sizeof(Node4) == 128, "wide node must be one 128-byte line");
constexpr int kEmptyLink = 0x7fffffff;
RT_HD inline int leaf_link(int slot, int count) { return (int)(0x80000000u | ((unsigned)slot << 2) | (unsigned)(count - 1)); }

struct BvhD {              // device/host view
    const Node *nodes;
    const float *tris;     // 9 floats per slot
    const int *ids;        // 2 ints per slot
    int num_nodes, num_tris;
    int stack_need;        // entries a traversal can need (hierarchy depth + 2); picks the kernel's LDS stack size
    const Node4 *wide = nullptr;   // 4-wide records (null: only the binary form exists, e.g. the edge gather's hierarchy)
    int num_wide = 0;
    int wide_stack_need = 0;       // entries a walk of the wide records can need
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
    RT_HD void pair(int i, Node &l, Node &r) const { l = nodes[i]; r = nodes[i + 1]; }      // the two children of an inner record
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
            Node l, r;
            fetch.pair(n.a, l, r);         // siblings are adjacent: one decision where they are read from (trace.hip: FetchStaged)
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

// ---- the walk over the 4-wide records (Node4) ----------------------------------------------------------------------------
// One ray per lane; a step fetches ONE 128-byte record (one dependent round trip), tests its four child boxes, orders the
// children that are hit by entry distance (a five-exchange network), pushes the farther ones and descends into the nearest;
// a leaf child carries its triangle range in the link, so its triangles are tested without another fetch.  Stack entries
// are links (32 bit), `stride` apart.  Hits are decided by ray_triangle / closer alone (raytri.h), so the result equals the
// binary walk's and the brute-force rule's whatever the order of the visits.
template <bool ANY>
RT_HD inline Hit traverse_wide(const BvhD &bvh, const float o[3], const float d[3], float tnear, float tfar,
                              int *stack, int stride, Counters *cnt) {
    Hit best{tfar, -1, -1};
    if (bvh.num_wide == 0) return best;
    const float inv[3] = {1.f / d[0], 1.f / d[1], 1.f / d[2]};
    unsigned long long nn = 0, nt = 0;
    int sp = 0, cur = 0;
    typedef float F2 __attribute__((vector_size(8)));
    const F2 oo[3] = {F2{o[0], o[0]}, F2{o[1], o[1]}, F2{o[2], o[2]}}, ii[3] = {F2{inv[0], inv[0]}, F2{inv[1], inv[1]}, F2{inv[2], inv[2]}};
    for (;;) {
        if (cur >= 0) {
            const Node4 nd = bvh.wide[cur];          // eight 16-byte loads, issued together
            ++nn;
            const float lim = best.shape < 0 ? tfar : best.t * 1.0000004f + 1e-30f;      // closed at best.t (tie-break by id)
            float t[4]; int l[4] = {nd.link[0], nd.link[1], nd.link[2], nd.link[3]};
            int nh = 0;
            // slab distances of the four children, two children per operation: on gfx950 a two-float vector subtract /
            // multiply is ONE packed instruction (v_pk_add_f32 / v_pk_mul_f32), the same IEEE result per component
            float ax4[4], bx4[4], ay4[4], by4[4], az4[4], bz4[4];
#define RT_SLAB(dst, src, k)                                                                              \
            { const F2 p0 = (F2{src[0], src[1]} - oo[k]) * ii[k], p1 = (F2{src[2], src[3]} - oo[k]) * ii[k]; \
              dst[0] = p0[0]; dst[1] = p0[1]; dst[2] = p1[0]; dst[3] = p1[1]; }
            RT_SLAB(ax4, nd.lox, 0) RT_SLAB(bx4, nd.hix, 0) RT_SLAB(ay4, nd.loy, 1) RT_SLAB(by4, nd.hiy, 1) RT_SLAB(az4, nd.loz, 2) RT_SLAB(bz4, nd.hiz, 2)
#undef RT_SLAB
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float ax = ax4[c], bx = bx4[c], ay = ay4[c], by = by4[c], az = az4[c], bz = bz4[c];
                // NaN (0 * inf) must not cull: fmaxf / fminf return the non-NaN operand (raytri.h: ray_box_once)
                float en = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
                float ex = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
                en -= fabsf(en) * 4e-7f;
                ex *= 1.0000004f;
                const float t0 = fmaxf(tnear, en), t1 = fminf(lim, ex);
                const bool hit = l[c] != kEmptyLink && t0 <= t1;
                t[c] = hit ? t0 : INFINITY;
                nh += hit ? 1 : 0;
            }
            if (nh > 0) {
                {                                       // misses carry t = inf and sink to the end
#define RT_CSWAP(a, b) { const bool sw = t[b] < t[a]; const float ta = sw ? t[b] : t[a], tb = sw ? t[a] : t[b]; \
                         const int la = sw ? l[b] : l[a], lb = sw ? l[a] : l[b]; t[a] = ta; t[b] = tb; l[a] = la; l[b] = lb; }
                    RT_CSWAP(0, 1) RT_CSWAP(2, 3) RT_CSWAP(0, 2) RT_CSWAP(1, 3) RT_CSWAP(1, 2)
#undef RT_CSWAP
                    if (nh > 3) { stack[sp * stride] = l[3]; ++sp; }
                    if (nh > 2) { stack[sp * stride] = l[2]; ++sp; }
                    if (nh > 1) { stack[sp * stride] = l[1]; ++sp; }
                }
                cur = l[0];
                continue;
            }
        } else {
            const int first = (int)(((unsigned)cur & 0x7fffffffu) >> 2), count = (cur & 3) + 1;
            for (int k = 0; k < count; ++k) {
                const int slot = first + k;
                const float *tv = bvh.tris + 9 * (size_t)slot;
                float th;
                ++nt;
                if (ray_triangle(o, d, tnear, tfar, tv, tv + 3, tv + 6, &th)) {
                    const int s = bvh.ids[2 * slot], p = bvh.ids[2 * slot + 1];
                    if (ANY) { if (cnt) { cnt->nodes += nn; cnt->tris += nt; } return Hit{th, s, p}; }
                    if (closer(th, s, p, best)) best = Hit{th, s, p};
                }
            }
        }
        if (sp == 0) break;
        --sp;
        cur = stack[sp * stride];
    }
    if (cnt) { cnt->nodes += nn; cnt->tris += nt; }
    return best;
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
    std::vector<Node4> wide;          // collapse_wide(): the 4-wide records over the same leaves (triangle hierarchy only)
    int wide_stack_need = 0;
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
// (Re)derives h.wide / h.wide_stack_need from the binary records: every interior record adopts the children of its
// largest-area interior child until it has four (or only leaves are left).  Linear in the node count; called after a build
// and after a refit (the boxes are copies of the binary records' boxes).
void collapse_wide(BvhHost &h);
}
