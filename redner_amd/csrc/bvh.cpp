// bvh.cpp -- host-side binned-SAH build of the triangle hierarchy described in bvh.h.
// Runs once per Scene (the reference rebuilds its Embree/OptiX scene per Scene object too,
// src/scene.cpp:128-154); GPU-side build/refit is SURVEY.md section 8f row 1.
#include "bvh.h"
#include "hostpool.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <stdexcept>

namespace rt {
namespace {

struct Prim { float lo[3], hi[3], c[3]; int shape, prim; float v[9]; };

struct Box {
    float lo[3], hi[3];
    Box() { for (int k = 0; k < 3; ++k) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -lo[k]; } }
    void grow(const float *l, const float *h) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], l[k]); hi[k] = std::max(hi[k], h[k]); } }
    void grow_pt(const float *p) { grow(p, p); }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0) return 0;
        return dx * dy + dy * dz + dz * dx;
    }
};

// Builds the subtree over prims[first, first+count) of a shared primitive array into its own node / triangle
// arrays (root = node 0).  Disjoint ranges can be built by different threads; the splice below puts them
// together in a fixed order, so the layout does not depend on thread timing.
struct Builder {
    Prim *prims = nullptr;
    BvhHost out;
    static constexpr int kMaxBins = 64;
    int kLeafMax = 4;          // triangles per leaf (RDR_BVH_LEAF, 1..4)
    int kBins = 64;            // tuning knobs (RDR_BVH_BINS / RDR_BVH_TCOST); defaults from a sweep on bunny_box (profiles/r1_notes.md)
    float kTravCost = 0.3f;

    void set_bounds(int node, const Box &b) {
        Node &n = out.nodes[node];
        for (int k = 0; k < 3; ++k) { n.lo[k] = b.lo[k]; n.hi[k] = b.hi[k]; }
        pad_box(n.lo, n.hi);
    }

    void make_leaf(int node, int first, int count) {
        int slot0 = (int)out.ids.size() / 2;
        for (int i = first; i < first + count; ++i) {
            out.tris.insert(out.tris.end(), prims[i].v, prims[i].v + 9);
            out.ids.push_back(prims[i].shape);
            out.ids.push_back(prims[i].prim);
        }
        out.nodes[node].a = slot0;
        out.nodes[node].b = count;
    }

    // Bounds of the range and its SAH split (partitions the range in place).  False: the range becomes a leaf.
    bool split_range(int first, int count, Box &b, int &mid) {
        Box cb;
        for (int i = first; i < first + count; ++i) { b.grow(prims[i].lo, prims[i].hi); cb.grow_pt(prims[i].c); }
        int best_axis = -1, best_bin = -1;
        float best_cost = std::numeric_limits<float>::infinity();
        if (count > 1) {
            for (int axis = 0; axis < 3; ++axis) {
                float ext = cb.hi[axis] - cb.lo[axis];
                if (!(ext > 0)) continue;
                Box bins[kMaxBins]; int cnt[kMaxBins] = {0};
                float scale = kBins / ext;
                for (int i = first; i < first + count; ++i) {
                    int bi = std::min(kBins - 1, std::max(0, (int)((prims[i].c[axis] - cb.lo[axis]) * scale)));
                    bins[bi].grow(prims[i].lo, prims[i].hi); cnt[bi]++;
                }
                float ra[kMaxBins]; int rc[kMaxBins];
                Box acc; int c = 0;
                for (int k = kBins - 1; k > 0; --k) { acc.grow(bins[k].lo, bins[k].hi); c += cnt[k]; ra[k] = acc.half_area(); rc[k] = c; }
                Box l; int lc = 0;
                for (int k = 0; k < kBins - 1; ++k) {
                    l.grow(bins[k].lo, bins[k].hi); lc += cnt[k];
                    if (lc == 0 || rc[k + 1] == 0) continue;
                    float cost = l.half_area() * lc + ra[k + 1] * rc[k + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = k; }
                }
            }
        }
        float leaf_cost = b.half_area() * count;
        bool split = best_axis >= 0 && (count > kLeafMax || best_cost + kTravCost * b.half_area() < leaf_cost);
        mid = first + count / 2;
        if (split) {
            float ext = cb.hi[best_axis] - cb.lo[best_axis];
            float scale = kBins / ext;
            float lo = cb.lo[best_axis];
            // (stable: the order of a leaf's triangles is then a function of the input order alone, and equals the device
            //  builder's ballot-ranked partition, bvh_gpu.cpp)
            Prim *it = std::stable_partition(prims + first, prims + first + count, [&](const Prim &p) {
                int bi = std::min(kBins - 1, std::max(0, (int)((p.c[best_axis] - lo) * scale)));
                return bi <= best_bin;
            });
            mid = (int)(it - prims);
            if (mid == first || mid == first + count) split = false;
        }
        if (!split) {
            if (count <= kLeafMax) return false;
            // degenerate (coincident centroids): median split by index keeps the tree finite
            mid = first + count / 2;
        }
        return true;
    }

    void build(int node, int first, int count, int depth) {
        out.depth = std::max(out.depth, depth);
        Box b; int mid;
        bool inner = split_range(first, count, b, mid);
        set_bounds(node, b);
        if (!inner) { make_leaf(node, first, count); return; }
        int left = (int)out.nodes.size();
        out.nodes.push_back(Node{});
        out.nodes.push_back(Node{});
        out.nodes[node].a = left;
        out.nodes[node].b = 0;
        build(left, first, mid - first, depth + 1);
        build(left + 1, mid, first + count - mid, depth + 1);
    }

    // Top of the tree: the two halves of a large range are built concurrently and spliced as
    // [root, left root, right root, rest of left, rest of right].
    static constexpr int kTaskMin = 1024, kTaskDepth = 5;
    void build_top(int first, int count, int depth) {
        out.nodes.assign(1, Node{});
        if (count < kTaskMin || depth >= kTaskDepth) { build(0, first, count, depth); return; }
        out.depth = depth;
        Box b; int mid;
        bool inner = split_range(first, count, b, mid);
        set_bounds(0, b);
        if (!inner) { make_leaf(0, first, count); return; }
        Builder L, R;
        L.prims = R.prims = prims; L.kBins = R.kBins = kBins; L.kTravCost = R.kTravCost = kTravCost; L.kLeafMax = R.kLeafMax = kLeafMax;
        auto left_job = hostpool::run([&] { L.build_top(first, mid - first, depth + 1); });
        R.build_top(mid, first + count - mid, depth + 1);
        left_job.wait();
        const int nl = (int)L.out.nodes.size(), nr = (int)R.out.nodes.size();
        const int tl = (int)L.out.ids.size() / 2;
        out.nodes.resize(1 + nl + nr);
        out.nodes[0].a = 1; out.nodes[0].b = 0;
        auto place = [&](const Node &src, int child_shift, int slot_shift) {
            Node n = src;
            if (n.b > 0) n.a += slot_shift; else n.a += child_shift;
            return n;
        };
        // local index i >= 1 of the left subtree lands at i + 2, of the right subtree at i + 1 + nl
        out.nodes[1] = place(L.out.nodes[0], 2, 0);
        out.nodes[2] = place(R.out.nodes[0], 1 + nl, tl);
        for (int i = 1; i < nl; ++i) out.nodes[i + 2] = place(L.out.nodes[i], 2, 0);
        for (int i = 1; i < nr; ++i) out.nodes[i + 1 + nl] = place(R.out.nodes[i], 1 + nl, tl);
        out.tris = std::move(L.out.tris); out.tris.insert(out.tris.end(), R.out.tris.begin(), R.out.tris.end());
        out.ids = std::move(L.out.ids); out.ids.insert(out.ids.end(), R.out.ids.begin(), R.out.ids.end());
        out.depth = std::max(L.out.depth, R.out.depth);
    }
};

} // namespace

namespace { double inner_half_area(const BvhHost &h); }

// Hierarchy over axis-aligned boxes (6 floats each: lo.xyz, hi.xyz) with the same builder: leaves list box indices in
// `ids` ({0, box index} per slot), `tris` stays empty.  Node bounds are padded unions of the given boxes, so a query that
// is monotone in box inclusion can cull with the nodes and decide exactly at the leaves (the NEE-mode edge gather).
BvhHost build_box_bvh(const float *boxes, int n) {
    Builder bd;
    if (n <= 0) return bd.out;
    std::vector<Prim> prims((size_t)n);
    for (int i = 0; i < n; ++i) {
        Prim &p = prims[i];
        p.shape = 0; p.prim = i;
        for (int a = 0; a < 3; ++a) {
            p.lo[a] = boxes[6 * i + a]; p.hi[a] = boxes[6 * i + 3 + a];
            p.c[a] = 0.5f * (p.lo[a] + p.hi[a]);
        }
        for (int k = 0; k < 9; ++k) p.v[k] = 0.f;
    }
    bd.prims = prims.data();
    bd.build_top(0, n, 0);
    bd.out.tris.clear();
    bd.out.inner_area = inner_half_area(bd.out);
    return bd.out;
}

// Node records in breadth-first order (siblings stay adjacent, the root stays 0): the first K records are then the top
// levels of the hierarchy, which the traversal kernels stage into LDS.  Leaf slots (tris / ids) are untouched.
static void reorder_breadth_first(BvhHost &h) {
    const int n = (int)h.nodes.size();
    if (n <= 1) return;
    std::vector<Node> out((size_t)n);
    std::vector<int> queue;           // old indices of inner nodes, in the order their children get their new places
    queue.reserve((size_t)n / 2);
    out[0] = h.nodes[0];
    int next = 1;
    std::vector<int> new_of_queue;    // new index of each queued node
    if (h.nodes[0].b == 0) { queue.push_back(0); new_of_queue.push_back(0); }
    for (size_t q = 0; q < queue.size(); ++q) {
        const Node &old = h.nodes[queue[q]];
        const int l = old.a, r = old.a + 1;
        out[new_of_queue[q]].a = next;
        out[next] = h.nodes[l]; out[next + 1] = h.nodes[r];
        if (h.nodes[l].b == 0) { queue.push_back(l); new_of_queue.push_back(next); }
        if (h.nodes[r].b == 0) { queue.push_back(r); new_of_queue.push_back(next + 1); }
        next += 2;
    }
    h.nodes.swap(out);
}

namespace {
double inner_half_area(const BvhHost &h) {
    double a = 0;
    for (const Node &n : h.nodes) {
        if (n.b > 0) continue;
        const double dx = (double)n.hi[0] - n.lo[0], dy = (double)n.hi[1] - n.lo[1], dz = (double)n.hi[2] - n.lo[2];
        a += dx * dy + dy * dz + dz * dx;
    }
    return a;
}
// leaf_box(slot, lo, hi): box of the primitive in leaf slot `slot`
template <class LeafBox> double refit_sweep(BvhHost &h, LeafBox leaf_box) {
    for (int i = (int)h.nodes.size() - 1; i >= 0; --i) {
        Node &n = h.nodes[i];
        Box b;
        if (n.b > 0) {
            for (int k = 0; k < n.b; ++k) { float lo[3], hi[3]; leaf_box(n.a + k, lo, hi); b.grow(lo, hi); }
            for (int k = 0; k < 3; ++k) { n.lo[k] = b.lo[k]; n.hi[k] = b.hi[k]; }
            pad_box(n.lo, n.hi);
        } else {                       // children are padded already; their union covers everything below
            const Node &l = h.nodes[n.a], &r = h.nodes[n.a + 1];
            for (int k = 0; k < 3; ++k) { n.lo[k] = std::min(l.lo[k], r.lo[k]); n.hi[k] = std::max(l.hi[k], r.hi[k]); }
        }
    }
    const double now = inner_half_area(h);
    return h.inner_area > 0 ? now / h.inner_area : 1.0;
}
}

static double refit_bvh_binary(BvhHost &h, const std::vector<MeshView> &meshes);
double refit_bvh(BvhHost &h, const std::vector<MeshView> &meshes) {
    const double r = refit_bvh_binary(h, meshes);
    collapse_wide(h);
    return r;
}
static double refit_bvh_binary(BvhHost &h, const std::vector<MeshView> &meshes) {
    const size_t slots = h.ids.size() / 2;
    for (size_t sl = 0; sl < slots; ++sl) {
        const MeshView &m = meshes[(size_t)h.ids[2 * sl]];
        const int t = h.ids[2 * sl + 1];
        for (int k = 0; k < 3; ++k) {
            const int vi = m.indices[3 * t + k];
            for (int a = 0; a < 3; ++a) h.tris[9 * sl + 3 * k + a] = m.vertices[3 * vi + a];
        }
    }
    return refit_sweep(h, [&](int slot, float lo[3], float hi[3]) {
        const float *v = h.tris.data() + 9 * (size_t)slot;
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(v[a], std::min(v[3 + a], v[6 + a]));
            hi[a] = std::max(v[a], std::max(v[3 + a], v[6 + a]));
        }
    });
}

double refit_box_bvh(BvhHost &h, const float *boxes) {
    return refit_sweep(h, [&](int slot, float lo[3], float hi[3]) {
        const float *b = boxes + 6 * (size_t)h.ids[2 * (size_t)slot + 1];
        for (int a = 0; a < 3; ++a) { lo[a] = b[a]; hi[a] = b[3 + a]; }
    });
}

void collapse_wide(BvhHost &h) {
    h.wide.clear();
    h.wide_stack_need = 0;
    if (h.nodes.empty()) return;
    auto area = [&](int i) {
        const Node &n = h.nodes[i];
        const float dx = n.hi[0] - n.lo[0], dy = n.hi[1] - n.lo[1], dz = n.hi[2] - n.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    struct Item { int binary; int wide; int pending; };      // binary interior record -> its wide record; pending: stack entries above it
    std::vector<Item> queue;                                // breadth first: the first records are the top of the hierarchy
    h.wide.push_back(Node4{});
    queue.push_back(Item{0, 0, 0});
    for (size_t q = 0; q < queue.size(); ++q) {
        const Item it = queue[q];
        int kids[4], nk = 0;
        if (h.nodes[it.binary].b > 0) kids[nk++] = it.binary;                 // a hierarchy that is one leaf
        else { kids[nk++] = h.nodes[it.binary].a; kids[nk++] = h.nodes[it.binary].a + 1; }
        while (nk < 4) {
            int pick = -1; float best = -1.f;
            for (int k = 0; k < nk; ++k) if (h.nodes[kids[k]].b == 0 && area(kids[k]) > best) { best = area(kids[k]); pick = k; }
            if (pick < 0) break;
            const int a = h.nodes[kids[pick]].a;
            kids[pick] = a; kids[nk++] = a + 1;
        }
        Node4 w;
        for (int k = 0; k < 4; ++k) {
            w.lox[k] = w.loy[k] = w.loz[k] = std::numeric_limits<float>::infinity();
            w.hix[k] = w.hiy[k] = w.hiz[k] = -std::numeric_limits<float>::infinity();
            w.link[k] = kEmptyLink; w.aux[k] = 0;
        }
        w.aux[0] = nk;
        // a walk that enters this record pushes at most nk - 1 entries and descends into one child
        const int below = it.pending + nk - 1;
        h.wide_stack_need = std::max(h.wide_stack_need, below + 1);
        for (int k = 0; k < nk; ++k) {
            const Node &c = h.nodes[kids[k]];
            w.lox[k] = c.lo[0]; w.loy[k] = c.lo[1]; w.loz[k] = c.lo[2];
            w.hix[k] = c.hi[0]; w.hiy[k] = c.hi[1]; w.hiz[k] = c.hi[2];
            if (c.b > 0) w.link[k] = leaf_link(c.a, c.b);
            else {
                w.link[k] = (int)h.wide.size();
                h.wide.push_back(Node4{});
                queue.push_back(Item{kids[k], w.link[k], below});
            }
        }
        h.wide[it.wide] = w;
    }
}

BvhHost build_bvh(const std::vector<MeshView> &meshes) {
    Builder bd;
    std::vector<Prim> prims;
    if (const char *e = std::getenv("RDR_BVH_BINS")) bd.kBins = std::min(64, std::max(2, std::atoi(e)));
    if (const char *e = std::getenv("RDR_BVH_TCOST")) bd.kTravCost = (float)std::atof(e);
    if (const char *e = std::getenv("RDR_BVH_LEAF")) bd.kLeafMax = std::min(4, std::max(1, std::atoi(e)));
    for (size_t s = 0; s < meshes.size(); ++s) {
        const MeshView &m = meshes[s];
        for (int t = 0; t < m.num_triangles; ++t) {
            Prim p; p.shape = (int)s; p.prim = t;
            for (int k = 0; k < 3; ++k) {
                int vi = m.indices[3 * t + k];
                for (int a = 0; a < 3; ++a) p.v[3 * k + a] = m.vertices[3 * vi + a];
            }
            for (int a = 0; a < 3; ++a) {
                p.lo[a] = std::min(p.v[a], std::min(p.v[3 + a], p.v[6 + a]));
                p.hi[a] = std::max(p.v[a], std::max(p.v[3 + a], p.v[6 + a]));
                p.c[a] = 0.5f * (p.lo[a] + p.hi[a]);
            }
            prims.push_back(p);
        }
    }
    if (prims.empty()) return bd.out;
    bd.prims = prims.data();
    bd.build_top(0, (int)prims.size(), 0);
    if (bd.out.depth + 2 > kTraverseStack) throw std::runtime_error("triangle hierarchy deeper than the traversal stack");
    reorder_breadth_first(bd.out);
    bd.out.inner_area = inner_half_area(bd.out);
    collapse_wide(bd.out);
    return bd.out;
}

} // namespace rt
