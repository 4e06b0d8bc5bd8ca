// edges.cpp -- host-side construction of the edge list, the primary-edge PMF/CDF and the two edge
// hierarchies (see edges.h for the behavioural spec and why the build is order-exact).
#include "edges.h"
#include "hostpool.h"
#include "camera.h"
#include "scene.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>

namespace rdr {

namespace {

// ---- the sequential merge sort the reference's CPU build runs (Thrust's sequential backend:
// top-down split at n/2, insertion sort below 33 elements, merge taking from the right run only
// when comp(right, left)).  Restated because two of the reference's comparators are not strict
// weak orders (they return true on equality), which makes the outcome depend on the algorithm.
template <class T, class Cmp>
void insertion_sort_seq(T *first, T *last, Cmp comp) {
    if (first == last) return;
    for (T *i = first + 1; i != last; ++i) {
        T tmp = *i;
        if (comp(tmp, *first)) {
            for (T *j = i; j != first; --j) *j = *(j - 1);
            *first = tmp;
        } else {
            T *j = i, *k = i - 1;
            while (comp(tmp, *k)) { *j = *k; j = k; --k; }
            *j = tmp;
        }
    }
}
// `scratch` holds at least (last - first) / 2 + 1 elements.  The halves of the top `par_levels` levels are sorted
// on two threads: the algorithm is deterministic and the halves are disjoint, so the outcome is the same.
template <class T, class Cmp>
void merge_sort_rec(T *first, T *last, T *scratch, Cmp comp, int par_levels) {
    ptrdiff_t n = last - first;
    if (n <= 32) { insertion_sort_seq(first, last, comp); return; }
    T *mid = first + n / 2;
    if (par_levels > 0 && n >= 4096) {
        std::vector<T> other((size_t)(n / 2 / 2 + 2));
        auto job = hostpool::run([&] { merge_sort_rec(first, mid, other.data(), comp, par_levels - 1); });
        merge_sort_rec(mid, last, scratch, comp, par_levels - 1);
        job.wait();
    } else {
        merge_sort_rec(first, mid, scratch, comp, 0);
        merge_sort_rec(mid, last, scratch, comp, 0);
    }
    const ptrdiff_t na = mid - first;
    for (ptrdiff_t k = 0; k < na; ++k) scratch[k] = first[k];
    const T *a = scratch, *a_end = scratch + na;
    const T *b = mid;
    T *out = first;
    while (a < a_end && b < last) {
        if (comp(*b, *a)) *out++ = *b++; else *out++ = *a++;
    }
    while (a < a_end) *out++ = *a++;
    // what is left of the right run is already in place
}
template <class T, class Cmp>
void merge_sort_seq(T *first, T *last, Cmp comp) {
    std::vector<T> scratch((size_t)((last - first) / 2 + 2));
    merge_sort_rec(first, last, scratch.data(), comp, 3);
}

bool f3_less_eq(F3 a, F3 b) {      // "less_than" of src/edge.cpp:93-102: true on equality
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    if (a.z != b.z) return a.z < b.z;
    return true;
}

struct SortedEnds { F3 lo, hi; };
SortedEnds sorted_ends(const ShapeD &sh, const EdgeD &e) {
    F3 a = f3_vertex(sh, e.v0), b = f3_vertex(sh, e.v1);
    if (f3_less_eq(b, a)) std::swap(a, b);
    return SortedEnds{a, b};
}

// Cohen-Sutherland clip of a screen-space segment against [0,1]^2 (src/line_clip.h:20-102).
int out_code(V2 v) {
    int c = 0;
    if (v.x < 0.f) c |= 1; else if (v.x > 1.f) c |= 2;
    if (v.y < 0.f) c |= 4; else if (v.y > 1.f) c |= 8;
    return c;
}
bool clip_unit_square(V2 a, V2 b, V2 &ac, V2 &bc) {
    int ca = out_code(a), cb = out_code(b);
    ac = a; bc = b;
    for (;;) {
        if (!(ca | cb)) return true;
        if (ca & cb) return false;
        V2 v = v2(0, 0);
        int co = ca ? ca : cb;
        if (co & 8) { v.x = ac.x + (bc.x - ac.x) * (1.f - ac.y) / (bc.y - ac.y); v.y = 1.f; }
        else if (co & 4) { v.x = ac.x + (bc.x - ac.x) * (0.f - ac.y) / (bc.y - ac.y); v.y = 0.f; }
        else if (co & 2) { v.y = ac.y + (bc.y - ac.y) * (1.f - ac.x) / (bc.x - ac.x); v.x = 1.f; }
        else if (co & 1) { v.y = ac.y + (bc.y - ac.y) * (0.f - ac.x) / (bc.x - ac.x); v.x = 0.f; }
        if (co == ca) { ac = v; ca = out_code(ac); } else { bc = v; cb = out_code(bc); }
    }
}

// ---- 6-D bounds of one edge (src/edge_tree.cpp:25-74) ----
struct Box6 { V3 p_min, p_max, d_min, d_max; };
Box6 empty_box6() {
    double inf = std::numeric_limits<double>::infinity();
    return Box6{v3(inf), v3(-inf), v3(inf), v3(-inf)};
}
V3 vmin(V3 a, V3 b) { return V3{std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
V3 vmax(V3 a, V3 b) { return V3{std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }
Box6 merge6(const Box6 &a, const Box6 &b) {
    return Box6{vmin(a.p_min, b.p_min), vmax(a.p_max, b.p_max), vmin(a.d_min, b.d_min), vmax(a.d_max, b.d_max)};
}

uint64_t expand3(uint64_t x) {          // two zero bits after each of 21 bits
    x &= 0x1fffff;
    x = (x | x << 32) & 0x1f00000000ffff;
    x = (x | x << 16) & 0x1f0000ff0000ff;
    x = (x | x << 8) & 0x100f00f00f00f00f;
    x = (x | x << 4) & 0x10c30c30c30c30c3;
    x = (x | x << 2) & 0x1249249249249249;
    return x;
}
uint64_t expand6(uint64_t x) {          // five zero bits after each of 10 bits
    uint64_t r = 0;
    for (int i = 0; i < 10; ++i) r |= ((x >> i) & 1ull) << (6 * i);
    return r;
}
double unit_coord(double v, double lo, double hi) {
    if (hi - lo <= 0.f) return 0.5f;
    return (v - lo) / (hi - lo);
}

int clz64(uint64_t x) { return x == 0 ? 64 : __builtin_clzll(x); }

// f(begin, end) over [0, n) in up to 16 contiguous chunks on separate threads (iterations must be independent).
using hostpool::parallel_chunks;

// ---- one hierarchy ------------------------------------------------------------------------------
struct TreeBuilder {
    bool is3d;
    const ShapeD *shapes;
    const std::vector<EdgeD> &edges;
    const std::vector<Box6> &bounds;
    std::vector<int> ids;               // edge ids of this tree, Morton-sorted
    std::vector<uint64_t> codes;
    std::vector<EdgeNode> nodes;        // [internal | leaves]
    std::vector<int> leaves_below;      // per node: leaves in its subtree (kept current by the treelet pass)
    int n = 0, n_internal = 0;

    TreeBuilder(bool is3d_, const ShapeD *shapes_, const std::vector<EdgeD> &edges_, const std::vector<Box6> &bounds_)
        : is3d(is3d_), shapes(shapes_), edges(edges_), bounds(bounds_) {}

    int leaf_ref(int i) const { return n_internal + i; }

    double area(const EdgeNode &nd) const { return edge_node_area(nd, is3d); }
    void merge_children(EdgeNode &dst, const EdgeNode &a, const EdgeNode &b) const { edge_node_merge(dst, a, b, is3d); }

    int lcp(int i, int j) const {
        if (i < 0 || i >= n || j < 0 || j >= n) return -1;
        uint64_t a = codes[i], b = codes[j];
        if (a == b) return clz64(a ^ b) + clz64((uint64_t)ids[i] ^ (uint64_t)ids[j]);
        return clz64(a ^ b);
    }

    void build(const std::vector<int> &edge_ids) {
        PhaseTimer timer(is3d ? "3-D tree" : "6-D tree");
        ids = edge_ids;
        n = (int)ids.size();
        if (n == 0) return;
        // scene bounds of this subset
        Box6 sb = empty_box6();
        for (int id : ids) sb = merge6(sb, bounds[id]);
        codes.resize(n);
        parallel_chunks(n, 4096, [&](int i_begin, int i_end) {
        for (int i = i_begin; i < i_end; ++i) {
            const Box6 &b = bounds[ids[i]];
            V3 pc = 0.5f * (b.p_min + b.p_max);
            if (is3d) {
                double s = (1 << 21) - 1;
                uint64_t x = (uint64_t)(unit_coord(pc.x, sb.p_min.x, sb.p_max.x) * s);
                uint64_t y = (uint64_t)(unit_coord(pc.y, sb.p_min.y, sb.p_max.y) * s);
                uint64_t z = (uint64_t)(unit_coord(pc.z, sb.p_min.z, sb.p_max.z) * s);
                codes[i] = (expand3(x) << 2u) | (expand3(y) << 1u) | expand3(z);
            } else {
                V3 dc = 0.5f * (b.d_min + b.d_max);
                uint64_t px = (uint64_t)(unit_coord(pc.x, sb.p_min.x, sb.p_max.x) * 1023);
                uint64_t py = (uint64_t)(unit_coord(pc.y, sb.p_min.y, sb.p_max.y) * 1023);
                uint64_t pz = (uint64_t)(unit_coord(pc.z, sb.p_min.z, sb.p_max.z) * 1023);
                uint64_t dx = (uint64_t)(unit_coord(dc.x, sb.d_min.x, sb.d_max.x) * 1023);
                uint64_t dy = (uint64_t)(unit_coord(dc.y, sb.d_min.y, sb.d_max.y) * 1023);
                uint64_t dz = (uint64_t)(unit_coord(dc.z, sb.d_min.z, sb.d_max.z) * 1023);
                codes[i] = (expand6(px) << 5u) | (expand6(py) << 4u) | (expand6(pz) << 3u) |
                           (expand6(dx) << 2u) | (expand6(dy) << 1u) | expand6(dz);
            }
        }
        });
        timer.lap("codes");
        // stable sort of (code, id) by code.  (code, position before the sort) is a unique key, so any comparison sort of
        // those records gives the stable order: four runs sorted on pool threads, then merged
        {
            struct Rec { uint64_t code; int pos, id; };
            std::vector<Rec> recs(n);
            for (int i = 0; i < n; ++i) recs[i] = Rec{codes[i], i, ids[i]};
            auto before = [](const Rec &a, const Rec &b) { return a.code != b.code ? a.code < b.code : a.pos < b.pos; };
            if (n >= 8192) {
                const int q1 = n / 4, q2 = n / 2, q3 = q2 + n / 4;
                {
                    auto j1 = hostpool::run([&] { std::sort(recs.begin(), recs.begin() + q1, before); });
                    auto j2 = hostpool::run([&] { std::sort(recs.begin() + q1, recs.begin() + q2, before); });
                    auto j3 = hostpool::run([&] { std::sort(recs.begin() + q2, recs.begin() + q3, before); });
                    std::sort(recs.begin() + q3, recs.end(), before);
                    j1.wait(); j2.wait(); j3.wait();
                }
                {
                    auto j1 = hostpool::run([&] { std::inplace_merge(recs.begin(), recs.begin() + q1, recs.begin() + q2, before); });
                    std::inplace_merge(recs.begin() + q2, recs.begin() + q3, recs.end(), before);
                    j1.wait();
                }
                std::inplace_merge(recs.begin(), recs.begin() + q2, recs.end(), before);
            } else {
                std::sort(recs.begin(), recs.end(), before);
            }
            for (int i = 0; i < n; ++i) { ids[i] = recs[i].id; codes[i] = recs[i].code; }
        }

        timer.lap("sort");
        n_internal = std::max(n - 1, 1);
        EdgeNode init;
        double inf = std::numeric_limits<double>::infinity();
        init.p_min = init.d_min = v3(inf); init.p_max = init.d_max = v3(-inf);
        init.wlen = 0; init.cost = 0; init.parent = -1; init.child0 = init.child1 = -1; init.edge_id = -1;
        nodes.assign(n_internal + n, init);

        // Karras radix tree over the sorted codes (ties broken by edge id); every internal node is independent
        parallel_chunks(n - 1, 2048, [&](int idx_begin, int idx_end) {
        for (int idx = idx_begin; idx < idx_end; ++idx) {
            int d = (lcp(idx, idx + 1) - lcp(idx, idx - 1) >= 0) ? 1 : -1;
            int dmin = lcp(idx, idx - d);
            int lmax = 2;
            while (lcp(idx, idx + lmax * d) > dmin) lmax *= 2;
            int l = 0, divider = 2;
            for (int t = lmax / divider; t >= 1;) {
                if (lcp(idx, idx + (l + t) * d) > dmin) l += t;
                if (t == 1) break;
                divider *= 2;
                t = lmax / divider;
            }
            int j = idx + l * d;
            int dnode = lcp(idx, j);
            int s = 0;
            divider = 2;
            for (int t = (l + (divider - 1)) / divider; t >= 1;) {
                if (lcp(idx, idx + (s + t) * d) > dnode) s += t;
                if (t == 1) break;
                divider *= 2;
                t = (l + (divider - 1)) / divider;
            }
            int gamma = idx + s * d + std::min(d, 0);
            EdgeNode &nd = nodes[idx];
            if (std::min(idx, j) == gamma) { nd.child0 = leaf_ref(gamma); nodes[leaf_ref(gamma)].parent = idx; }
            else { nd.child0 = gamma; nodes[gamma].parent = idx; }
            if (std::max(idx, j) == gamma + 1) { nd.child1 = leaf_ref(gamma + 1); nodes[leaf_ref(gamma + 1)].parent = idx; }
            else { nd.child1 = gamma + 1; nodes[gamma + 1].parent = idx; }
        }
        });

        timer.lap("radix tree");
        // leaves + bottom-up bounds / weighted length
        std::vector<int> counter(n_internal + n, 0);
        leaves_below.assign(n_internal + n, 1);
        parallel_chunks(n, 2048, [&](int begin, int end) {
            for (int i = begin; i < end; ++i) {
                EdgeNode &lf = nodes[leaf_ref(i)];
                const Box6 &b = bounds[ids[i]];
                lf.p_min = b.p_min; lf.p_max = b.p_max;
                if (!is3d) { lf.d_min = b.d_min; lf.d_max = b.d_max; }
                const EdgeD &e = edges[ids[i]];
                lf.wlen = f3_distance(edge_v0f(shapes, e), edge_v1f(shapes, e)) * edge_exterior_dihedral(shapes, e);
                lf.edge_id = ids[i];
            }
        });
        // (arrival counters: whoever reaches a node second finds both children complete and carries on upwards)
        parallel_chunks(n, 1024, [&](int begin, int end) {
        for (int i = begin; i < end; ++i) {
            EdgeNode &lf = nodes[leaf_ref(i)];
            int cur = lf.parent;
            while (cur >= 0) {
                if (__atomic_fetch_add(&counter[cur], 1, __ATOMIC_ACQ_REL) == 0) break;       // first arrival waits for the sibling
                EdgeNode &nd = nodes[cur];
                merge_children(nd, nodes[nd.child0], nodes[nd.child1]);
                nd.wlen = nodes[nd.child0].wlen + nodes[nd.child1].wlen;
                leaves_below[cur] = leaves_below[nd.child0] + leaves_below[nd.child1];
                cur = nd.parent;
            }
        }
        });
        if (n == 1) { nodes[0] = nodes[leaf_ref(0)]; }     // single primitive: the root is a copy of the leaf

        // treelet optimisation, bottom-up: a node is restructured once both child subtrees are done
        // (src/edge_tree.cpp:724-790 climbs from the leaves with arrival counters).  A restructuring only touches
        // nodes below its root, so any schedule that respects that dependency gives the same tree; the subtrees
        // below the top levels are independent jobs.
        timer.lap("bounds");
        for (int i = 0; i < n; ++i) nodes[leaf_ref(i)].cost = area(nodes[leaf_ref(i)]);
        if (n > 1) optimize_subtree(0, 0);
        timer.lap("treelets");
    }

    void optimize_subtree(int node, int depth) {
        if (nodes[node].edge_id != -1) return;
        const int c0 = nodes[node].child0, c1 = nodes[node].child1;
        if (depth < 24 && leaves_below[node] >= 512) {          // (a node's leaf set does not change when its subtree is restructured)
            auto job = hostpool::run([&] { optimize_subtree(c0, depth + 1); });
            optimize_subtree(c1, depth + 1);
            job.wait();
        } else {
            optimize_sequential(c0);
            optimize_sequential(c1);
        }
        treelet_optimize(node);
    }
    void optimize_sequential(int top) {          // post-order walk without recursion (radix trees can be deep)
        if (nodes[top].edge_id != -1) return;
        std::vector<std::pair<int, int>> stack;   // (node, children pushed?)
        stack.emplace_back(top, 0);
        while (!stack.empty()) {
            auto &e = stack.back();
            const int node = e.first;
            if (e.second == 0) {
                e.second = 1;
                const int c0 = nodes[node].child0, c1 = nodes[node].child1;
                if (nodes[c1].edge_id == -1) stack.emplace_back(c1, 0);
                if (nodes[c0].edge_id == -1) stack.emplace_back(c0, 0);
            } else {
                stack.pop_back();
                treelet_optimize(node);
            }
        }
    }

    // ---- treelet restructuring with <= 7 leaves (behaviour: src/edge_tree.cpp:464-711; Karras & Aila 2013, Algorithm 2) ----
    // The result must equal the reference's tree link for link (tests/test_edge_build.py): which inner node lands where is
    // fixed by (i) the order the treelet's leaves are collected in, (ii) the first minimal partition in the reference's
    // enumeration order and (iii) the order the freed inner nodes are handed out.  Inside those constraints the formulation is
    // ours: subset bounds are built incrementally (one merge per subset instead of up to six; unions are exact, so the areas
    // are the same doubles), the dynamic programme runs over subsets in increasing numeric order (every proper subset of s is
    // smaller than s, hence final -- no popcount sweeps), and the treelet is rebuilt by a recursion that refreshes bounds,
    // weights and costs in post-order on its way out (the reference climbs from every leaf re-checking its ancestors).
    void refresh(int node) {
        EdgeNode &nd = nodes[node];
        merge_children(nd, nodes[nd.child0], nodes[nd.child1]);
        nd.wlen = nodes[nd.child0].wlen + nodes[nd.child1].wlen;
        nd.cost = area(nd) + nodes[nd.child0].cost + nodes[nd.child1].cost;
        leaves_below[node] = leaves_below[nd.child0] + leaves_below[nd.child1];
    }

    struct Treelet {
        int leaves[7], inner[5];
        int nl = 0, ni = 0, next_inner = 0;
        uint8_t best_left[128];
    };

    // Subtree for the leaf subset `part`, hung under `parent` as child `side`.  Inner nodes are taken from t.inner in the
    // reference's hand-out order: a node first, then everything under its RIGHT half, then its left half.
    void hang(Treelet &t, uint8_t part, int side, int parent) {
        int node;
        if ((part & (part - 1)) == 0) {
            node = t.leaves[__builtin_ctz(part)];
        } else {
            node = t.inner[t.next_inner++];
            const uint8_t left = t.best_left[part], right = (uint8_t)(part & ~left);
            hang(t, right, 1, node);
            hang(t, left, 0, node);
            refresh(node);
        }
        if (side == 0) nodes[parent].child0 = node; else nodes[parent].child1 = node;
        nodes[node].parent = parent;
    }

    void treelet_optimize(int root) {
        if (nodes[root].edge_id != -1) return;
        Treelet t;
        // grow the treelet: repeatedly open the inner leaf with the largest box; it is replaced by the last leaf, its children
        // go to the end (this fixes the leaf numbering the partitions are expressed in)
        t.leaves[t.nl++] = nodes[root].child0;
        t.leaves[t.nl++] = nodes[root].child1;
        while (t.nl < 7) {
            int widest = -1;
            double widest_area = -1;
            for (int i = 0; i < t.nl; ++i) {
                if (nodes[t.leaves[i]].edge_id != -1) continue;
                const double ar = area(nodes[t.leaves[i]]);
                if (ar > widest_area) { widest_area = ar; widest = i; }
            }
            if (widest < 0) break;
            const int opened = t.leaves[widest];
            t.inner[t.ni++] = opened;
            t.leaves[widest] = t.leaves[t.nl - 1];
            t.leaves[t.nl - 1] = nodes[opened].child0;
            t.leaves[t.nl++] = nodes[opened].child1;
        }
        const int full = (1 << t.nl) - 1;
        // Surface area of every leaf subset.  [quirk] the reference starts every subset's union from leaf 0's box whether or
        // not leaf 0 is in the subset (src/edge_tree.cpp:560-568), so the area it prices subset s with is that of s | 1: those
        // are built incrementally (one merge each, from the subset without its lowest leaf other than leaf 0) and shared.
        EdgeNode box[128];
        double sa[128], best_cost[128];
        box[1] = nodes[t.leaves[0]];
        sa[1] = area(box[1]);
        for (int s = 3; s <= full; s += 2) {
            const int low = __builtin_ctz(s & ~1), rest = s & ~(1 << low);
            box[s] = box[rest];
            merge_children(box[s], box[rest], nodes[t.leaves[low]]);
            sa[s] = area(box[s]);
        }
        for (int s = 2; s <= full; s += 2) sa[s] = sa[s | 1];
        for (int i = 0; i < t.nl; ++i) best_cost[1 << i] = nodes[t.leaves[i]].cost;
        // cheapest split of every subset with >= 2 leaves; ties go to the first split in the order p = (p - d) & s
        for (int s = 3; s <= full; ++s) {
            if ((s & (s - 1)) == 0) continue;
            double cheapest = std::numeric_limits<double>::infinity();
            uint32_t arg = 0;
            const uint32_t d = ((uint32_t)s - 1u) & (uint32_t)s;
            uint32_t p = (0u - d) & (uint32_t)s;
            do {
                const double c = best_cost[p] + best_cost[(uint32_t)s ^ p];
                if (c < cheapest) { cheapest = c; arg = p; }
                p = (p - d) & (uint32_t)s;
            } while (p != 0);
            best_cost[s] = sa[s] + cheapest;
            t.best_left[s] = (uint8_t)arg;
        }
        const uint8_t left = t.best_left[full], right = (uint8_t)(full & ~left);
        // the reference rebuilds the left half completely before the right half (the other way round below the root)
        hang(t, left, 0, root);
        hang(t, right, 1, root);
        refresh(root);
    }
};

} // namespace

void delete_edge_data(EdgeData *e) {
    if (!e) return;
    for (void *p : e->owned) exec::pool_free(p);
    delete e;
}

EdgeData *compute_edge_data(const Scene &scene) {
    // one build at a time PER DEVICE: the topology caches below are per device (like scene.cpp's edge / topology caches), and
    // builds of different Scenes may be in flight together now that they run beside the caller (scene.cpp: create_scene)
    const int dev_slot = (scene.gpu_index < 0 ? 0 : scene.gpu_index) & 15;
    static std::mutex *build_locks = new std::mutex[16];
    std::lock_guard<std::mutex> build_guard(build_locks[dev_slot]);
    PhaseTimer timer("edge build");
    std::unique_ptr<EdgeData> ed(new EdgeData());
    const int ns = (int)scene.shapes.size();
    // host views of the shapes (same records, pointers into the host mirrors)
    std::vector<ShapeD> hs(scene.shapes);
    for (int i = 0; i < ns; ++i) {
        hs[i].geom = nullptr;          // host view: read the mesh arrays
        hs[i].vertices = scene.h_vertices[i].data();
        hs[i].indices = scene.h_indices[i].data();
        hs[i].normals = scene.shapes[i].normals ? scene.h_normals[i].data() : nullptr;
    }
    const ShapeD *shapes = hs.data();

    // ---- edge list (src/edge.cpp:233-297) ----
    // The first half of the construction -- half-edges sorted by vertex ids, runs merged into edges -- depends on the index
    // buffer only; a shape whose index buffer equals the one seen last at its position reuses it (an optimisation loop moves
    // vertices, not connectivity).  Everything after it depends on positions and is redone.
    struct MergedCache { std::vector<std::vector<int>> indices; std::vector<std::vector<EdgeD>> merged; };
    static MergedCache *merged_caches = new MergedCache[16];         // one build at a time per device (build_locks above; scene.cpp: EdgeBuilder)
    MergedCache *merged_cache = merged_caches + dev_slot;
    const bool cache_allowed = !(scene.build_flags & RDR_BUILD_NO_REFIT);
    if ((int)merged_cache->indices.size() != ns) { merged_cache->indices.assign(ns, {}); merged_cache->merged.assign(ns, {}); }
    std::vector<EdgeD> &edges = ed->edges;
    // The CANONICAL edges: every shape's merged edges in id order -- a function of the index buffers alone.  The final list is a
    // position-dependent permutation of a position-dependent subset of them (coplanar faces drop their edge); `can_of[i]` =
    // canonical index of final edge i.  The billboard hierarchy of the gather is kept over the canonical edges (below).
    std::vector<EdgeD> canon;
    std::vector<int> can_of;
    for (int sid = 0; sid < ns; ++sid) {
        const ShapeD &sh = hs[sid];
        std::vector<EdgeD> merged;
        if (cache_allowed && sh.num_triangles >= 256 && merged_cache->indices[sid] == scene.h_indices[sid]) {
            merged = merged_cache->merged[sid];
        } else {
        std::vector<EdgeD> he(3 * (size_t)sh.num_triangles);
        for (int t = 0; t < sh.num_triangles; ++t) {
            int i0 = sh.indices[3 * t], i1 = sh.indices[3 * t + 1], i2 = sh.indices[3 * t + 2];
            he[3 * t + 0] = EdgeD{sid, std::min(i0, i1), std::max(i0, i1), t, -1};
            he[3 * t + 1] = EdgeD{sid, std::min(i1, i2), std::max(i1, i2), t, -1};
            he[3 * t + 2] = EdgeD{sid, std::min(i2, i0), std::max(i2, i0), t, -1};
        }
        merge_sort_seq(he.data(), he.data() + he.size(), [](const EdgeD &a, const EdgeD &b) {
            if (a.v0 == b.v0) return a.v1 < b.v1;
            return a.v0 < b.v0;
        });
        // merge runs of identical (v0, v1): f1 of the run = f0 of its last member
        for (size_t i = 0; i < he.size();) {
            EdgeD cur = he[i];
            size_t j = i + 1;
            while (j < he.size() && he[j].v0 == cur.v0 && he[j].v1 == cur.v1) { cur.f1 = he[j].f0; ++j; }
            merged.push_back(cur);
            i = j;
        }
        if (cache_allowed && sh.num_triangles >= 256) { merged_cache->indices[sid] = scene.h_indices[sid]; merged_cache->merged[sid] = merged; }
        }
        // sort by endpoint positions so duplicated (e.g. UV-seam) edges become neighbours
        const int can_base = (int)canon.size();
        canon.insert(canon.end(), merged.begin(), merged.end());
        std::vector<int> can_local(merged.size());
        {
            struct Keyed { SortedEnds ends; EdgeD e; int can; };
            std::vector<Keyed> keyed(merged.size());
            for (size_t i = 0; i < merged.size(); ++i) keyed[i] = Keyed{sorted_ends(sh, merged[i]), merged[i], can_base + (int)i};
            merge_sort_seq(keyed.data(), keyed.data() + keyed.size(), [](const Keyed &a, const Keyed &b) {
                if (f3_ne(a.ends.lo, b.ends.lo)) return f3_less_eq(a.ends.lo, b.ends.lo);
                if (f3_ne(a.ends.hi, b.ends.hi)) return f3_less_eq(a.ends.hi, b.ends.hi);
                return true;
            });
            for (size_t i = 0; i < merged.size(); ++i) { merged[i] = keyed[i].e; can_local[i] = keyed[i].can; }
        }
        const int ne = (int)merged.size();
        std::vector<int> f1(ne);
        for (int i = 0; i < ne; ++i) {
            f1[i] = merged[i].f1;
            if (f1[i] != -1) continue;
            SortedEnds me = sorted_ends(sh, merged[i]);
            if (i > 0) {
                SortedEnds o = sorted_ends(sh, merged[i - 1]);
                if (f3_eq(me.lo, o.lo) && f3_eq(me.hi, o.hi)) f1[i] = merged[i - 1].f0;
            }
            if (i < ne - 1) {
                SortedEnds o = sorted_ends(sh, merged[i + 1]);
                if (f3_eq(me.lo, o.lo) && f3_eq(me.hi, o.hi)) f1[i] = merged[i + 1].f0;
            }
        }
        for (int i = 0; i < ne; ++i) { merged[i].f1 = f1[i]; edges.push_back(merged[i]); can_of.push_back(can_local[i]); }
    }
    // drop edges between coplanar faces
    {
        std::vector<unsigned char> remove(edges.size(), 0);
        parallel_chunks((int)edges.size(), 4096, [&](int begin, int end) {
            for (int i = begin; i < end; ++i) {
                const EdgeD &e = edges[i];
                if (e.f0 == -1 || e.f1 == -1) continue;
                V3 a = edge_v0(shapes, e), b = edge_v1(shapes, e);
                V3 o0 = to_v3(edge_opp0f(shapes, e)), o1 = to_v3(edge_opp1f(shapes, e));
                V3 n0 = normalize(cross(a - o0, b - o0)), n1 = normalize(cross(b - o1, a - o1));
                remove[i] = dot(n0, n1) >= (1 - 1e-6f);
            }
        });
        std::vector<EdgeD> kept;
        std::vector<int> kept_can;
        kept.reserve(edges.size()); kept_can.reserve(edges.size());
        for (size_t i = 0; i < edges.size(); ++i) if (!remove[i]) { kept.push_back(edges[i]); kept_can.push_back(can_of[i]); }
        edges.swap(kept); can_of.swap(kept_can);
    }
    const int ne = (int)edges.size();
    const CameraD &cam = scene.d.cam;
    V3 cam_org = xfm_point(cam.cam_to_world, v3(0));

    // ---- primary edges: PMF = clipped screen-space length of camera silhouettes (:186-214, :299-334)
    if (scene.use_primary_edges) {
        timer.lap("edge list (sort, merge)");
        ed->primary_pmf.assign(ne, 0); ed->primary_cdf.assign(ne, 0);
        parallel_chunks(ne, 4096, [&](int begin, int end) {
            for (int i = begin; i < end; ++i) {
                const EdgeD &e = edges[i];
                V2 s0, s1, c0, c1;
                double w = 0;
                if (project_segment(cam, edge_v0(shapes, e), edge_v1(shapes, e), s0, s1))
                    if (clip_unit_square(s0, s1, c0, c1))
                        if (edge_is_silhouette(shapes, cam_org, e)) w = len(c1 - c0);
                ed->primary_pmf[i] = w;
            }
        });
        double total = 0;
        for (int i = 0; i < ne; ++i) total += ed->primary_pmf[i];          // in edge order, like the reference's running sum
        double run = 0;
        for (int i = 0; i < ne; ++i) { ed->primary_pmf[i] = ed->primary_pmf[i] / total; }
        for (int i = 0; i < ne; ++i) { ed->primary_cdf[i] = run; run += ed->primary_pmf[i]; }
    }

    // ---- secondary edges: the two hierarchies (src/edge_tree.cpp:724-882) ----
    if (scene.use_secondary_edges && ne > 0) {
        const bool host_trees = (scene.build_flags & RDR_BUILD_EDGE_HOST_BUILD) != 0;      // A/B, and the check of one against the other
        ed->device_trees = exec::kDeviceEdgeTrees && !host_trees;
        const bool spatial_only = ed->device_trees;      // the kernels compute the Hough-space bounds themselves (edges_gpu.cpp)
        std::vector<int> cs_ids, ncs_ids;
        std::vector<unsigned char> is_sil(ne, 0);
        std::vector<Box6> bounds(ne);
        parallel_chunks(ne, 4096, [&](int begin, int end) {
        for (int i = begin; i < end; ++i) {
            is_sil[i] = edge_is_silhouette(shapes, cam_org, edges[i]);
            const EdgeD &e = edges[i];
            F3 a = edge_v0f(shapes, e), b = edge_v1f(shapes, e);
            Box6 bx;
            bx.p_min = V3{(double)std::min(a.x, b.x), (double)std::min(a.y, b.y), (double)std::min(a.z, b.z)};
            bx.p_max = V3{(double)std::max(a.x, b.x), (double)std::max(a.y, b.y), (double)std::max(a.z, b.z)};
            if (spatial_only) { bx.d_min = bx.d_max = v3(0); bounds[i] = bx; continue; }
            V3 n0 = edge_n0(shapes, e);
            V3 n1 = (e.f1 == -1) ? -n0 : edge_n1(shapes, e);
            F3 mid = F3{0.5f * (a.x + b.x), 0.5f * (a.y + b.y), 0.5f * (a.z + b.z)};
            V3 p = to_v3(mid) - cam_org;
            double p0d = dot(p, n0), p1d = dot(p, n1);
            V3 h0 = V3{n0.x * p0d, n0.y * p0d, n0.z * p0d}, h1 = V3{n1.x * p1d, n1.y * p1d, n1.z * p1d};
            bx.d_min = vmin(h0, h1); bx.d_max = vmax(h0, h1);
            bounds[i] = bx;
        }
        });
        for (int i = 0; i < ne; ++i) (is_sil[i] ? cs_ids : ncs_ids).push_back(i);
        timer.lap("pmf, bounds, split");
        // The billboard hierarchy of the NEE-mode gather (stages_edge.h: SecEdgeGatherN) needs only the edge bounds and
        // the billboard half-width (two running sums in the reference's order: sequential): both are computed on another
        // thread while this one builds the two reference hierarchies.
        // Boxes: each edge's own spatial bounds grown by the half-width (rounded outwards; the builder pads on top).
        // (the billboard hierarchy only has to be conservative: with the edge list of the previous Scene its topology is kept
        //  and its boxes are refitted; rebuilt when the inner surface area has grown by more than 30 %)
        // The hierarchy is kept over the CANONICAL edges (every merged edge of every shape, whether the current positions
        // keep it in the list or not): its topology then depends on the index buffers alone, and a Scene with the connectivity
        // of the previous one refits it -- whatever the motion did to the order of the list and to which coplanar edges
        // dropped out.  A canonical edge that is not in the current list keeps its place and its box (it is a real edge of
        // the mesh) but gets a leaf record no query accepts (GatherLeaf with an empty Hough interval).
        struct GatherCache { std::vector<EdgeD> canon; rt::BvhHost bvh; };
        static GatherCache *gather_caches = new GatherCache[16];         // one build at a time per device (build_locks above)
        GatherCache *gather_cache = gather_caches + dev_slot;
        rt::BvhHost gather_built;
        double &expand_out = ed->edge_bounds_expand;
        const bool gather_on_device = ed->device_trees && exec::kDeviceBvh;
        EdgeData *edp = ed.get();
        auto gather_job = hostpool::run([&gather_built, &edges, &canon, &can_of, &cs_ids, &ncs_ids, shapes, ne, &expand_out, cache_allowed, gather_on_device, edp, gather_cache] {
            // mean absolute deviation of the endpoints -> billboard half-width
            std::vector<int> all_ids(cs_ids);
            all_ids.insert(all_ids.end(), ncs_ids.begin(), ncs_ids.end());
            V3 mean = v3(0);
            for (int id : all_ids) {
                F3 a = edge_v0f(shapes, edges[id]), b = edge_v1f(shapes, edges[id]);
                mean += to_v3(F3{a.x + b.x, a.y + b.y, a.z + b.z});
            }
            mean = mean / (2. * double(ne));
            V3 mad = v3(0);
            for (int id : all_ids) {
                F3 a = edge_v0f(shapes, edges[id]), b = edge_v1f(shapes, edges[id]);
                V3 aa = V3{fabs(a.x - mean.x), fabs(a.y - mean.y), fabs(a.z - mean.z)};
                V3 bb = V3{fabs(b.x - mean.x), fabs(b.y - mean.y), fabs(b.z - mean.z)};
                mad += aa + bb;
            }
            mad = mad / double(ne);
            const double e = 0.01f * len(mad);
            expand_out = e;

            const int nc = (int)canon.size();
            std::vector<float> boxes((size_t)6 * nc);
            for (int i = 0; i < nc; ++i) {
                const F3 a = edge_v0f(shapes, canon[i]), b = edge_v1f(shapes, canon[i]);       // (the spatial bounds of edges.cpp: Box6::p_min / p_max)
                const double lo[3] = {(double)std::min(a.x, b.x) - e, (double)std::min(a.y, b.y) - e, (double)std::min(a.z, b.z) - e};
                const double hi[3] = {(double)std::max(a.x, b.x) + e, (double)std::max(a.y, b.y) + e, (double)std::max(a.z, b.z) + e};
                for (int k = 0; k < 3; ++k) {
                    boxes[6 * (size_t)i + k] = std::nextafterf((float)lo[k], -std::numeric_limits<float>::infinity());
                    boxes[6 * (size_t)i + 3 + k] = std::nextafterf((float)hi[k], std::numeric_limits<float>::infinity());
                }
            }
            if (gather_on_device) {
                // built / refitted by kernels when the structures are published (publish_edge_data -> gather_hierarchy_device)
                edp->gather_boxes = std::move(boxes);
                edp->gather_cur_of.assign((size_t)nc, -1);
                for (int i = 0; i < ne; ++i) edp->gather_cur_of[can_of[i]] = i;
                edp->gather_canon = canon;
                return;
            }
            bool have = false;
            if (cache_allowed && !gather_cache->bvh.nodes.empty() && gather_cache->canon.size() == canon.size() &&
                std::memcmp(gather_cache->canon.data(), canon.data(), sizeof(EdgeD) * canon.size()) == 0) {
                gather_built = gather_cache->bvh;
                have = rt::refit_box_bvh(gather_built, boxes.data()) <= 1.3;
            }
            if (!have) {
                gather_built = rt::build_box_bvh(boxes.data(), nc);
                gather_cache->canon = canon;
                gather_cache->bvh = gather_built;
            }
            // leaf slots name CURRENT edges from here on (-1: the slot's canonical edge is not in the list)
            std::vector<int> cur_of((size_t)nc, -1);
            for (int i = 0; i < ne; ++i) cur_of[can_of[i]] = i;
            for (size_t sl = 0; sl + 1 < gather_built.ids.size(); sl += 2) gather_built.ids[sl + 1] = cur_of[gather_built.ids[sl + 1]];
        });
        if (ed->device_trees) {
            // The two reference hierarchies are built by kernels on the stream of the first gradient render (edges_gpu.cpp);
            // the host contributes what must carry the host libm's last bit: length x exterior dihedral angle (acos) per edge
            ed->wlen.resize(ne);
            parallel_chunks(ne, 2048, [&](int begin, int end) {
                for (int i = begin; i < end; ++i) {
                    const EdgeD &e = edges[i];
                    ed->wlen[i] = f3_distance(edge_v0f(shapes, e), edge_v1f(shapes, e)) * edge_exterior_dihedral(shapes, e);
                }
            });
            timer.lap("edge weights");
            gather_job.wait();
            ed->cs_ids = cs_ids; ed->ncs_ids = ncs_ids;
            ed->gather = std::move(gather_built);
            ed->gather_refit_allowed = cache_allowed;
            if (ed->gather.depth + 2 > 64) throw std::runtime_error("edge gather hierarchy deeper than the traversal stack (64)");
            if (ed->gather.ids.size() / 2 >= ((size_t)1 << 24) || ed->gather.nodes.size() >= ((size_t)1 << 30))
                throw std::runtime_error("edge gather hierarchy: more than 2^24 edges are not supported");
        } else {
        TreeBuilder cs(true, shapes, edges, bounds), ncs(false, shapes, edges, bounds);
        {   // the two hierarchies are independent
            auto cs_job = hostpool::run([&] { cs.build(cs_ids); });
            ncs.build(ncs_ids);
            cs_job.wait();
        }
        timer.lap("3-D and 6-D trees");
        ed->cs_nodes.swap(cs.nodes); ed->cs_leaves = cs.n;
        ed->ncs_nodes.swap(ncs.nodes); ed->ncs_leaves = ncs.n;
        // One depth-first pass per tree gives (a) its depth -- the NEE-mode walk keeps one pending sibling per level in a
        // fixed stack -- and (b) the order in which the reference's walk reaches the leaves: roots pushed 3-D tree first,
        // 6-D tree second, children pushed 0 then 1, last pushed popped first (src/edge.cpp:1239-1318); a leaf's rank is
        // its position in that order, which is all the order-free gather needs to replay the reservoir.
        std::vector<int> leaf_rank(ne, 0);
        std::vector<double> leaf_dx((size_t)2 * ne, 0.0);
        // (subtrees are independent once their first rank is known: child 1's leaves come first, then child 0's)
        struct Walker {
            const std::vector<EdgeNode> &tree; const std::vector<int> &below; bool hough;
            std::vector<int> &leaf_rank; std::vector<double> &leaf_dx;
            int serial(int top, int top_depth, int rank) const {
                const double inf = std::numeric_limits<double>::infinity();
                int max_depth = 0;
                std::vector<std::pair<int, int>> todo;
                todo.reserve(256);
                todo.push_back({top, top_depth});
                while (!todo.empty()) {
                    auto [node, depth] = todo.back();
                    todo.pop_back();
                    max_depth = std::max(max_depth, depth);
                    const EdgeNode &nd = tree[node];
                    if (nd.edge_id != -1) {
                        leaf_rank[nd.edge_id] = rank++;
                        leaf_dx[2 * (size_t)nd.edge_id] = hough ? nd.d_min.x : -inf;
                        leaf_dx[2 * (size_t)nd.edge_id + 1] = hough ? nd.d_max.x : inf;
                    } else if (nd.child0 >= 0) { todo.push_back({nd.child0, depth + 1}); todo.push_back({nd.child1, depth + 1}); }
                }
                return max_depth;
            }
            int walk(int node, int depth, int rank) const {
                const EdgeNode &nd = tree[node];
                if (nd.edge_id != -1 || nd.child0 < 0 || below[node] < 1024 || depth > 24) return serial(node, depth, rank);
                int d0 = 0;
                auto job = hostpool::run([&] { d0 = walk(nd.child0, depth + 1, rank + below[nd.child1]); });
                const int d1 = walk(nd.child1, depth + 1, rank);
                job.wait();
                return std::max(d0, d1);
            }
        };
        auto walk_tree = [&](const std::vector<EdgeNode> &tree, const std::vector<int> &below, bool hough, int first_rank) {
            if (tree.empty()) return 0;
            return Walker{tree, below, hough, leaf_rank, leaf_dx}.walk(0, 1, first_rank);
        };
        // the 6-D tree is walked first, so the 3-D tree's ranks start after its leaves
        int cs_depth_value = 0;
        auto cs_depth = hostpool::run([&] { cs_depth_value = walk_tree(ed->cs_nodes, cs.leaves_below, false, ed->ncs_leaves); });
        const int ncs_depth_value = walk_tree(ed->ncs_nodes, ncs.leaves_below, true, 0);
        cs_depth.wait();
        const int depths[2] = {ncs_depth_value, cs_depth_value};
        for (int max_depth : depths) {
            if (max_depth == 0) continue;
            if (max_depth + 2 > 64) throw std::runtime_error("edge hierarchy deeper than the traversal stack (64)");
            ed->max_stack = std::max(ed->max_stack, max_depth + 2);
        }
        timer.lap("depth, leaf order");
        {
            gather_job.wait();
            ed->gather = std::move(gather_built);
            if (ed->gather.depth + 2 > 64) throw std::runtime_error("edge gather hierarchy deeper than the traversal stack (64)");
            const size_t slots = ed->gather.ids.size() / 2;
            // stack entries of the gather pack a leaf as (count << 24 | first slot) under bit 30 (stages_edge.h: gather_entry)
            if (slots >= ((size_t)1 << 24) || ed->gather.nodes.size() >= ((size_t)1 << 30))
                throw std::runtime_error("edge gather hierarchy: more than 2^24 edges are not supported");
            ed->gleaf.resize(slots);
            parallel_chunks((int)slots, 2048, [&](int sl_begin, int sl_end) {
            for (size_t sl = (size_t)sl_begin; sl < (size_t)sl_end; ++sl) {
                const int eid = ed->gather.ids[2 * sl + 1];
                GatherLeaf &gl = ed->gleaf[sl];
                if (eid < 0) { gl = dead_gather_leaf(); continue; }
                gl.dx_lo = leaf_dx[2 * (size_t)eid]; gl.dx_hi = leaf_dx[2 * (size_t)eid + 1];
                F3 a = edge_v0f(shapes, edges[eid]), b = edge_v1f(shapes, edges[eid]);
                const EdgeD &e = edges[eid];
                F3 o0 = e.f0 != -1 ? edge_opp0f(shapes, e) : a, o1 = e.f1 != -1 ? edge_opp1f(shapes, e) : b;      // as in EdgeGeom
                gl.v0[0] = a.x; gl.v0[1] = a.y; gl.v0[2] = a.z; gl.v1[0] = b.x; gl.v1[1] = b.y; gl.v1[2] = b.z;
                gl.o0[0] = o0.x; gl.o0[1] = o0.y; gl.o0[2] = o0.z; gl.o1[0] = o1.x; gl.o1[1] = o1.y; gl.o1[2] = o1.z;
                gl.eid = eid; gl.rank = leaf_rank[eid];
                gl.f0 = e.f0 == -1 ? -1 : 0; gl.f1 = e.f1 == -1 ? -1 : 0;
                gl.has_normals = scene.shapes[e.shape_id].normals != nullptr;
            }
            });
        }
        }
        timer.lap("gather hierarchy");
    }

    // ---- what the device will hold, still on the host (publish_edge_data copies it over) ----
    {
        // per-edge geometry records, from the host copies of the shapes
        std::vector<EdgeGeom> &geom = ed->geom;
        geom.resize(edges.size());
        parallel_chunks((int)edges.size(), 4096, [&](int begin, int end) {
        for (int i = begin; i < end; ++i) {
            const EdgeD &e = edges[i];
            EdgeGeom &g = geom[i];
            F3 a = edge_v0f(shapes, e), b = edge_v1f(shapes, e);
            F3 o0 = e.f0 != -1 ? edge_opp0f(shapes, e) : a, o1 = e.f1 != -1 ? edge_opp1f(shapes, e) : b;
            g.v0[0] = a.x; g.v0[1] = a.y; g.v0[2] = a.z; g.v1[0] = b.x; g.v1[1] = b.y; g.v1[2] = b.z;
            g.o0[0] = o0.x; g.o0[1] = o0.y; g.o0[2] = o0.z; g.o1[0] = o1.x; g.o1[1] = o1.y; g.o1[2] = o1.z;
            g.f0 = e.f0; g.f1 = e.f1; g.has_normals = scene.shapes[e.shape_id].normals != nullptr; g.pad = 0;
        }
        });
    }
    timer.lap("edge geometry");
    // [internal | leaves]: n - 1 interior nodes first, then the n leaves (see TreeBuilder)
    auto fatten = [&](const std::vector<EdgeNode> &nodes, int num_leaves, int tree_bit, int &root, std::vector<EdgeNodeP> &out) {
        root = kNoEdgeTree;
        out.clear();
        if (nodes.empty()) return;
        const int num_inner = (int)nodes.size() - num_leaves;
        auto to_f32 = [](double x) {
            float f = (float)x;
            if ((double)f != x) throw std::runtime_error("edge hierarchy: spatial bounds are not fp32 values");   // vertices are fp32
            return f;
        };
        auto ref_of = [&](int idx) { return idx >= num_inner ? ~nodes[idx].edge_id : idx; };
        root = nodes[0].edge_id != -1 ? ~nodes[0].edge_id : (0 | tree_bit);
        if (num_inner <= 0) return;               // a single edge: the root reference is the leaf
        out.resize(num_inner);
        parallel_chunks(num_inner, 4096, [&](int begin, int end) {
        for (int i = begin; i < end; ++i) {
            const EdgeNode &n = nodes[i];
            EdgeNodeP &o = out[i];
            const double lo[3] = {n.p_min.x, n.p_min.y, n.p_min.z}, hi[3] = {n.p_max.x, n.p_max.y, n.p_max.z};
            for (int k = 0; k < 3; ++k) { o.p_min[k] = to_f32(lo[k]); o.p_max[k] = to_f32(hi[k]); }
            const int ch[2] = {n.child0, n.child1};
            for (int c = 0; c < 2; ++c) {
                const EdgeNode &cn = nodes[ch[c]];
                const double clo[3] = {cn.p_min.x, cn.p_min.y, cn.p_min.z}, chi[3] = {cn.p_max.x, cn.p_max.y, cn.p_max.z};
                for (int k = 0; k < 3; ++k) { o.c_pmin[c][k] = to_f32(clo[k]); o.c_pmax[c][k] = to_f32(chi[k]); }
                o.c_dx_min[c] = cn.d_min.x; o.c_dx_max[c] = cn.d_max.x; o.c_wlen[c] = cn.wlen;
                o.c_ref[c] = ref_of(ch[c]);
            }
        }
        });
    };
    ed->d.cs_nodes = ed->d.ncs_nodes = nullptr;
    ed->d.cs_root = ed->d.ncs_root = kNoEdgeTree;
    if (!ed->device_trees) {
        fatten(ed->cs_nodes, ed->cs_leaves, 0, ed->d.cs_root, ed->cs_fat);
        fatten(ed->ncs_nodes, ed->ncs_leaves, kEdgeTreeBit, ed->d.ncs_root, ed->ncs_fat);
    }
    timer.lap("sampler records");
    ed->d.num_edges = ne;
    ed->d.edge_bounds_expand = ed->edge_bounds_expand;
    ed->d.max_stack = ed->max_stack;
    ed->d.cam_org = cam_org;
    ed->d.ltc = scene.ltc_table;
    return ed.release();
}

// Device copies of what compute_edge_data prepared and, in the gfx950 build, the hierarchy kernels (on the calling thread's
// stream; the caller flushes).
void publish_edge_data(EdgeData &ed) {
    PhaseTimer timer("edge publish");
    auto up = [&](const void *src, size_t bytes) -> void * {
        void *p = exec::pool_alloc(bytes);
        ed.owned.push_back(p);
        if (bytes) exec::upload_async(p, src, bytes);
        return p;
    };
    EdgeSceneD &d = ed.d;
    const size_t ne = ed.edges.size();
    d.edges = (const EdgeD *)up(ed.edges.data(), sizeof(EdgeD) * ne);
    d.geom = (const EdgeGeom *)up(ed.geom.data(), sizeof(EdgeGeom) * ed.geom.size());
    d.primary_pmf = ed.primary_pmf.empty() ? nullptr : (const double *)up(ed.primary_pmf.data(), sizeof(double) * ne);
    d.primary_cdf = ed.primary_cdf.empty() ? nullptr : (const double *)up(ed.primary_cdf.data(), sizeof(double) * ne);
    d.gather = rt::BvhD{nullptr, nullptr, nullptr, 0, 0, 2};
    d.gleaf = nullptr;
    if (!ed.gather.nodes.empty()) {
        d.gather.nodes = (const rt::Node *)up(ed.gather.nodes.data(), sizeof(rt::Node) * ed.gather.nodes.size());
        d.gather.num_nodes = (int)ed.gather.nodes.size();
        d.gather.num_tris = (int)(ed.gather.ids.size() / 2);
        d.gather.stack_need = ed.gather.depth + 2;
    }
    if (ed.device_trees) {
        if (d.gather.num_tris > 0) d.gather.ids = (const int *)up(ed.gather.ids.data(), sizeof(int) * ed.gather.ids.size());
        if (!ed.gather_boxes.empty()) gather_hierarchy_device(ed);
        timer.lap("device copies");
        if (!ed.cs_ids.empty() || !ed.ncs_ids.empty()) build_edge_trees_device(ed);
        timer.lap("hierarchies (device)");
        return;
    }
    d.cs_nodes = ed.cs_fat.empty() ? nullptr : (const EdgeNodeP *)up(ed.cs_fat.data(), sizeof(EdgeNodeP) * ed.cs_fat.size());
    d.ncs_nodes = ed.ncs_fat.empty() ? nullptr : (const EdgeNodeP *)up(ed.ncs_fat.data(), sizeof(EdgeNodeP) * ed.ncs_fat.size());
    if (!ed.gather.nodes.empty()) d.gleaf = (const GatherLeaf *)up(ed.gleaf.data(), sizeof(GatherLeaf) * ed.gleaf.size());
    timer.lap("device copies");
}

} // namespace rdr
