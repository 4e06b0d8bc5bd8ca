// edges_gpu.cpp -- the two order-exact edge hierarchies built on the GPU (row 8f-1 of SURVEY.md).
//
// What the reference does on the host for every Scene (src/edge_tree.cpp:724-882: Morton codes, sort, radix tree, bounds,
// treelet restructuring) and what edges.cpp: TreeBuilder restates on host threads, as gfx950 kernels on the stream of the
// first gradient render (scene.cpp: Scene::edge_data -> edges.cpp: publish_edge_data -> build_edge_trees_device).  The
// result must equal the reference's trees link for link and bit for bit (tests/test_edge_build.py compares the dump of a GPU
// build with the oracle's): every quantity is computed by the same fp64 expressions (edges.h: edge_node_area /
// edge_node_merge, shared with the host builder), with no fused contraction, and every choice the reference makes by
// iteration order is made by the same order here:
//   * codes            one thread per edge; scene bounds by a one-workgroup min/max reduction (exact)
//   * sort             our own stable LSD radix sort on the 64-bit code, payload = edge id (input ids ascend, so stable =
//                      the reference's order of equal codes): eight passes of 8 bits -- per-workgroup digit histograms, one
//                      scan, and a scatter that ranks equal digits inside a wave by ballots (radix_* kernels below)
//   * radix tree       one thread per interior node (Karras 2012), ties by edge id
//   * bounds           one thread per leaf climbs; whoever reaches a node second finds both children complete (counters)
//   * treelets         in order of radix-tree height, one launch per height, one WAVE per node: the wave restructures the
//                      node's 7-leaf treelet (Karras & Aila 2013, Algorithm 2) entirely in LDS -- lane 0 grows the treelet,
//                      the 128 subset areas and the dynamic programme over subsets run two subsets per lane (rounds by subset
//                      size), lane 0 rebuilds, all lanes write the records back
//   * leaf order       one thread per leaf climbs to the root summing the leaf counts of the subtrees the reference's walk
//                      visits before it (its rank in that walk, stages_edge.h: the gather's replay) and its depth
//   * records          the samplers' 128-byte interior records (EdgeNodeP) and the gather's per-slot leaf records
// The per-edge weight (length x exterior dihedral angle) comes from the host: it goes through acos(), and only the host
// libm's last bit is the oracle's.
#include "edges.h"
#include "scene.h"

#include <algorithm>
#include <cstring>
#include <limits>
#include <mutex>
#include <stdexcept>

namespace rdr {
namespace {

struct Box6D { V3 p_min, p_max, d_min, d_max; };

// ---- stable LSD radix sort of (64-bit key, int payload) pairs, 8 bits per pass, wave64-native ------------------------------
// Per pass: (1) radix_hist: every 256-thread workgroup counts the digits of its 256 x kSortItems keys in LDS and writes the
// counts digit-major (hist[digit * blocks + block]); (2) radix_scan: one workgroup turns that table into exclusive offsets --
// digit-major order IS the output order: all keys of digit 0 by block, then digit 1, ...; (3) radix_scatter: a workgroup
// walks its keys in input order, 256 at a time: the lanes of a wave that hold the same digit find each other with eight
// ballots (one per digit bit), a lane's rank among them is a popcount below its lane, the waves' counts per digit are
// combined through LDS in wave order, and a running per-digit base carries over to the next 256 keys.  Equal digits keep
// their input order at every step, so every pass -- and the whole sort -- is stable.
constexpr int kSortItems = 8;                      // keys per thread and pass
constexpr int kSortTile = 256 * kSortItems;
__global__ void __launch_bounds__(256) radix_hist_kernel(const uint64_t *keys, int n, int shift, int blocks, int *hist) {
    __shared__ int cnt[256];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * kSortTile;
    for (int it = 0; it < kSortItems; ++it) {
        const int i = base + it * 256 + threadIdx.x;
        if (i < n) atomicAdd(&cnt[(int)((keys[i] >> shift) & 255u)], 1);
    }
    __syncthreads();
    hist[threadIdx.x * blocks + blockIdx.x] = cnt[threadIdx.x];
}
__global__ void __launch_bounds__(256) radix_scan_kernel(int *hist, int total) {       // exclusive scan of `total` ints, one workgroup
    __shared__ int part[256];
    const int per = (total + 255) / 256;
    const int beg = threadIdx.x * per, end = min(beg + per, total);
    int s = 0;
    for (int i = beg; i < end; ++i) s += hist[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { int run = 0; for (int i = 0; i < 256; ++i) { const int t = part[i]; part[i] = run; run += t; } }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = beg; i < end; ++i) { const int t = hist[i]; hist[i] = run; run += t; }
}
__global__ void __launch_bounds__(256) radix_scatter_kernel(const uint64_t *keys_in, const int *vals_in, int n, int shift, int blocks,
                                                            const int *offsets, uint64_t *keys_out, int *vals_out) {
    __shared__ int digit_base[256];            // where the next key of each digit goes
    __shared__ int wave_cnt[4][256];           // this round: keys of each digit held by each wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    digit_base[threadIdx.x] = offsets[threadIdx.x * blocks + blockIdx.x];
    const int base = blockIdx.x * kSortTile;
    for (int it = 0; it < kSortItems; ++it) {
        for (int w = 0; w < 4; ++w) wave_cnt[w][threadIdx.x] = 0;
        __syncthreads();
        const int i = base + it * 256 + threadIdx.x;
        const bool valid = i < n;
        uint64_t key = 0; int val = 0, digit = 0;
        if (valid) { key = keys_in[i]; val = vals_in[i]; digit = (int)((key >> shift) & 255u); }
        unsigned long long same = __ballot(valid);
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot(valid && ((digit >> b) & 1));
            same &= ((digit >> b) & 1) ? m : ~m;
        }
        const int rank = __popcll(same & ((1ull << lane) - 1ull));
        if (valid && rank == 0) wave_cnt[wave][digit] = __popcll(same);       // the first lane of each digit group
        __syncthreads();
        if (valid) {
            int before = 0;
            for (int w = 0; w < wave; ++w) before += wave_cnt[w][digit];
            const int at = digit_base[digit] + before + rank;
            keys_out[at] = key; vals_out[at] = val;
        }
        __syncthreads();
        digit_base[threadIdx.x] += wave_cnt[0][threadIdx.x] + wave_cnt[1][threadIdx.x] + wave_cnt[2][threadIdx.x] + wave_cnt[3][threadIdx.x];
        __syncthreads();
    }
}
// sorts n pairs from (keys_in, vals_in) into (keys_out, vals_out); (keys_tmp, vals_tmp): n entries of scratch; hist: 256 x blocks ints
inline void radix_sort_pairs_u64(hipStream_t s, uint64_t *keys_tmp, int *vals_tmp, const uint64_t *keys_in, const int *vals_in,
                                 uint64_t *keys_out, int *vals_out, int n, int *hist) {
    const int blocks = (n + kSortTile - 1) / kSortTile;
    // pass k reads src_k and writes dst_k: in -> tmp -> out -> tmp -> out -> tmp -> out -> tmp -> out (8 passes)
    const uint64_t *ksrc = keys_in; const int *vsrc = vals_in;
    for (int pass = 0; pass < 8; ++pass) {
        uint64_t *kdst = (pass & 1) ? keys_out : keys_tmp;
        int *vdst = (pass & 1) ? vals_out : vals_tmp;
        hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ksrc, n, 8 * pass, blocks, hist);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(256), 0, s, hist, 256 * blocks);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ksrc, vsrc, n, 8 * pass, blocks, (const int *)hist, kdst, vdst);
        ksrc = kdst; vsrc = vdst;
    }
}

__device__ inline V3 vmin_std(V3 a, V3 b) { return V3{dmin_std(a.x, b.x), dmin_std(a.y, b.y), dmin_std(a.z, b.z)}; }
__device__ inline V3 vmax_std(V3 a, V3 b) { return V3{dmax_std(a.x, b.x), dmax_std(a.y, b.y), dmax_std(a.z, b.z)}; }

// ---- per-edge 6-D bounds (src/edge_tree.cpp:25-74; edges.cpp: compute_edge_data does the same for the host builder) ----
__global__ void __launch_bounds__(256) edge_bounds_kernel(const EdgeGeom *geom, int ne, V3 cam_org, Box6D *bounds) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ne) return;
    const EdgeGeom g = geom[i];
    const F3 a = F3{g.v0[0], g.v0[1], g.v0[2]}, b = F3{g.v1[0], g.v1[1], g.v1[2]};
    Box6D bx;
    bx.p_min = V3{(double)fminf_std(a.x, b.x), (double)fminf_std(a.y, b.y), (double)fminf_std(a.z, b.z)};
    bx.p_max = V3{(double)fmaxf_std(a.x, b.x), (double)fmaxf_std(a.y, b.y), (double)fmaxf_std(a.z, b.z)};
    const V3 av = to_v3(a), bv = to_v3(b);
    const V3 o0 = V3{(double)g.o0[0], (double)g.o0[1], (double)g.o0[2]}, o1 = V3{(double)g.o1[0], (double)g.o1[1], (double)g.o1[2]};
    V3 n0 = cross(av - o0, bv - o0);
    { const double l2 = len_sq(n0); n0 = l2 < 1e-20 ? v3(0) : n0 / sqrt(l2); }
    V3 n1;
    if (g.f1 == -1) n1 = -n0;
    else { n1 = cross(bv - o1, av - o1); const double l2 = len_sq(n1); n1 = l2 < 1e-20 ? v3(0) : n1 / sqrt(l2); }
    const F3 mid = F3{0.5f * (a.x + b.x), 0.5f * (a.y + b.y), 0.5f * (a.z + b.z)};
    const V3 p = to_v3(mid) - cam_org;
    const double p0d = dot(p, n0), p1d = dot(p, n1);
    const V3 h0 = V3{n0.x * p0d, n0.y * p0d, n0.z * p0d}, h1 = V3{n1.x * p1d, n1.y * p1d, n1.z * p1d};
    bx.d_min = vmin_std(h0, h1); bx.d_max = vmax_std(h0, h1);
    bounds[i] = bx;
}

// ---- bounds of one tree's edges: exact min / max, partial results per workgroup, then one workgroup over the partials ----
// (ids == nullptr: `bounds` holds the partials themselves)
__global__ void __launch_bounds__(256) scene_bounds_kernel(const Box6D *bounds, const int *ids, int n, Box6D *out) {
    __shared__ double lo[6][256], hi[6][256];
    const double inf = INFINITY;
    double l[6] = {inf, inf, inf, inf, inf, inf}, h[6] = {-inf, -inf, -inf, -inf, -inf, -inf};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const Box6D b = bounds[ids ? ids[i] : i];
        const double bl[6] = {b.p_min.x, b.p_min.y, b.p_min.z, b.d_min.x, b.d_min.y, b.d_min.z};
        const double bh[6] = {b.p_max.x, b.p_max.y, b.p_max.z, b.d_max.x, b.d_max.y, b.d_max.z};
#pragma unroll
        for (int k = 0; k < 6; ++k) { l[k] = dmin_std(l[k], bl[k]); h[k] = dmax_std(h[k], bh[k]); }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) { lo[k][threadIdx.x] = l[k]; hi[k][threadIdx.x] = h[k]; }
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                lo[k][threadIdx.x] = dmin_std(lo[k][threadIdx.x], lo[k][threadIdx.x + s]);
                hi[k][threadIdx.x] = dmax_std(hi[k][threadIdx.x], hi[k][threadIdx.x + s]);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        Box6D o;
        o.p_min = V3{lo[0][0], lo[1][0], lo[2][0]}; o.d_min = V3{lo[3][0], lo[4][0], lo[5][0]};
        o.p_max = V3{hi[0][0], hi[1][0], hi[2][0]}; o.d_max = V3{hi[3][0], hi[4][0], hi[5][0]};
        out[blockIdx.x] = o;
    }
}

// ---- Morton codes (src/edge_tree.cpp:76-140) ----
__device__ inline uint64_t expand3(uint64_t x) {
    x &= 0x1fffff;
    x = (x | x << 32) & 0x1f00000000ffff;
    x = (x | x << 16) & 0x1f0000ff0000ff;
    x = (x | x << 8) & 0x100f00f00f00f00f;
    x = (x | x << 4) & 0x10c30c30c30c30c3;
    x = (x | x << 2) & 0x1249249249249249;
    return x;
}
__device__ inline uint64_t expand6(uint64_t x) {
    uint64_t r = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) r |= ((x >> i) & 1ull) << (6 * i);
    return r;
}
__device__ inline double unit_coord(double v, double lo, double hi) {
    if (hi - lo <= 0.f) return 0.5f;
    return (v - lo) / (hi - lo);
}
__global__ void __launch_bounds__(256) codes_kernel(const Box6D *bounds, const int *ids, int n, const Box6D *scene, int is3d,
                                                    uint64_t *codes) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Box6D sb = *scene;
    const Box6D b = bounds[ids[i]];
    const V3 pc = 0.5f * (b.p_min + b.p_max);
    uint64_t code;
    if (is3d) {
        const double s = (1 << 21) - 1;
        const uint64_t x = (uint64_t)(unit_coord(pc.x, sb.p_min.x, sb.p_max.x) * s);
        const uint64_t y = (uint64_t)(unit_coord(pc.y, sb.p_min.y, sb.p_max.y) * s);
        const uint64_t z = (uint64_t)(unit_coord(pc.z, sb.p_min.z, sb.p_max.z) * s);
        code = (expand3(x) << 2u) | (expand3(y) << 1u) | expand3(z);
    } else {
        const V3 dc = 0.5f * (b.d_min + b.d_max);
        const uint64_t px = (uint64_t)(unit_coord(pc.x, sb.p_min.x, sb.p_max.x) * 1023);
        const uint64_t py = (uint64_t)(unit_coord(pc.y, sb.p_min.y, sb.p_max.y) * 1023);
        const uint64_t pz = (uint64_t)(unit_coord(pc.z, sb.p_min.z, sb.p_max.z) * 1023);
        const uint64_t dx = (uint64_t)(unit_coord(dc.x, sb.d_min.x, sb.d_max.x) * 1023);
        const uint64_t dy = (uint64_t)(unit_coord(dc.y, sb.d_min.y, sb.d_max.y) * 1023);
        const uint64_t dz = (uint64_t)(unit_coord(dc.z, sb.d_min.z, sb.d_max.z) * 1023);
        code = (expand6(px) << 5u) | (expand6(py) << 4u) | (expand6(pz) << 3u) | (expand6(dx) << 2u) | (expand6(dy) << 1u) | expand6(dz);
    }
    codes[i] = code;
}

// ---- nodes: [n_internal interior | n leaves] ----
constexpr int kMaxLevels = 128;                    // radix-tree height bound: 64 code bits + 32 id bits, with room
struct TreeD {
    EdgeNode *nodes; int *below; int *counter;     // per node: the record, leaves in its subtree, arrival counter
    int *height;                                   // per node: height in the radix tree (leaves 0), the treelet pass's schedule
    int *level_count;                              // [kMaxLevels] interior nodes per height, [kMaxLevels] = the root's height
    int *level_list;                               // interior nodes grouped by height
    const uint64_t *codes; const int *ids;         // sorted
    int n, n_internal, is3d;
};

__global__ void __launch_bounds__(256) init_nodes_kernel(TreeD t, const Box6D *bounds, const double *wlen) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= t.n_internal + t.n) return;
    const double inf = INFINITY;
    EdgeNode nd;
    nd.p_min = nd.d_min = v3(inf); nd.p_max = nd.d_max = v3(-inf);
    nd.wlen = 0; nd.cost = 0; nd.parent = -1; nd.child0 = nd.child1 = -1; nd.edge_id = -1;
    if (i >= t.n_internal) {
        const int id = t.ids[i - t.n_internal];
        const Box6D b = bounds[id];
        nd.p_min = b.p_min; nd.p_max = b.p_max;
        if (!t.is3d) { nd.d_min = b.d_min; nd.d_max = b.d_max; }
        nd.wlen = wlen[id];
        nd.edge_id = id;
        nd.cost = edge_node_area(nd, t.is3d != 0);
    }
    // (the radix-tree kernel, which runs after this one, fills in the links)
    t.nodes[i] = nd;
    t.below[i] = 1;
    t.counter[i] = 0;
    t.height[i] = 0;
}

__device__ inline int lcp(const TreeD &t, int i, int j) {
    if (i < 0 || i >= t.n || j < 0 || j >= t.n) return -1;
    const uint64_t a = t.codes[i], b = t.codes[j];
    if (a == b) return 64 + __clzll((long long)((uint64_t)t.ids[i] ^ (uint64_t)t.ids[j]));
    return __clzll((long long)(a ^ b));
}
// Karras radix tree over the sorted codes; every interior node is independent (src/edge_tree.cpp:142-236)
__global__ void __launch_bounds__(256) radix_tree_kernel(TreeD t) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= t.n - 1) return;
    const int d = (lcp(t, idx, idx + 1) - lcp(t, idx, idx - 1) >= 0) ? 1 : -1;
    const int dmin = lcp(t, idx, idx - d);
    int lmax = 2;
    while (lcp(t, idx, idx + lmax * d) > dmin) lmax *= 2;
    int l = 0, divider = 2;
    for (int s = lmax / divider; s >= 1;) {
        if (lcp(t, idx, idx + (l + s) * d) > dmin) l += s;
        if (s == 1) break;
        divider *= 2;
        s = lmax / divider;
    }
    const int j = idx + l * d;
    const int dnode = lcp(t, idx, j);
    int sp = 0;
    divider = 2;
    for (int s = (l + (divider - 1)) / divider; s >= 1;) {
        if (lcp(t, idx, idx + (sp + s) * d) > dnode) sp += s;
        if (s == 1) break;
        divider *= 2;
        s = (l + (divider - 1)) / divider;
    }
    const int gamma = idx + sp * d + (d < 0 ? d : 0);
    const int lo = idx < j ? idx : j, hi = idx < j ? j : idx;
    const int c0 = lo == gamma ? t.n_internal + gamma : gamma;
    const int c1 = hi == gamma + 1 ? t.n_internal + gamma + 1 : gamma + 1;
    t.nodes[idx].child0 = c0; t.nodes[idx].child1 = c1;
    t.nodes[c0].parent = idx; t.nodes[c1].parent = idx;
}

// bottom-up bounds / weights / leaf counts: whoever reaches a node second finds both children complete
__global__ void __launch_bounds__(256) bounds_up_kernel(TreeD t) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= t.n) return;
    int cur = t.nodes[t.n_internal + i].parent;
    while (cur >= 0) {
        __threadfence();
        if (atomicAdd(&t.counter[cur], 1) == 0) break;
        __threadfence();
        EdgeNode *nd = &t.nodes[cur];
        const int c0 = nd->child0, c1 = nd->child1;
        const EdgeNode a = t.nodes[c0], b = t.nodes[c1];
        EdgeNode m = *nd;
        edge_node_merge(m, a, b, t.is3d != 0);
        m.wlen = a.wlen + b.wlen;
        *nd = m;
        t.below[cur] = t.below[c0] + t.below[c1];
        const int h0 = t.height[c0], h1 = t.height[c1];
        int h = 1 + (h0 > h1 ? h0 : h1);
        if (h >= kMaxLevels) h = kMaxLevels - 1;                      // (reported by the host: the build gives up)
        t.height[cur] = h;
        if (m.parent < 0) t.level_count[kMaxLevels] = h;
        cur = m.parent;
    }
}
__global__ void __launch_bounds__(256) level_histogram_kernel(TreeD t) {
    __shared__ int hist[kMaxLevels];
    if (threadIdx.x < kMaxLevels) hist[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < t.n - 1) atomicAdd(&hist[t.height[i]], 1);
    __syncthreads();
    if (threadIdx.x < kMaxLevels && hist[threadIdx.x] > 0) atomicAdd(&t.level_count[threadIdx.x], hist[threadIdx.x]);
}
__global__ void single_leaf_root_kernel(TreeD t) {          // n == 1: the root is a copy of the leaf
    if (threadIdx.x == 0 && blockIdx.x == 0) { t.nodes[0] = t.nodes[t.n_internal]; t.below[0] = 1; }
}
__global__ void __launch_bounds__(256) reset_counters_kernel(int *counter, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) counter[i] = 0;
}

// ---- treelet restructuring, one wave per climbing leaf (behaviour: src/edge_tree.cpp:464-711; formulation: edges.cpp) ----
// Everything a restructuring touches lives in LDS: the records of the treelet's leaves (read-only apart from their parent
// link), of its interior nodes and of its root (rebuilt), and the leaf counts.  Global memory is read once per node while the
// treelet grows and written once when the rebuilt records go back.
struct TreeletShared {
    EdgeNode leaf[7];              // the treelet's leaves (copies)
    EdgeNode rec[6];               // its interior nodes in hand-out order, [5] = the root
    double sa[128], best[128];
    unsigned char best_left[128];
    int leaves[7], inner[6];       // node indices; inner[5] = the root
    int leaf_below[7], rec_below[6];
    int leaf_parent[7];
    double leaf_area[7];           // of the interior ones among the leaves (the growth opens the largest)
    struct Frame { unsigned char part; signed char side; unsigned char stage; signed char parent_slot, slot, c0, c1; } frames[16];
    unsigned char list[64];        // the subsets of one size (dynamic programme)
    int list_n;
    int nl, ni, open;
};
constexpr int kRootSlot = 5;

// A child reference inside the rebuild: 0..6 = treelet leaf i, 8..13 = interior record (slot - 8)
__device__ inline const EdgeNode &slot_node(const TreeletShared &sh, int slot) { return slot < 8 ? sh.leaf[slot] : sh.rec[slot - 8]; }
__device__ inline int slot_below(const TreeletShared &sh, int slot) { return slot < 8 ? sh.leaf_below[slot] : sh.rec_below[slot - 8]; }
__device__ inline int slot_index(const TreeletShared &sh, int slot) { return slot < 8 ? sh.leaves[slot] : sh.inner[slot - 8]; }

// lane 0: rebuild the treelet under the root from best_left (interior nodes handed out in the reference's order: a node
// first, then everything under its RIGHT half, then its left half; at the root the left half comes first).  Bounds, weights
// and costs are refreshed in post-order on the way out.
__device__ inline void rebuild_treelet(TreeletShared &sh, int full, bool is3d) {
    using Frame = TreeletShared::Frame;
    Frame *st = sh.frames;              // (in LDS: a private array indexed at run time would live in scratch memory)
    int sp = 0, next_inner = 0;
    int root_c0 = -1, root_c1 = -1;
    auto refresh = [&](int r, int c0, int c1) {
        EdgeNode &nd = sh.rec[r];
        const EdgeNode &a = slot_node(sh, c0), &b = slot_node(sh, c1);
        edge_node_merge(nd, a, b, is3d);
        nd.wlen = a.wlen + b.wlen;
        nd.cost = edge_node_area(nd, is3d) + a.cost + b.cost;
        nd.child0 = slot_index(sh, c0); nd.child1 = slot_index(sh, c1);
        sh.rec_below[r] = slot_below(sh, c0) + slot_below(sh, c1);
        const int me = sh.inner[r];
        if (c0 < 8) sh.leaf_parent[c0] = me; else sh.rec[c0 - 8].parent = me;
        if (c1 < 8) sh.leaf_parent[c1] = me; else sh.rec[c1 - 8].parent = me;
    };
    const unsigned char left = sh.best_left[full], right = (unsigned char)(full & ~left);
    // frames are popped last-in first-out: push the right half first so that the left half is processed first
    st[sp++] = Frame{right, 1, 0, (signed char)(8 + kRootSlot), -1, -1, -1};
    st[sp++] = Frame{left, 0, 0, (signed char)(8 + kRootSlot), -1, -1, -1};
    while (sp > 0) {
        Frame &f = st[sp - 1];
        int done_slot = -1;
        if ((f.part & (f.part - 1)) == 0) {               // one leaf
            done_slot = __builtin_ctz((unsigned)f.part);
        } else if (f.stage == 0) {
            f.slot = (signed char)(8 + next_inner++);
            f.stage = 1;
            const unsigned char l = sh.best_left[f.part], r = (unsigned char)(f.part & ~l);
            const signed char me = f.slot;
            // right half first, then the left half (pushed in reverse)
            st[sp++] = Frame{l, 0, 0, me, -1, -1, -1};
            st[sp++] = Frame{r, 1, 0, me, -1, -1, -1};
            continue;
        } else {
            refresh(f.slot - 8, f.c0, f.c1);
            done_slot = f.slot;
        }
        // hand the finished subtree to its parent frame (or to the root)
        const int parent_slot = f.parent_slot, side = f.side;
        --sp;
        if (parent_slot == 8 + kRootSlot) { if (side == 0) root_c0 = done_slot; else root_c1 = done_slot; }
        else {
            // the parent frame is the nearest frame below with that slot
            for (int k = sp - 1; k >= 0; --k) if (st[k].slot == parent_slot) { if (side == 0) st[k].c0 = (signed char)done_slot; else st[k].c1 = (signed char)done_slot; break; }
        }
    }
    refresh(kRootSlot, root_c0, root_c1);
}

__device__ inline void treelet_optimize_wave(const TreeD &t, TreeletShared &sh, int root, int lane) {
    const bool is3d = t.is3d != 0;
    // grow the treelet: repeatedly open the inner leaf with the largest box; it is replaced by the last leaf, its children go
    // to the end (this fixes the leaf numbering the partitions are expressed in).  An expansion costs one round of two loads
    // (the opened node's children), the choice scans LDS.
    if (lane == 2) { sh.rec[kRootSlot] = t.nodes[root]; sh.inner[kRootSlot] = root; }
    if (lane < 2) {
        const int c = lane == 0 ? t.nodes[root].child0 : t.nodes[root].child1;
        sh.leaves[lane] = c;
        const EdgeNode nd = t.nodes[c];
        sh.leaf[lane] = nd;
        sh.leaf_area[lane] = nd.edge_id != -1 ? -2.0 : edge_node_area(nd, is3d);        // -2: a real leaf, never opened
        sh.leaf_below[lane] = t.below[c];
    }
    if (lane == 0) { sh.nl = 2; sh.ni = 0; }
    __syncthreads();
    for (;;) {
        const int nl_now = sh.nl;
        if (nl_now >= 7) break;
        if (lane == 0) {
            int widest = -1;
            double widest_area = -1;
            for (int i = 0; i < nl_now; ++i) {
                const double ar = sh.leaf_area[i];
                if (ar > widest_area) { widest_area = ar; widest = i; }
            }
            sh.open = widest;
            if (widest >= 0) {
                const int c0 = sh.leaf[widest].child0, c1 = sh.leaf[widest].child1;
                const int ni = sh.ni;
                sh.inner[ni] = sh.leaves[widest];
                sh.rec[ni] = sh.leaf[widest];
                sh.ni = ni + 1;
                sh.leaves[widest] = sh.leaves[nl_now - 1];
                sh.leaf[widest] = sh.leaf[nl_now - 1];
                sh.leaf_area[widest] = sh.leaf_area[nl_now - 1];
                sh.leaf_below[widest] = sh.leaf_below[nl_now - 1];
                sh.leaves[nl_now - 1] = c0;
                sh.leaves[nl_now] = c1;
                sh.nl = nl_now + 1;
            }
        }
        __syncthreads();
        if (sh.open < 0) break;
        if (lane < 2) {
            const int c = sh.leaves[nl_now - 1 + lane];
            const EdgeNode nd = t.nodes[c];
            sh.leaf[nl_now - 1 + lane] = nd;
            sh.leaf_area[nl_now - 1 + lane] = nd.edge_id != -1 ? -2.0 : edge_node_area(nd, is3d);
            sh.leaf_below[nl_now - 1 + lane] = t.below[c];
        }
        __syncthreads();
    }
    const int nl = sh.nl;
    const int full = (1 << nl) - 1;
    // Surface area of every leaf subset.  [quirk] the reference starts every subset's union from leaf 0's box whether or not
    // leaf 0 is in the subset (src/edge_tree.cpp:560-568): the area it prices subset s with is that of s | 1
    for (int s = lane + 1; s <= full; s += 64) {
        EdgeNode box = sh.leaf[0];
        for (int i = 1; i < nl; ++i) if ((s >> i) & 1) { const EdgeNode o = sh.leaf[i]; edge_node_merge(box, box, o, is3d); }
        sh.sa[s] = edge_node_area(box, is3d);
    }
    if (lane < nl) sh.best[1 << lane] = sh.leaf[lane].cost;
    __syncthreads();
    // Cheapest split of every subset with >= 2 leaves, by subset size; ties go to the first split in the reference's order
    // p = (p - d) & s with d = s without its lowest leaf, i.e. the j-th split puts the bits of j into the positions of d's set
    // bits (j = 1, 2, ...).  Sizes 2 and 3: a lane per subset.  From size 4 on a subset's 2^(size-1) - 1 splits are evaluated
    // by a group of 8 / 16 / 32 / 64 lanes at once and reduced to their minimum; among equal minima the lowest j wins, which
    // is the first in that order -- the same choice as the sequential scan with a strict comparison.
    for (int k = 2; k <= nl; ++k) {
        if (k < 4) {
            for (int s = lane + 1; s <= full; s += 64) {
                if (__popc((unsigned)s) != k) continue;
                double cheapest = INFINITY;
                unsigned arg = 0;
                const unsigned d = ((unsigned)s - 1u) & (unsigned)s;
                unsigned p = (0u - d) & (unsigned)s;
                do {
                    const double c = sh.best[p] + sh.best[(unsigned)s ^ p];
                    if (c < cheapest) { cheapest = c; arg = p; }
                    p = (p - d) & (unsigned)s;
                } while (p != 0);
                sh.best[s] = sh.sa[s] + cheapest;
                sh.best_left[s] = (unsigned char)arg;
            }
            __syncthreads();
            continue;
        }
        // the subsets of this size, in increasing order
        {
            const int s0 = lane + 1, s1 = lane + 65;
            const bool f0 = s0 <= full && __popc((unsigned)s0) == k, f1 = s1 <= full && __popc((unsigned)s1) == k;
            const unsigned long long b0 = __ballot(f0), b1 = __ballot(f1);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (f0) sh.list[__popcll(b0 & below)] = (unsigned char)s0;
            if (f1) sh.list[__popcll(b0) + __popcll(b1 & below)] = (unsigned char)s1;
            if (lane == 0) sh.list_n = __popcll(b0) + __popcll(b1);
        }
        __syncthreads();
        const int count = sh.list_n;
        const int group = 1 << (k - 1);              // lanes per subset: 8, 16, 32, 64
        const int per_pass = 64 / group;
        const int g = lane / group, j = lane % group;            // this lane evaluates split j + 1 of its group's subset
        for (int first = 0; first < count; first += per_pass) {
            const bool has = first + g < count;
            const unsigned s = has ? sh.list[first + g] : 0u;
            const unsigned d = (s - 1u) & s;
            unsigned p = 0;
            { unsigned m = d, x = (unsigned)j + 1u; while (m) { const unsigned low = m & (0u - m); if (x & 1u) p |= low; x >>= 1; m &= m - 1u; } }
            const bool valid = has && j < group - 1;
            const double c = valid ? sh.best[p] + sh.best[s ^ p] : INFINITY;
            double m = c;
            for (int step = 1; step < group; step <<= 1) { const double o = __shfl_xor(m, step, 64); m = o < m ? o : m; }
            const unsigned long long hit = __ballot(valid && c == m && c < INFINITY);
            const unsigned long long mine = group == 64 ? hit : (hit >> (g * group)) & ((1ull << group) - 1ull);
            if (valid && mine != 0ull && j == __builtin_ctzll(mine)) {
                sh.best[s] = sh.sa[s] + m;
                sh.best_left[s] = (unsigned char)p;
            } else if (has && mine == 0ull && j == 0) {          // no finite split (the sequential scan keeps its initial values)
                sh.best[s] = sh.sa[s] + INFINITY;
                sh.best_left[s] = 0;
            }
        }
        __syncthreads();
    }
    if (lane == 0) rebuild_treelet(sh, full, is3d);
    __syncthreads();
    // the rebuilt records go back: ni interior nodes + the root, 16 eight-byte words each; the leaves' parent links
    {
        const int ni = sh.ni;
        for (int w = lane; w < 16 * (ni + 1); w += 64) {
            const int r = w >> 4 < ni ? w >> 4 : kRootSlot;
            reinterpret_cast<unsigned long long *>(&t.nodes[sh.inner[r]])[w & 15] = reinterpret_cast<const unsigned long long *>(&sh.rec[r])[w & 15];
        }
        if (lane <= ni) { const int r = lane < ni ? lane : kRootSlot; t.below[sh.inner[r]] = sh.rec_below[r]; }
        if (lane < nl) t.nodes[sh.leaves[lane]].parent = sh.leaf_parent[lane];
    }
    __syncthreads();
}

// The schedule: a node may be restructured once everything below it in the RADIX tree has been (a restructuring only moves
// nodes inside the subtree of its root), i.e. in order of radix-tree height.  One launch per height, one wave per node of
// that height, both trees in the same launch; the launch boundary is the only synchronisation (climbing waves with arrival
// counters need a device-scope release / acquire per level -- an L2 write-back and invalidate on a part with one L2 per XCD:
// 3 ms for the 21 k-edge tree of the benchmark scene against 0.5 ms this way).
__global__ void __launch_bounds__(256) level_scatter_kernel(TreeD t, const int *level_offset, int *cursor) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= t.n - 1) return;
    const int h = t.height[i];
    t.level_list[level_offset[h] + atomicAdd(&cursor[h], 1)] = i;
}
__global__ void __launch_bounds__(64) treelet_level_kernel(TreeD t0, int first0, int count0, TreeD t1, int first1) {
    __shared__ TreeletShared sh;
    const bool second = (int)blockIdx.x >= count0;
    const TreeD &t = second ? t1 : t0;
    const int node = second ? t1.level_list[first1 + (int)blockIdx.x - count0] : t0.level_list[first0 + (int)blockIdx.x];
    treelet_optimize_wave(t, sh, node, threadIdx.x);
}

// ---- order in which the reference's walk reaches the leaves, and the depth of the tree ----
__global__ void __launch_bounds__(256) leaf_rank_kernel(TreeD t, int first_rank, int hough, int *leaf_rank, double *leaf_dx, int *max_depth) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= t.n) return;
    int node = t.n == 1 ? 0 : t.n_internal + i;
    const EdgeNode lf = t.nodes[node];
    int rank = first_rank, depth = 1;
    int cur = t.n == 1 ? -1 : lf.parent;
    while (cur >= 0) {
        const EdgeNode nd = t.nodes[cur];
        if (nd.child0 == node) rank += t.below[nd.child1];       // child 1 is visited first
        ++depth;
        node = cur;
        cur = nd.parent;
    }
    leaf_rank[lf.edge_id] = rank;
    leaf_dx[2 * (size_t)lf.edge_id] = hough ? lf.d_min.x : -INFINITY;
    leaf_dx[2 * (size_t)lf.edge_id + 1] = hough ? lf.d_max.x : INFINITY;
    atomicMax(max_depth, depth);
}

// ---- the samplers' interior records ----
__global__ void __launch_bounds__(256) fatten_kernel(TreeD t, EdgeNodeP *out, int *not_fp32) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int num_inner = t.n - 1;
    if (i >= num_inner) return;
    const EdgeNode n = t.nodes[i];
    EdgeNodeP o;
    bool bad = false;
    auto to_f32 = [&](double x) { const float f = (float)x; bad = bad || (double)f != x; return f; };
    const double lo[3] = {n.p_min.x, n.p_min.y, n.p_min.z}, hi[3] = {n.p_max.x, n.p_max.y, n.p_max.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) { o.p_min[k] = to_f32(lo[k]); o.p_max[k] = to_f32(hi[k]); }
    const int ch[2] = {n.child0, n.child1};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const EdgeNode cn = t.nodes[ch[c]];
        const double clo[3] = {cn.p_min.x, cn.p_min.y, cn.p_min.z}, chi[3] = {cn.p_max.x, cn.p_max.y, cn.p_max.z};
#pragma unroll
        for (int k = 0; k < 3; ++k) { o.c_pmin[c][k] = to_f32(clo[k]); o.c_pmax[c][k] = to_f32(chi[k]); }
        o.c_dx_min[c] = cn.d_min.x; o.c_dx_max[c] = cn.d_max.x; o.c_wlen[c] = cn.wlen;
        o.c_ref[c] = ch[c] >= num_inner ? ~cn.edge_id : ch[c];
    }
    out[i] = o;
    if (bad) *not_fp32 = 1;
}

// ---- the gather's per-slot leaf records ----
__global__ void __launch_bounds__(256) gather_leaf_kernel(const int *slot_ids, int slots, const EdgeGeom *geom, const int *leaf_rank,
                                                          const double *leaf_dx, GatherLeaf *out) {
    const int sl = blockIdx.x * 256 + threadIdx.x;
    if (sl >= slots) return;
    const int eid = slot_ids[2 * sl + 1];
    if (eid < 0) { out[sl] = dead_gather_leaf(); return; }      // the slot's canonical edge is not in the current list (edges.cpp)
    const EdgeGeom g = geom[eid];
    GatherLeaf gl;
    gl.dx_lo = leaf_dx[2 * (size_t)eid]; gl.dx_hi = leaf_dx[2 * (size_t)eid + 1];
#pragma unroll
    for (int k = 0; k < 3; ++k) { gl.v0[k] = g.v0[k]; gl.v1[k] = g.v1[k]; gl.o0[k] = g.o0[k]; gl.o1[k] = g.o1[k]; }
    gl.eid = eid; gl.rank = leaf_rank[eid];
    gl.f0 = g.f0 == -1 ? -1 : 0; gl.f1 = g.f1 == -1 ? -1 : 0;
    gl.has_normals = g.has_normals;
    out[sl] = gl;
}

inline dim3 grid_of(int n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

// Builds both hierarchies and everything derived from them on the calling thread's stream.  Inputs already on the device:
// d.geom (per-edge geometry), d.gather (the billboard hierarchy with its slot -> edge table).  ed.owned receives what the
// samplers (and the debug dump) read -- node arrays, the samplers' records, the gather's leaf records -- released with the Scene.  Returns after the one read-back the host needs: the depth of the deeper tree.
// The billboard hierarchy of the NEE-mode gather over the canonical edges' boxes: built by the kernels of bvh_gpu.cpp, or --
// same canonical edges as the last build (an optimisation loop moves vertices) -- a refit of a copy of that build's records;
// the leaf slots are then re-pointed to the CURRENT edge ids.
__global__ void __launch_bounds__(256) gather_ids_kernel(const int *canon_ids, const int *cur_of, int slots, int *out) {
    const int sl = blockIdx.x * 256 + threadIdx.x;
    if (sl < slots) { out[2 * sl] = 0; out[2 * sl + 1] = cur_of[canon_ids[2 * sl + 1]]; }
}
namespace {
struct GatherTreeCache { std::mutex lock; std::vector<EdgeD> canon; std::shared_ptr<rt::BvhDev> tree; int device = -1; };
// per device, like scene.cpp's edge / topology caches: two devices driven from one process do not evict each other's tree
GatherTreeCache *gather_tree_cache(int device) { static GatherTreeCache *c = new GatherTreeCache[16]; return c + ((device < 0 ? 0 : device) & 15); }
}
void drop_gather_cache() {
    for (int d = 0; d < 16; ++d) {
        GatherTreeCache *c = gather_tree_cache(d);
        std::lock_guard<std::mutex> lk(c->lock);        // (the edge-builder thread may be in gather_hierarchy_device)
        c->canon.clear(); c->tree.reset(); c->device = -1;
    }
}
void gather_hierarchy_device(EdgeData &ed) {
    GatherTreeCache *cache = gather_tree_cache(exec::current_device());      // one build at a time per device (scene.cpp: EdgeBuilder); rdr_trim_cache() may drop it
    std::lock_guard<std::mutex> cache_lock(cache->lock);
    hipStream_t s = exec::ctx().stream;
    const int nc = (int)ed.gather_cur_of.size();
    if (nc == 0) return;
    auto up = [&](const void *src, size_t bytes) -> void * {
        void *p = exec::pool_alloc(bytes);
        ed.owned.push_back(p);
        exec::upload_async(p, src, bytes);
        return p;
    };
    const float *boxes = (const float *)up(ed.gather_boxes.data(), sizeof(float) * ed.gather_boxes.size());
    const int *cur_of = (const int *)up(ed.gather_cur_of.data(), sizeof(int) * (size_t)nc);
    auto tree = std::make_shared<rt::BvhDev>();
    bool have = false;
    const int device = exec::current_device();
    if (ed.gather_refit_allowed && cache->tree && cache->device == device && cache->canon.size() == ed.gather_canon.size() &&
        std::memcmp(cache->canon.data(), ed.gather_canon.data(), sizeof(EdgeD) * ed.gather_canon.size()) == 0) {
        rt::refit_box_bvh_device(*cache->tree, boxes, *tree);
        tree->parent = cache->tree;
        double area = 0;
        exec::upload_flush();
        exec::download(&area, tree->area, sizeof(double));
        have = tree->inner_area > 0 && area / tree->inner_area <= 1.3;
        if (!have) tree = std::make_shared<rt::BvhDev>();
    }
    if (!have) {
        exec::upload_flush();                     // the boxes travel through the staging buffer; the build synchronises per level
        rt::build_box_bvh_device(boxes, nc, rt::BvhBuildParams{}, *tree);
        cache->canon = ed.gather_canon;
        cache->tree = tree;
        cache->device = device;
    }
    if (tree->depth + 2 > 64) throw std::runtime_error("edge gather hierarchy deeper than the traversal stack (64)");
    if ((size_t)tree->num_slots >= ((size_t)1 << 24)) throw std::runtime_error("edge gather hierarchy: more than 2^24 edges are not supported");
    int *ids = (int *)exec::pool_alloc(sizeof(int) * 2 * (size_t)tree->num_slots);
    ed.owned.push_back(ids);
    hipLaunchKernelGGL(gather_ids_kernel, grid_of(tree->num_slots), dim3(256), 0, s, (const int *)tree->ids, cur_of, tree->num_slots, ids);
    ed.gather_dev = tree;
    ed.d.gather = rt::BvhD{tree->nodes, nullptr, ids, tree->num_nodes, tree->num_slots, tree->depth + 2};
}

void build_edge_trees_device(EdgeData &ed) {
    hipStream_t s = exec::ctx().stream;
    auto alloc = [&](size_t bytes) -> void * { void *p = exec::pool_alloc(bytes ? bytes : 16); ed.owned.push_back(p); return p; };
    // what only the build reads (bounds, codes, sort scratch, counters, heights, level lists) goes back to the pool when the
    // build ends -- after the stream has drained, also when it ends by an exception
    struct Temporaries {
        hipStream_t s; std::vector<void *> blocks;
        ~Temporaries() { (void)hipStreamSynchronize(s); for (void *p : blocks) exec::pool_free(p); }
    } temporaries{s, {}};
    auto talloc = [&](size_t bytes) -> void * { void *p = exec::pool_alloc(bytes ? bytes : 16); temporaries.blocks.push_back(p); return p; };
    auto up = [&](const void *src, size_t bytes) -> void * { void *p = talloc(bytes); if (bytes) exec::upload_async(p, src, bytes); return p; };
    EdgeSceneD &d = ed.d;
    const int ne = (int)ed.edges.size();
    Box6D *bounds = (Box6D *)talloc(sizeof(Box6D) * (size_t)ne);
    const double *wlen = (const double *)up(ed.wlen.data(), sizeof(double) * (size_t)ne);
    int *leaf_rank = (int *)talloc(sizeof(int) * (size_t)ne);
    double *leaf_dx = (double *)talloc(sizeof(double) * 2 * (size_t)ne);
    // small integers the host reads back: per tree kMaxLevels + 1 level counts; then [0] depth of the 3-D tree, [1] of the
    // 6-D tree, [2] bounds not fp32
    constexpr int kLevelInts = kMaxLevels + 1;
    int *ints = (int *)talloc(sizeof(int) * (size_t)(4 * kLevelInts + 4));
    exec::zero(ints, sizeof(int) * (size_t)(4 * kLevelInts + 4));
    int *level_count[2] = {ints, ints + kLevelInts}, *cursor[2] = {ints + 2 * kLevelInts, ints + 3 * kLevelInts};
    int *flags = ints + 4 * kLevelInts;
    hipLaunchKernelGGL(edge_bounds_kernel, grid_of(ne), dim3(256), 0, s, d.geom, ne, d.cam_org, bounds);

    // ---- both trees up to their bounds (and the heights the treelet pass is scheduled by) ----
    const std::vector<int> *id_lists[2] = {&ed.cs_ids, &ed.ncs_ids};
    TreeD trees[2];
    for (int tree = 0; tree < 2; ++tree) {
        const std::vector<int> &ids_h = *id_lists[tree];
        const int n = (int)ids_h.size();
        TreeD &t = trees[tree];
        t = TreeD{};
        t.n = n; t.n_internal = n - 1 > 1 ? n - 1 : 1; t.is3d = tree == 0 ? 1 : 0;
        ed.dev_nodes[tree] = nullptr; ed.dev_n[tree] = n;
        if (n == 0) continue;
        const int total = t.n_internal + n;
        const int *ids_in = (const int *)up(ids_h.data(), sizeof(int) * (size_t)n);
        const int sb_blocks = std::max(1, std::min(128, n / 2048));
        Box6D *sb_part = (Box6D *)talloc(sizeof(Box6D) * (size_t)sb_blocks);
        Box6D *sb = (Box6D *)talloc(sizeof(Box6D));
        uint64_t *codes_in = (uint64_t *)talloc(sizeof(uint64_t) * (size_t)n), *codes = (uint64_t *)talloc(sizeof(uint64_t) * (size_t)n);
        int *ids = (int *)talloc(sizeof(int) * (size_t)n);
        hipLaunchKernelGGL(scene_bounds_kernel, dim3((unsigned)sb_blocks), dim3(256), 0, s, bounds, ids_in, n, sb_part);
        hipLaunchKernelGGL(scene_bounds_kernel, dim3(1), dim3(256), 0, s, (const Box6D *)sb_part, (const int *)nullptr, sb_blocks, sb);
        hipLaunchKernelGGL(codes_kernel, grid_of(n), dim3(256), 0, s, bounds, ids_in, n, sb, t.is3d, codes_in);
        {
            uint64_t *codes_tmp = (uint64_t *)talloc(sizeof(uint64_t) * (size_t)n);
            int *ids_tmp = (int *)talloc(sizeof(int) * (size_t)n);
            int *hist = (int *)talloc(sizeof(int) * 256 * (size_t)((n + kSortTile - 1) / kSortTile));
            radix_sort_pairs_u64(s, codes_tmp, ids_tmp, codes_in, ids_in, codes, ids, n, hist);
        }
        t.nodes = (EdgeNode *)alloc(sizeof(EdgeNode) * (size_t)total);
        t.below = (int *)talloc(sizeof(int) * (size_t)total);
        t.counter = (int *)talloc(sizeof(int) * (size_t)total);
        t.height = (int *)talloc(sizeof(int) * (size_t)total);
        t.level_count = level_count[tree];
        t.level_list = (int *)talloc(sizeof(int) * (size_t)t.n_internal);
        t.codes = codes; t.ids = ids;
        ed.dev_nodes[tree] = t.nodes;
        hipLaunchKernelGGL(init_nodes_kernel, grid_of(total), dim3(256), 0, s, t, bounds, wlen);
        if (n == 1) {
            hipLaunchKernelGGL(single_leaf_root_kernel, dim3(1), dim3(64), 0, s, t);
        } else {
            hipLaunchKernelGGL(radix_tree_kernel, grid_of(n - 1), dim3(256), 0, s, t);
            hipLaunchKernelGGL(bounds_up_kernel, grid_of(n), dim3(256), 0, s, t);
            hipLaunchKernelGGL(level_histogram_kernel, grid_of(n - 1), dim3(256), 0, s, t);
        }
    }
    exec::check(hipGetLastError(), "edge hierarchy kernels");

    // ---- treelet pass, level by level ----
    std::vector<int> h_levels((size_t)2 * kLevelInts, 0);
    {
        exec::DownloadItem item{h_levels.data(), ints, sizeof(int) * (size_t)(2 * kLevelInts)};
        exec::download_batch(&item, 1);                   // flushes the queued uploads, waits for the kernels so far
    }
    int top = 0;
    std::vector<int> h_offsets((size_t)2 * kMaxLevels, 0);
    for (int tree = 0; tree < 2; ++tree) {
        const int *cnt = h_levels.data() + (size_t)tree * kLevelInts;
        if (trees[tree].n > 1 && cnt[kMaxLevels] >= kMaxLevels - 1) throw std::runtime_error("edge hierarchy: radix tree higher than 126 levels");
        int run = 0;
        for (int h = 0; h < kMaxLevels; ++h) { h_offsets[(size_t)tree * kMaxLevels + h] = run; run += cnt[h]; if (cnt[h] > 0 && h > top) top = h; }
    }
    const int *d_offsets = (const int *)up(h_offsets.data(), sizeof(int) * h_offsets.size());
    for (int tree = 0; tree < 2; ++tree)
        if (trees[tree].n > 1)
            hipLaunchKernelGGL(level_scatter_kernel, grid_of(trees[tree].n - 1), dim3(256), 0, s, trees[tree], d_offsets + tree * kMaxLevels, cursor[tree]);
    for (int h = 1; h <= top; ++h) {
        // the 6-D tree's nodes first (the larger tree), the 3-D tree's after them
        const int c1 = trees[1].n > 1 ? h_levels[(size_t)kLevelInts + h] : 0, c0 = trees[0].n > 1 ? h_levels[(size_t)h] : 0;
        if (c0 + c1 == 0) continue;
        hipLaunchKernelGGL(treelet_level_kernel, dim3((unsigned)(c0 + c1)), dim3(64), 0, s, trees[1], h_offsets[(size_t)kMaxLevels + h], c1,
                           trees[0], h_offsets[(size_t)h]);
    }

    // ---- leaf order, depth, the samplers' records ----
    const EdgeNodeP *fat[2] = {nullptr, nullptr};
    int roots[2] = {kNoEdgeTree, kNoEdgeTree};
    for (int tree = 0; tree < 2; ++tree) {
        const TreeD &t = trees[tree];
        if (t.n == 0) continue;
        // the 6-D tree is walked first, so the 3-D tree's ranks start after its leaves
        const int first_rank = tree == 0 ? (int)ed.ncs_ids.size() : 0;
        hipLaunchKernelGGL(leaf_rank_kernel, grid_of(t.n), dim3(256), 0, s, t, first_rank, tree == 1 ? 1 : 0, leaf_rank, leaf_dx, flags + tree);
        if (t.n > 1) {
            EdgeNodeP *out = (EdgeNodeP *)alloc(sizeof(EdgeNodeP) * (size_t)(t.n - 1));
            hipLaunchKernelGGL(fatten_kernel, grid_of(t.n - 1), dim3(256), 0, s, t, out, flags + 2);
            fat[tree] = out;
            roots[tree] = tree == 0 ? 0 : kEdgeTreeBit;
        } else {
            roots[tree] = ~(*id_lists[tree])[0];                    // a single edge: the root reference is the leaf
        }
    }
    d.cs_nodes = fat[0]; d.ncs_nodes = fat[1];
    d.cs_root = roots[0]; d.ncs_root = roots[1];
    if (d.gather.num_tris > 0) {
        GatherLeaf *gl = (GatherLeaf *)alloc(sizeof(GatherLeaf) * (size_t)d.gather.num_tris);
        hipLaunchKernelGGL(gather_leaf_kernel, grid_of(d.gather.num_tris), dim3(256), 0, s, d.gather.ids, d.gather.num_tris, d.geom, leaf_rank, leaf_dx, gl);
        d.gleaf = gl;
    }
    exec::check(hipGetLastError(), "edge hierarchy kernels");
    int h_flags[4] = {0, 0, 0, 0};
    exec::DownloadItem item{h_flags, flags, sizeof(h_flags)};
    exec::download_batch(&item, 1);
    if (h_flags[2]) throw std::runtime_error("edge hierarchy: spatial bounds are not fp32 values");
    for (int k = 0; k < 2; ++k) {
        if (h_flags[k] == 0) continue;
        if (h_flags[k] + 2 > 64) throw std::runtime_error("edge hierarchy deeper than the traversal stack (64)");
        ed.max_stack = std::max(ed.max_stack, h_flags[k] + 2);
    }
    d.max_stack = ed.max_stack;
}

// Debug dump / tests: the device-built node arrays back on the host ([interior | leaves], as the host builder lays them out).
void download_edge_trees(EdgeData &ed) {
    for (int tree = 0; tree < 2; ++tree) {
        std::vector<EdgeNode> &dst = tree == 0 ? ed.cs_nodes : ed.ncs_nodes;
        const int n = ed.dev_n[tree];
        (tree == 0 ? ed.cs_leaves : ed.ncs_leaves) = n;
        dst.clear();
        if (n == 0 || !ed.dev_nodes[tree]) continue;
        const int n_internal = n - 1 > 1 ? n - 1 : 1;
        dst.resize((size_t)n_internal + n);
        exec::DownloadItem item{dst.data(), ed.dev_nodes[tree], sizeof(EdgeNode) * dst.size()};
        exec::download_batch(&item, 1);
    }
}

}  // namespace rdr
