// bvh_gpu.cpp -- the triangle hierarchy (and any hierarchy over boxes) built, refitted and widened on the GPU
// (row 8f-1 of SURVEY.md: "scene/edge build on GPU").
//
// The reference hands its meshes to Embree / OptiX Prime per Scene (src/scene.cpp:128-154); rounds 1-3 built the hierarchy
// of bvh.h on host threads (bvh.cpp) and uploaded it.  Here the same binned-SAH top-down build runs as kernels, level by
// level, one WAVE per node of the level:
//   decide   bounds and centroid bounds of the node's primitives (lanes stride over the range, xor-shuffle reductions); the
//            primitives binned per axis into LDS (atomic min / max on order-preserving integer images of the floats, atomic
//            counts); three lanes sweep the bins of an axis each and price the splits; the wave takes the cheapest
//   scan     one workgroup ranks the nodes that split: the children of the level get consecutive node records, in node
//            order -- breadth-first layout with siblings adjacent, which is what the ray kernels stage into LDS
//   apply    a stable partition of the node's slice of the primitive permutation (ballot + popcount), the two child ranges
//            become work items of the next level; a leaf's range is final: leaf slot = position in the permutation
// Every float expression and every tie rule is the host builder's (bvh.cpp: Builder::split_range; min / max are exact and
// order-free, the sweeps are sequential per axis like the host's), so the device tree EQUALS the host tree node for node --
// tests/test_raytri.py compares them (rdr_debug_bvh_check), and the CPU debugging harness, which has no kernels, keeps using
// bvh.cpp.  Then, still on the device: the 4-wide records (collapse_wide: greedy adoption by area, breadth first), the
// triangle records gathered in leaf order from the caller's vertex arrays, and -- for a Scene whose connectivity equals the
// previous Scene's -- the REFIT: new triangle records, boxes bottom-up level by level, the wide records' boxes, and the sum
// of the inner half-areas that tells the host when the topology has gone stale.  Nothing of the hierarchy crosses PCIe.
#include "bvh.h"
#include "bvh_gpu.h"
#include "hip/exec.h"

#include <algorithm>
#include <limits>
#include <stdexcept>

namespace rt {
namespace {

constexpr int kBinsMax = 64;
struct WorkItem { int node, first, count; };
struct Decision { int kind, axis, bin, mid; float lo, scale; };      // kind: 0 leaf, 1 SAH split, 2 median split
struct LevelBook { int nwork_next, node_count, wide_count, wide_need; };

__device__ inline unsigned enc(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ inline float dec(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
__device__ inline float half_area(const float lo[3], const float hi[3]) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0) return 0;
    return dx * dy + dy * dz + dz * dx;
}
__device__ inline float wave_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ inline float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }

// ---- primitive boxes of a triangle soup: prim p = (shape, triangle) -------------------------------------------------------
struct ShapeRef { const float *vertices; const int *indices; };
__global__ void __launch_bounds__(256) tri_boxes_kernel(const ShapeRef *shapes, const int *prim_ids, int n, float *boxes) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const ShapeRef sh = shapes[prim_ids[2 * p]];
    const int t = prim_ids[2 * p + 1];
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -lo[a]; }
    for (int k = 0; k < 3; ++k) {
        const int vi = sh.indices[3 * t + k];
        for (int a = 0; a < 3; ++a) { const float x = sh.vertices[3 * (size_t)vi + a]; lo[a] = fminf(lo[a], x); hi[a] = fmaxf(hi[a], x); }
    }
    for (int a = 0; a < 3; ++a) { boxes[6 * (size_t)p + a] = lo[a]; boxes[6 * (size_t)p + 3 + a] = hi[a]; }
}
// triangle records + ids in leaf order (slot -> prim via `perm`, or -- refit -- via the ids already there)
__global__ void __launch_bounds__(256) tri_gather_kernel(const ShapeRef *shapes, const int *prim_ids, const int *perm, int n, float *tris, int *ids) {
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= n) return;
    int s, t;
    if (perm) { const int p = perm[slot]; s = prim_ids[2 * p]; t = prim_ids[2 * p + 1]; ids[2 * slot] = s; ids[2 * slot + 1] = t; }
    else { s = ids[2 * slot]; t = ids[2 * slot + 1]; }
    const ShapeRef sh = shapes[s];
    for (int k = 0; k < 3; ++k) {
        const int vi = sh.indices[3 * t + k];
        for (int a = 0; a < 3; ++a) tris[9 * (size_t)slot + 3 * k + a] = sh.vertices[3 * (size_t)vi + a];
    }
}
__global__ void __launch_bounds__(256) box_ids_kernel(const int *perm, int n, int *ids) {
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot < n) { ids[2 * slot] = 0; ids[2 * slot + 1] = perm[slot]; }
}

// ---- build: decide ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) decide_kernel(const WorkItem *work, int nwork, const int *perm, const float *boxes, Node *nodes,
                                                     Decision *decisions, int leaf_max, int bins, float trav_cost) {
    // (one wave per workgroup: __syncthreads() is the wave's own LDS fence)
    __shared__ unsigned s_lo[1][3][kBinsMax][3], s_hi[1][3][kBinsMax][3];
    __shared__ int s_cnt[1][3][kBinsMax];
    __shared__ float s_ra[1][3][kBinsMax];
    __shared__ int s_rc[1][3][kBinsMax];
    const int wave = 0, lane = threadIdx.x;
    const int item = blockIdx.x;
    if (item >= nwork) return;
    const WorkItem w = work[item];
    const float inf = std::numeric_limits<float>::infinity();
    float blo[3] = {inf, inf, inf}, bhi[3] = {-inf, -inf, -inf}, clo[3] = {inf, inf, inf}, chi[3] = {-inf, -inf, -inf};
    for (int j = lane; j < w.count; j += 64) {
        const float *b = boxes + 6 * (size_t)perm[w.first + j];
        for (int a = 0; a < 3; ++a) {
            const float l = b[a], h = b[3 + a], c = 0.5f * (l + h);
            blo[a] = fminf(blo[a], l); bhi[a] = fmaxf(bhi[a], h);
            clo[a] = fminf(clo[a], c); chi[a] = fmaxf(chi[a], c);
        }
    }
    for (int a = 0; a < 3; ++a) { blo[a] = wave_min(blo[a]); bhi[a] = wave_max(bhi[a]); clo[a] = wave_min(clo[a]); chi[a] = wave_max(chi[a]); }
    int best_axis = -1, best_bin = -1;
    float best_cost = inf;
    if (w.count > 1) {
        for (int e = lane; e < 3 * kBinsMax; e += 64) {
            const int a = e / kBinsMax, k = e % kBinsMax;
            for (int c = 0; c < 3; ++c) { s_lo[wave][a][k][c] = enc(inf); s_hi[wave][a][k][c] = enc(-inf); }
            s_cnt[wave][a][k] = 0;
        }
        __syncthreads();
        for (int a = 0; a < 3; ++a) {
            const float ext = chi[a] - clo[a];
            if (!(ext > 0)) continue;
            const float scale = bins / ext;
            for (int j = lane; j < w.count; j += 64) {
                const float *b = boxes + 6 * (size_t)perm[w.first + j];
                const float c = 0.5f * (b[a] + b[3 + a]);
                const int bi = min(bins - 1, max(0, (int)((c - clo[a]) * scale)));
                for (int k = 0; k < 3; ++k) { atomicMin(&s_lo[wave][a][bi][k], enc(b[k])); atomicMax(&s_hi[wave][a][bi][k], enc(b[3 + k])); }
                atomicAdd(&s_cnt[wave][a][bi], 1);
            }
        }
        __syncthreads();
        // one lane per axis: the host's two sweeps (suffix areas / counts, then prefixes and the split costs)
        float my_cost = inf; int my_bin = -1;
        if (lane < 3 && (chi[lane] - clo[lane]) > 0) {
            const int a = lane;
            float alo[3] = {inf, inf, inf}, ahi[3] = {-inf, -inf, -inf};
            int c = 0;
            for (int k = bins - 1; k > 0; --k) {
                for (int x = 0; x < 3; ++x) { alo[x] = fminf(alo[x], dec(s_lo[wave][a][k][x])); ahi[x] = fmaxf(ahi[x], dec(s_hi[wave][a][k][x])); }
                c += s_cnt[wave][a][k];
                s_ra[wave][a][k] = half_area(alo, ahi); s_rc[wave][a][k] = c;
            }
            float llo[3] = {inf, inf, inf}, lhi[3] = {-inf, -inf, -inf};
            int lc = 0;
            for (int k = 0; k < bins - 1; ++k) {
                for (int x = 0; x < 3; ++x) { llo[x] = fminf(llo[x], dec(s_lo[wave][a][k][x])); lhi[x] = fmaxf(lhi[x], dec(s_hi[wave][a][k][x])); }
                lc += s_cnt[wave][a][k];
                if (lc == 0 || s_rc[wave][a][k + 1] == 0) continue;
                const float cost = half_area(llo, lhi) * lc + s_ra[wave][a][k + 1] * s_rc[wave][a][k + 1];
                if (cost < my_cost) { my_cost = cost; my_bin = k; }
            }
        }
        for (int a = 0; a < 3; ++a) {                         // axis order, strictly cheaper wins: the host's loop
            const float ca = __shfl(my_cost, a); const int ba = __shfl(my_bin, a);
            if (ba >= 0 && ca < best_cost) { best_cost = ca; best_axis = a; best_bin = ba; }
        }
    }
    if (lane != 0) return;
    Node &n = nodes[w.node];
    for (int k = 0; k < 3; ++k) { n.lo[k] = blo[k]; n.hi[k] = bhi[k]; }
    pad_box(n.lo, n.hi);
    const float area = half_area(blo, bhi);
    const float leaf_cost = area * w.count;
    bool split = best_axis >= 0 && (w.count > leaf_max || best_cost + trav_cost * area < leaf_cost);
    Decision d{0, 0, 0, w.first + w.count / 2, 0.f, 0.f};
    if (split) {
        int left = 0;
        for (int k = 0; k <= best_bin; ++k) left += s_cnt[wave][best_axis][k];
        if (left == 0 || left == w.count) split = false;
        else { d.kind = 1; d.axis = best_axis; d.bin = best_bin; d.mid = w.first + left; d.lo = clo[best_axis]; d.scale = bins / (chi[best_axis] - clo[best_axis]); }
    }
    if (!split && w.count > leaf_max) { d.kind = 2; d.mid = w.first + w.count / 2; }      // coincident centroids: median by index
    decisions[item] = d;
}

// ---- build: rank the splitting nodes of the level (one workgroup) ----------------------------------------------------------
__global__ void __launch_bounds__(256) rank_kernel(const Decision *decisions, int nwork, int *rank, LevelBook *book) {
    __shared__ int part[256];
    const int per = (nwork + 255) / 256;
    const int beg = threadIdx.x * per, end = min(beg + per, nwork);
    int s = 0;
    for (int i = beg; i < end; ++i) s += decisions[i].kind != 0;
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) { const int t = part[i]; part[i] = run; run += t; }
        book->nwork_next = 2 * run;
        // (node_count is advanced by apply_kernel's host side: the children of this level start at the current count)
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = beg; i < end; ++i) { rank[i] = run; run += decisions[i].kind != 0; }
}

// ---- build: partition, emit the children ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) apply_kernel(const WorkItem *work, int nwork, const Decision *decisions, const int *rank, int child_base,
                                                    const int *perm_in, int *perm_out, int *perm_final, const float *boxes, Node *nodes,
                                                    WorkItem *work_next, int bins) {
    const int lane = threadIdx.x;
    const int item = blockIdx.x;
    if (item >= nwork) return;
    const WorkItem w = work[item];
    const Decision d = decisions[item];
    if (d.kind == 0) {
        for (int j = lane; j < w.count; j += 64) perm_final[w.first + j] = perm_in[w.first + j];
        if (lane == 0) { nodes[w.node].a = w.first; nodes[w.node].b = w.count; }
        return;
    }
    const int cb = child_base + 2 * rank[item];
    int nl = 0, nr = 0;
    for (int j0 = 0; j0 < w.count; j0 += 64) {
        const int j = j0 + lane;
        bool valid = j < w.count, left = false;
        int p = 0;
        if (valid) {
            p = perm_in[w.first + j];
            if (d.kind == 1) {
                const float *b = boxes + 6 * (size_t)p;
                const float c = 0.5f * (b[d.axis] + b[3 + d.axis]);
                left = min(bins - 1, max(0, (int)((c - d.lo) * d.scale))) <= d.bin;
            } else left = w.first + j < d.mid;
        }
        const unsigned long long ml = __ballot(valid && left), mr = __ballot(valid && !left);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (valid) {
            if (left) perm_out[w.first + nl + __popcll(ml & below)] = p;
            else perm_out[d.mid + nr + __popcll(mr & below)] = p;
        }
        nl += __popcll(ml); nr += __popcll(mr);
    }
    if (lane == 0) {
        nodes[w.node].a = cb; nodes[w.node].b = 0;
        work_next[2 * rank[item]] = WorkItem{cb, w.first, d.mid - w.first};
        work_next[2 * rank[item] + 1] = WorkItem{cb + 1, d.mid, w.first + w.count - d.mid};
    }
}

// ---- the 4-wide records (bvh.cpp: collapse_wide), one breadth-first level per launch triple -------------------------------
struct WideItem { int binary, wide, pending; };
struct WideKids { int kid[4]; int nk, interior; };
__device__ inline float node_area(const Node &n) { const float dx = n.hi[0] - n.lo[0], dy = n.hi[1] - n.lo[1], dz = n.hi[2] - n.lo[2]; return dx * dy + dy * dz + dz * dx; }
__global__ void __launch_bounds__(256) wide_pick_kernel(const WideItem *items, int n, const Node *nodes, WideKids *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const WideItem it = items[i];
    WideKids k;
    k.nk = 0;
    if (nodes[it.binary].b > 0) k.kid[k.nk++] = it.binary;
    else { k.kid[k.nk++] = nodes[it.binary].a; k.kid[k.nk++] = nodes[it.binary].a + 1; }
    while (k.nk < 4) {
        int pick = -1; float best = -1.f;
        for (int j = 0; j < k.nk; ++j) if (nodes[k.kid[j]].b == 0 && node_area(nodes[k.kid[j]]) > best) { best = node_area(nodes[k.kid[j]]); pick = j; }
        if (pick < 0) break;
        const int a = nodes[k.kid[pick]].a;
        k.kid[pick] = a; k.kid[k.nk++] = a + 1;
    }
    for (int j = k.nk; j < 4; ++j) k.kid[j] = -1;
    k.interior = 0;
    for (int j = 0; j < k.nk; ++j) k.interior += nodes[k.kid[j]].b == 0;
    out[i] = k;
}
__global__ void __launch_bounds__(256) wide_rank_kernel(const WideKids *kids, int n, int *rank, LevelBook *book) {
    __shared__ int part[256];
    const int per = (n + 255) / 256;
    const int beg = threadIdx.x * per, end = min(beg + per, n);
    int s = 0;
    for (int i = beg; i < end; ++i) s += kids[i].interior;
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) { const int t = part[i]; part[i] = run; run += t; }
        book->nwork_next = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = beg; i < end; ++i) { rank[i] = run; run += kids[i].interior; }
}
__global__ void __launch_bounds__(256) wide_emit_kernel(const WideItem *items, int n, const WideKids *kids, const int *rank, int wide_base,
                                                        const Node *nodes, Node4 *wide, int *wide_src, WideItem *next, LevelBook *book) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const WideItem it = items[i];
    const WideKids k = kids[i];
    Node4 w;
    const float inf = std::numeric_limits<float>::infinity();
    for (int j = 0; j < 4; ++j) {
        w.lox[j] = w.loy[j] = w.loz[j] = inf; w.hix[j] = w.hiy[j] = w.hiz[j] = -inf;
        w.link[j] = kEmptyLink; w.aux[j] = 0;
    }
    w.aux[0] = k.nk;
    const int below = it.pending + k.nk - 1;
    atomicMax(&book->wide_need, below + 1);
    int order = 0;
    for (int j = 0; j < k.nk; ++j) {
        const Node c = nodes[k.kid[j]];
        w.lox[j] = c.lo[0]; w.loy[j] = c.lo[1]; w.loz[j] = c.lo[2];
        w.hix[j] = c.hi[0]; w.hiy[j] = c.hi[1]; w.hiz[j] = c.hi[2];
        if (c.b > 0) w.link[j] = leaf_link(c.a, c.b);
        else {
            const int at = rank[i] + order++;
            w.link[j] = wide_base + at;
            next[at] = WideItem{k.kid[j], wide_base + at, below};
        }
    }
    wide[it.wide] = w;
    for (int j = 0; j < 4; ++j) wide_src[4 * it.wide + j] = k.kid[j];
}

// ---- refit ------------------------------------------------------------------------------------------------------------------
// nodes [first, end) of one level: a leaf from its triangle records (or boxes), an inner node from its (already refitted) children
__global__ void __launch_bounds__(256) refit_level_kernel(Node *nodes, int first, int end, const float *tris, const float *boxes, const int *ids) {
    const int i = first + blockIdx.x * 256 + threadIdx.x;
    if (i >= end) return;
    Node n = nodes[i];
    const float inf = std::numeric_limits<float>::infinity();
    if (n.b > 0) {
        float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
        for (int k = 0; k < n.b; ++k) {
            if (tris) {
                const float *v = tris + 9 * (size_t)(n.a + k);
                for (int a = 0; a < 3; ++a) {
                    lo[a] = fminf(lo[a], fminf(v[a], fminf(v[3 + a], v[6 + a])));
                    hi[a] = fmaxf(hi[a], fmaxf(v[a], fmaxf(v[3 + a], v[6 + a])));
                }
            } else {
                const float *b = boxes + 6 * (size_t)ids[2 * (size_t)(n.a + k) + 1];
                for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], b[a]); hi[a] = fmaxf(hi[a], b[3 + a]); }
            }
        }
        for (int a = 0; a < 3; ++a) { n.lo[a] = lo[a]; n.hi[a] = hi[a]; }
        pad_box(n.lo, n.hi);
    } else {
        const Node l = nodes[n.a], r = nodes[n.a + 1];
        for (int a = 0; a < 3; ++a) { n.lo[a] = fminf(l.lo[a], r.lo[a]); n.hi[a] = fmaxf(l.hi[a], r.hi[a]); }
    }
    nodes[i] = n;
}
__global__ void __launch_bounds__(256) wide_refit_kernel(const Node *nodes, const int *wide_src, int nwide, Node4 *wide) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nwide) return;
    Node4 w = wide[i];
    for (int j = 0; j < 4; ++j) {
        const int b = wide_src[4 * i + j];
        if (b < 0) continue;
        const Node c = nodes[b];
        w.lox[j] = c.lo[0]; w.loy[j] = c.lo[1]; w.loz[j] = c.lo[2];
        w.hix[j] = c.hi[0]; w.hiy[j] = c.hi[1]; w.hiz[j] = c.hi[2];
    }
    wide[i] = w;
}
// sum of the inner nodes' half-areas (fp64, fixed reduction tree: the same number from run to run)
__global__ void __launch_bounds__(256) inner_area_kernel(const Node *nodes, int n, double *out) {
    __shared__ double part[256];
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const Node nd = nodes[i];
        if (nd.b > 0) continue;
        const double dx = (double)nd.hi[0] - nd.lo[0], dy = (double)nd.hi[1] - nd.lo[1], dz = (double)nd.hi[2] - nd.lo[2];
        s += dx * dy + dy * dz + dz * dx;
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) *out = part[0];
}

inline dim3 grid_of(int n) { return dim3((unsigned)((n + 255) / 256)); }
template <class T> T *take(BvhDev &d, size_t count) {
    T *p = (T *)exec::pool_alloc(sizeof(T) * (count ? count : 1));
    d.owned.push_back(p);
    return p;
}
struct Temp {             // scratch of one build, back to the pool when it ends (the build ends with a synchronisation)
    std::vector<void *> blocks;
    template <class T> T *get(size_t count) { T *p = (T *)exec::pool_alloc(sizeof(T) * (count ? count : 1)); blocks.push_back(p); return p; }
    ~Temp() { for (void *p : blocks) exec::pool_free(p); }
};

// the level loop over prim boxes: fills d.nodes / d.level_first / d.depth and returns the final permutation (slot -> prim)
int *build_levels(const float *boxes, int n, const BvhBuildParams &prm, BvhDev &d, Temp &tmp) {
    hipStream_t st = exec::ctx().stream;
    d.nodes = take<Node>(d, (size_t)2 * n);
    int *perm[2] = {tmp.get<int>(n), tmp.get<int>(n)};
    int *perm_final = tmp.get<int>(n);
    WorkItem *work[2] = {tmp.get<WorkItem>(n), tmp.get<WorkItem>(n)};
    Decision *decisions = tmp.get<Decision>(n);
    int *rank = tmp.get<int>(n);
    LevelBook *book = tmp.get<LevelBook>(1);
    {
        std::vector<int> iota((size_t)n);
        for (int i = 0; i < n; ++i) iota[i] = i;
        exec::upload_async(perm[0], iota.data(), sizeof(int) * n);
        const WorkItem root{0, 0, n};
        exec::upload_async(work[0], &root, sizeof(root));
    }
    int nwork = 1, node_count = 1, level = 0;
    d.level_first.assign(1, 0);
    while (nwork > 0) {
        if (level + 2 > kTraverseStack) throw std::runtime_error("triangle hierarchy deeper than the traversal stack");
        const int cur = level & 1;
        hipLaunchKernelGGL(decide_kernel, dim3((unsigned)nwork), dim3(64), 0, st, work[cur], nwork, perm[cur], boxes, d.nodes, decisions,
                           prm.leaf_max, prm.bins, prm.trav_cost);
        hipLaunchKernelGGL(rank_kernel, dim3(1), dim3(256), 0, st, decisions, nwork, rank, book);
        hipLaunchKernelGGL(apply_kernel, dim3((unsigned)nwork), dim3(64), 0, st, work[cur], nwork, decisions, rank, node_count, perm[cur], perm[cur ^ 1],
                           perm_final, boxes, d.nodes, work[cur ^ 1], prm.bins);
        exec::check(hipGetLastError(), "bvh build launch");
        LevelBook h;
        exec::download(&h, book, sizeof(h));                 // how many nodes the next level has: sizes its launches
        d.level_first.push_back(node_count);
        node_count += h.nwork_next;
        nwork = h.nwork_next;
        d.depth = level;
        ++level;
    }
    d.num_nodes = node_count;
    d.level_first.back() = node_count;       // (the last pushed entry is where a level that does not exist would start)
    return perm_final;
}

void widen(BvhDev &d, Temp &tmp) {
    hipStream_t st = exec::ctx().stream;
    const int cap = d.num_nodes;
    d.wide = take<Node4>(d, (size_t)cap);
    d.wide_src = take<int>(d, (size_t)4 * cap);
    WideItem *items[2] = {tmp.get<WideItem>(cap), tmp.get<WideItem>(cap)};
    WideKids *kids = tmp.get<WideKids>(cap);
    int *rank = tmp.get<int>(cap);
    LevelBook *book = tmp.get<LevelBook>(1);
    const LevelBook zero{0, 0, 0, 0};
    exec::upload_async(book, &zero, sizeof(zero));
    const WideItem root{0, 0, 0};
    exec::upload_async(items[0], &root, sizeof(root));
    int n = 1, wide_count = 1, level = 0;
    while (n > 0) {
        const int cur = level & 1;
        hipLaunchKernelGGL(wide_pick_kernel, grid_of(n), dim3(256), 0, st, items[cur], n, d.nodes, kids);
        hipLaunchKernelGGL(wide_rank_kernel, dim3(1), dim3(256), 0, st, kids, n, rank, book);
        hipLaunchKernelGGL(wide_emit_kernel, grid_of(n), dim3(256), 0, st, items[cur], n, kids, rank, wide_count, d.nodes, d.wide, d.wide_src,
                           items[cur ^ 1], book);
        exec::check(hipGetLastError(), "bvh widen launch");
        LevelBook h;
        exec::download(&h, book, sizeof(h));
        wide_count += h.nwork_next;
        n = h.nwork_next;
        d.wide_stack_need = h.wide_need;
        ++level;
    }
    d.num_wide = wide_count;
}

} // namespace

void build_tri_bvh_device(const void *d_shapes, const int *h_prim_ids, int n, const BvhBuildParams &prm, BvhDev &d) {
    if (n <= 0) return;
    hipStream_t st = exec::ctx().stream;
    Temp tmp;
    d.prim_ids = take<int>(d, (size_t)2 * n);
    exec::upload_async(d.prim_ids, h_prim_ids, sizeof(int) * 2 * n);
    d.shapes = d_shapes;
    float *boxes = tmp.get<float>((size_t)6 * n);
    hipLaunchKernelGGL(tri_boxes_kernel, grid_of(n), dim3(256), 0, st, (const ShapeRef *)d_shapes, d.prim_ids, n, boxes);
    int *perm = build_levels(boxes, n, prm, d, tmp);
    d.num_slots = n;
    d.tris = take<float>(d, (size_t)9 * n);
    d.ids = take<int>(d, (size_t)2 * n);
    hipLaunchKernelGGL(tri_gather_kernel, grid_of(n), dim3(256), 0, st, (const ShapeRef *)d_shapes, d.prim_ids, perm, n, d.tris, d.ids);
    widen(d, tmp);
    d.area = take<double>(d, 1);
    hipLaunchKernelGGL(inner_area_kernel, dim3(1), dim3(256), 0, st, d.nodes, d.num_nodes, d.area);
    exec::check(hipGetLastError(), "bvh build launch");
    exec::download(&d.inner_area, d.area, sizeof(double));          // also: the scratch may go back to the pool now
}

void build_box_bvh_device(const float *d_boxes, int n, const BvhBuildParams &prm, BvhDev &d) {
    if (n <= 0) return;
    hipStream_t st = exec::ctx().stream;
    Temp tmp;
    int *perm = build_levels(d_boxes, n, prm, d, tmp);
    d.num_slots = n;
    d.ids = take<int>(d, (size_t)2 * n);
    hipLaunchKernelGGL(box_ids_kernel, grid_of(n), dim3(256), 0, st, perm, n, d.ids);
    d.area = take<double>(d, 1);
    hipLaunchKernelGGL(inner_area_kernel, dim3(1), dim3(256), 0, st, d.nodes, d.num_nodes, d.area);
    exec::check(hipGetLastError(), "bvh build launch");
    exec::download(&d.inner_area, d.area, sizeof(double));
}

// A Scene with the connectivity of `src`'s: its own copies of the records that hold positions (nodes, triangle records, wide
// records), refitted from `d_shapes`; what holds connectivity only (ids, the wide records' sources, the level table) is shared.
// Stream-ordered, nothing comes back: `*area_ratio_dev` (device) receives inner area now / inner area at build time.
void refit_tri_bvh_device(const BvhDev &src, const void *d_shapes, BvhDev &d) {        // (the caller keeps `src` alive: BvhDev::parent)
    hipStream_t st = exec::ctx().stream;
    d.num_nodes = src.num_nodes; d.num_slots = src.num_slots; d.depth = src.depth; d.num_wide = src.num_wide;
    d.wide_stack_need = src.wide_stack_need; d.level_first = src.level_first; d.inner_area = src.inner_area;
    d.ids = src.ids; d.wide_src = src.wide_src; d.prim_ids = src.prim_ids; d.shapes = d_shapes;
    d.nodes = take<Node>(d, (size_t)src.num_nodes);
    d.tris = take<float>(d, (size_t)9 * src.num_slots);
    d.wide = take<Node4>(d, (size_t)src.num_wide);
    d.area = take<double>(d, 1);
    exec::copy_dev(d.nodes, src.nodes, sizeof(Node) * src.num_nodes);          // links + leaf ranges (the boxes are overwritten)
    exec::copy_dev(d.wide, src.wide, sizeof(Node4) * src.num_wide);
    if (d.num_slots > 0)             // (a zero-size grid is a launch error)
        hipLaunchKernelGGL(tri_gather_kernel, grid_of(d.num_slots), dim3(256), 0, st, (const ShapeRef *)d_shapes, (const int *)nullptr, (const int *)nullptr,
                           d.num_slots, d.tris, d.ids);
    for (int l = (int)d.level_first.size() - 2; l >= 0; --l) {
        const int first = d.level_first[l], end = d.level_first[l + 1];
        if (end > first) hipLaunchKernelGGL(refit_level_kernel, grid_of(end - first), dim3(256), 0, st, d.nodes, first, end, d.tris, (const float *)nullptr, (const int *)nullptr);
    }
    if (d.num_wide > 0) hipLaunchKernelGGL(wide_refit_kernel, grid_of(d.num_wide), dim3(256), 0, st, d.nodes, d.wide_src, d.num_wide, d.wide);
    hipLaunchKernelGGL(inner_area_kernel, dim3(1), dim3(256), 0, st, d.nodes, d.num_nodes, d.area);
    exec::check(hipGetLastError(), "bvh refit launch");
}

void refit_box_bvh_device(const BvhDev &src, const float *d_boxes, BvhDev &d) {
    hipStream_t st = exec::ctx().stream;
    d.num_nodes = src.num_nodes; d.num_slots = src.num_slots; d.depth = src.depth;
    d.level_first = src.level_first; d.inner_area = src.inner_area;
    d.ids = src.ids;
    d.nodes = take<Node>(d, (size_t)src.num_nodes);
    d.area = take<double>(d, 1);
    exec::copy_dev(d.nodes, src.nodes, sizeof(Node) * src.num_nodes);
    for (int l = (int)d.level_first.size() - 2; l >= 0; --l) {
        const int first = d.level_first[l], end = d.level_first[l + 1];
        if (end > first) hipLaunchKernelGGL(refit_level_kernel, grid_of(end - first), dim3(256), 0, st, d.nodes, first, end, (const float *)nullptr, d_boxes, (const int *)d.ids);
    }
    hipLaunchKernelGGL(inner_area_kernel, dim3(1), dim3(256), 0, st, d.nodes, d.num_nodes, d.area);
    exec::check(hipGetLastError(), "bvh refit launch");
}

BvhDev::~BvhDev() { for (void *p : owned) exec::pool_free(p); }

} // namespace rt
