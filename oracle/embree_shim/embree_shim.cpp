// Embree stand-in used ONLY by the parity oracle (oracle/_ref).  See embree3/rtcore.h.
//
// A plain median-split binary BVH over all triangles of all attached geometries, queried one ray
// at a time, nearer child first (round 6: -11 % on the oracle's bunny_box render; the walk's order
// never decides a hit).  The hit rule is the shared predicate in redner_amd/csrc/raytri.h, so the
// result is, by construction, the same as a brute-force scan with that predicate
// (tests/test_raytri.py checks this against brute force).  Scalar code: real Embree (SIMD boxes,
// SAH, packets) would be faster still -- bench.py's cpu_baseline says which stand-in it timed.
#include "embree3/rtcore.h"
#include "../../redner_amd/csrc/raytri.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

struct RTCDeviceTy { int refs; };

struct RTCGeometryTy {
    int refs = 1;
    std::vector<unsigned char> vbuf, ibuf;
    size_t vstride = 0, istride = 0, nverts = 0, ntris = 0;
};

namespace {
struct Tri { float a[3], b[3], c[3]; int geom, prim; };
struct Node { float lo[3], hi[3]; int left, right, first, count; };
}

struct RTCSceneTy {
    std::vector<RTCGeometryTy *> geoms;
    std::vector<Tri> tris;
    std::vector<Node> nodes;
};

namespace {

void tri_bounds(const Tri &t, float lo[3], float hi[3]) {
    for (int k = 0; k < 3; ++k) {
        lo[k] = std::min(t.a[k], std::min(t.b[k], t.c[k]));
        hi[k] = std::max(t.a[k], std::max(t.b[k], t.c[k]));
    }
}

int build(RTCSceneTy *s, int first, int count) {
    Node n;
    for (int k = 0; k < 3; ++k) { n.lo[k] = std::numeric_limits<float>::infinity(); n.hi[k] = -n.lo[k]; }
    float clo[3], chi[3];
    for (int k = 0; k < 3; ++k) { clo[k] = n.lo[k]; chi[k] = n.hi[k]; }
    for (int i = first; i < first + count; ++i) {
        float lo[3], hi[3];
        tri_bounds(s->tris[i], lo, hi);
        for (int k = 0; k < 3; ++k) {
            n.lo[k] = std::min(n.lo[k], lo[k]); n.hi[k] = std::max(n.hi[k], hi[k]);
            float c = 0.5f * (lo[k] + hi[k]);
            clo[k] = std::min(clo[k], c); chi[k] = std::max(chi[k], c);
        }
    }
    rt::pad_box(n.lo, n.hi);
    n.left = n.right = -1; n.first = first; n.count = count;
    int id = (int)s->nodes.size();
    s->nodes.push_back(n);
    if (count > 4) {
        int axis = 0;
        if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
        if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
        int mid = first + count / 2;
        std::nth_element(s->tris.begin() + first, s->tris.begin() + mid, s->tris.begin() + first + count,
            [axis](const Tri &x, const Tri &y) {
                float cx = x.a[axis] + x.b[axis] + x.c[axis];
                float cy = y.a[axis] + y.b[axis] + y.c[axis];
                if (cx != cy) return cx < cy;
                if (x.geom != y.geom) return x.geom < y.geom;
                return x.prim < y.prim;
            });
        int l = build(s, first, mid - first);
        int r = build(s, mid, first + count - mid);
        s->nodes[id].left = l; s->nodes[id].right = r; s->nodes[id].count = 0;
    }
    return id;
}

template <bool ANY>
rt::Hit query(const RTCSceneTy *s, const float o[3], const float d[3], float tnear, float tfar) {
    rt::Hit best{tfar, -1, -1};
    if (s->nodes.empty()) return best;
    float inv[3] = {1.f / d[0], 1.f / d[1], 1.f / d[2]};
    int stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp > 0) {
        const Node &n = s->nodes[stack[--sp]];
        float tn;
        // '<=' on ties: keep the search window closed at best.t so equal-t candidates are seen
        float far_lim = best.shape < 0 ? tfar : best.t * 1.0000004f + 1e-30f;
        if (!rt::ray_box(o, inv, tnear, far_lim, n.lo, n.hi, &tn)) continue;
        if (n.left < 0) {
            for (int i = n.first; i < n.first + n.count; ++i) {
                const Tri &t = s->tris[i];
                float th;
                if (rt::ray_triangle(o, d, tnear, tfar, t.a, t.b, t.c, &th)) {
                    if (ANY) return rt::Hit{th, t.geom, t.prim};
                    if (rt::closer(th, t.geom, t.prim, best)) best = rt::Hit{th, t.geom, t.prim};
                }
            }
        } else {
            // nearer child first (entry distances from the same slab test): the search window closes sooner.  Which child is
            // visited first never decides a hit -- closer() is a total order over (t, shape, triangle) -- only how many boxes the
            // walk opens.
            const Node &a = s->nodes[n.left], &b = s->nodes[n.right];
            float ta, tb;
            const bool ha = rt::ray_box(o, inv, tnear, far_lim, a.lo, a.hi, &ta);
            const bool hb = rt::ray_box(o, inv, tnear, far_lim, b.lo, b.hi, &tb);
            if (ha && hb) {
                if (tb < ta) { stack[sp++] = n.left; stack[sp++] = n.right; }
                else { stack[sp++] = n.right; stack[sp++] = n.left; }
            } else if (ha) stack[sp++] = n.left;
            else if (hb) stack[sp++] = n.right;
        }
    }
    return best;
}

} // namespace

extern "C" {

RTCDevice rtcNewDevice(const char *) { return new RTCDeviceTy{1}; }
void rtcReleaseDevice(RTCDevice d) { delete d; }
RTCScene rtcNewScene(RTCDevice) { return new RTCSceneTy(); }
void rtcReleaseScene(RTCScene s) {
    if (!s) return;
    for (auto *g : s->geoms) if (--g->refs == 0) delete g;
    delete s;
}
void rtcSetSceneBuildQuality(RTCScene, enum RTCBuildQuality) {}
void rtcSetSceneFlags(RTCScene, enum RTCSceneFlags) {}
RTCGeometry rtcNewGeometry(RTCDevice, enum RTCGeometryType) { return new RTCGeometryTy(); }
void *rtcSetNewGeometryBuffer(RTCGeometry g, enum RTCBufferType type, unsigned int, enum RTCFormat,
                              size_t stride, size_t count) {
    if (type == RTC_BUFFER_TYPE_VERTEX) {
        g->vstride = stride; g->nverts = count; g->vbuf.assign(stride * count + 16, 0);
        return g->vbuf.data();
    }
    g->istride = stride; g->ntris = count; g->ibuf.assign(stride * count + 16, 0);
    return g->ibuf.data();
}
void rtcSetGeometryVertexAttributeCount(RTCGeometry, unsigned int) {}
void rtcCommitGeometry(RTCGeometry) {}
unsigned int rtcAttachGeometry(RTCScene s, RTCGeometry g) {
    g->refs++;
    s->geoms.push_back(g);
    return (unsigned int)s->geoms.size() - 1;
}
void rtcReleaseGeometry(RTCGeometry g) { if (--g->refs == 0) delete g; }

void rtcCommitScene(RTCScene s) {
    s->tris.clear(); s->nodes.clear();
    for (size_t gi = 0; gi < s->geoms.size(); ++gi) {
        const RTCGeometryTy *g = s->geoms[gi];
        for (size_t ti = 0; ti < g->ntris; ++ti) {
            const unsigned int *ix = (const unsigned int *)(g->ibuf.data() + ti * g->istride);
            Tri t; t.geom = (int)gi; t.prim = (int)ti;
            std::memcpy(t.a, g->vbuf.data() + ix[0] * g->vstride, 12);
            std::memcpy(t.b, g->vbuf.data() + ix[1] * g->vstride, 12);
            std::memcpy(t.c, g->vbuf.data() + ix[2] * g->vstride, 12);
            s->tris.push_back(t);
        }
    }
    if (!s->tris.empty()) build(s, 0, (int)s->tris.size());
}

void rtcIntersect1(RTCScene s, struct RTCIntersectContext *, struct RTCRayHit *rh) {
    float o[3] = {rh->ray.org_x, rh->ray.org_y, rh->ray.org_z};
    float d[3] = {rh->ray.dir_x, rh->ray.dir_y, rh->ray.dir_z};
    rt::Hit h = query<false>(s, o, d, rh->ray.tnear, rh->ray.tfar);
    if (h.shape >= 0) {
        rh->ray.tfar = h.t;
        rh->hit.geomID = (unsigned int)h.shape;
        rh->hit.primID = (unsigned int)h.prim;
    }
}

void rtcOccluded1(RTCScene s, struct RTCIntersectContext *, struct RTCRay *r) {
    float o[3] = {r->org_x, r->org_y, r->org_z};
    float d[3] = {r->dir_x, r->dir_y, r->dir_z};
    rt::Hit h = query<true>(s, o, d, r->tnear, r->tfar);
    if (h.shape >= 0) r->tfar = -std::numeric_limits<float>::infinity();
}

} // extern "C"
