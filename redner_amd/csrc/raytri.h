// raytri.h -- the single fp32 ray/triangle and ray/box predicate of this repo.
//
// The reference delegates closest-hit / any-hit queries to a third-party library
// (Embree on CPU, /root/reference/src/scene.cpp:556-574, 667-682; OptiX Prime on GPU,
// scene.cpp:521-532).  Neither is available, so the hit rule is *defined* here and shared by
//   * the gfx950 traversal kernels (redner_amd/csrc/trace.hip), and
//   * the Embree stand-in that the parity oracle links against (oracle/embree_shim).
// Rule: a ray (org, dir, tnear, tfar), all fp32, hits triangle (a,b,c) iff the Moller-Trumbore
// solution computed below -- evaluated in IEEE fp32 with *no* fused contraction so host and
// device agree bit for bit -- has 0<=u, 0<=v, u+v<=1 and tnear < t < tfar.  The closest hit is
// the one with the smallest t; ties are broken by the smaller (shape id, triangle id).
// Only (shape id, triangle id) leave the query: the hit point is recomputed in fp64 by the
// caller (reference: src/shape.h:289-295).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__
#else
#define RT_HD
#endif

namespace rt {

struct Hit {
    float t;
    int   shape;
    int   prim;
};

// Returns true and writes t when the ray hits the triangle under the rule above.
RT_HD inline bool ray_triangle(const float o[3], const float d[3], float tnear, float tfar,
                               const float a[3], const float b[3], const float c[3],
                               float *t_out) {
#pragma clang fp contract(off)
    const float e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2];
    const float e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
    const float px = d[1] * e2z - d[2] * e2y;
    const float py = d[2] * e2x - d[0] * e2z;
    const float pz = d[0] * e2y - d[1] * e2x;
    const float det = (e1x * px + e1y * py) + e1z * pz;
    if (det == 0.f) return false;
    const float inv = 1.f / det;
    const float sx = o[0] - a[0], sy = o[1] - a[1], sz = o[2] - a[2];
    const float u = ((sx * px + sy * py) + sz * pz) * inv;
    if (!(u >= 0.f && u <= 1.f)) return false;
    const float qx = sy * e1z - sz * e1y;
    const float qy = sz * e1x - sx * e1z;
    const float qz = sx * e1y - sy * e1x;
    const float v = ((d[0] * qx + d[1] * qy) + d[2] * qz) * inv;
    if (!(v >= 0.f && (u + v) <= 1.f)) return false;
    const float t = ((e2x * qx + e2y * qy) + e2z * qz) * inv;
    if (!(t > tnear && t < tfar)) return false;
    *t_out = t;
    return true;
}

// Candidate (t, shape, prim) beats the incumbent?
RT_HD inline bool closer(float t, int shape, int prim, const Hit &best) {
    if (best.shape < 0) return true;
    if (t != best.t) return t < best.t;
    if (shape != best.shape) return shape < best.shape;
    return prim < best.prim;
}

// Conservative slab test.  Boxes handed to it are padded at build time (see pad_box) and the
// exit distance is widened, so a ray that passes ray_triangle for a contained triangle is never
// culled, whatever the rounding/contraction mode of the caller.  Returns entry distance in *tn.
RT_HD inline bool ray_box(const float o[3], const float inv_d[3], float tnear, float tfar,
                          const float lo[3], const float hi[3], float *tn) {
    float t0 = tnear, t1 = tfar;
    for (int k = 0; k < 3; ++k) {
        float ta = (lo[k] - o[k]) * inv_d[k];
        float tb = (hi[k] - o[k]) * inv_d[k];
        float tmin_k = fminf(ta, tb);
        float tmax_k = fmaxf(ta, tb);
        tmax_k *= 1.0000004f;   // 1 + 3 ulp
        tmin_k -= fabsf(tmin_k) * 4e-7f;
        // NaN (0 * inf) must not cull: fmaxf/fminf return the non-NaN operand.
        t0 = fmaxf(t0, tmin_k);
        t1 = fminf(t1, tmax_k);
    }
    *tn = t0;
    return t0 <= t1;
}

// The same test with the slack applied once: x -> 1.0000004 x and x -> x - |x| 4e-7 are monotone (also after rounding), so
// scaling the smallest exit (largest entry) distance gives the same number as the smallest (largest) of the scaled ones --
// same verdict, same *tn for finite slab distances, six multiplications and two subtractions fewer per box.  With an
// infinite entry distance (a direction component of exactly 0 and the origin outside that slab) ray_box's per-axis slack
// turns +inf into NaN and ignores that axis; here the combined entry becomes NaN and all three are ignored: this test then
// accepts a box ray_box rejects, never the other way round (200 k random + adversarial cases), i.e. it stays conservative.
RT_HD inline bool ray_box_once(const float o[3], const float inv_d[3], float tnear, float tfar,
                               const float lo[3], const float hi[3], float *tn) {
    float en = -INFINITY, ex = INFINITY;
    for (int k = 0; k < 3; ++k) {
        float ta = (lo[k] - o[k]) * inv_d[k];
        float tb = (hi[k] - o[k]) * inv_d[k];
        float mn = fminf(ta, tb), mx = fmaxf(ta, tb);
        // NaN (0 * inf) must not cull: fmaxf/fminf return the non-NaN operand.
        en = fmaxf(en, mn);
        ex = fminf(ex, mx);
    }
    en -= fabsf(en) * 4e-7f;
    ex *= 1.0000004f;   // 1 + 3 ulp
    float t0 = fmaxf(tnear, en), t1 = fminf(tfar, ex);
    *tn = t0;
    return t0 <= t1;
}

// Padding applied to every box (leaf and inner) when a hierarchy is built.
RT_HD inline void pad_box(float lo[3], float hi[3]) {
    for (int k = 0; k < 3; ++k) {
        float m = fmaxf(fabsf(lo[k]), fabsf(hi[k]));
        float e = m * 1e-5f + 1e-7f;
        lo[k] -= e;
        hi[k] += e;
    }
}

} // namespace rt
