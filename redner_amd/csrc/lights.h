// lights.h -- light selection for next-event estimation.
//
// Behavioural spec: light_point_sampler src/scene.cpp:692-741 (light by upper_bound on light_cdf,
// triangle by upper_bound on the per-light area CDF, uniform point; shadow ray tmin = 1e-3f,
// tmax = (1 - 1e-3f) * distance).  PMF/CDF construction is in scene_build.cpp.
#pragma once
#include "surface.h"

namespace rdr {

// index of the first element > v  (thrust::upper_bound semantics)
RDR_FN int upper_bound_idx(const double *a, int n, double v) {
    if (n <= 4) {
        // short tables -- a scene's lights, the two triangles of a quad light: every entry is loaded at once and the entries that
        // are not above v are counted (the table is non-decreasing, so that count IS the index of the first entry above v; a NaN
        // compares like the search does): no dependent load per halving step
        const double a0 = n > 0 ? a[0] : 0, a1 = n > 1 ? a[1] : 0, a2 = n > 2 ? a[2] : 0, a3 = n > 3 ? a[3] : 0;
        return (n > 0 && !(v < a0) ? 1 : 0) + (n > 1 && !(v < a1) ? 1 : 0) + (n > 2 && !(v < a2) ? 1 : 0) + (n > 3 && !(v < a3) ? 1 : 0);
    }
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (!(v < a[mid])) lo = mid + 1; else hi = mid;
    }
    return lo;
}

struct LightPick { int light_id, shape_id, tri_id; };

RDR_FN LightPick pick_light(const SceneD &sc, double light_sel, double tri_sel) {
    LightPick p;
    p.light_id = iclamp(upper_bound_idx(sc.light_cdf, sc.num_lights, light_sel) - 1, 0, sc.num_lights - 1);
    if (sc.envmap != nullptr && p.light_id == sc.num_lights - 1) {
        p.shape_id = -1; p.tri_id = -1;
        return p;
    }
    const LightD &l = sc.lights[p.light_id];
    const ShapeD &sh = sc.shapes[l.shape_id];
    const double *cdf = sc.area_cdf_pool + sc.area_cdf_offset[p.light_id];
    p.shape_id = l.shape_id;
    p.tri_id = iclamp(upper_bound_idx(cdf, sh.num_triangles, tri_sel) - 1, 0, sh.num_triangles - 1);
    return p;
}

// Shadow ray from `from` towards a sampled light point.
RDR_FN Ray shadow_ray_to(V3 from, V3 light_pos) {
    Ray r;
    r.org = from;
    r.dir = normalize(light_pos - from);
    r.tmin = 1e-3f;
    r.tmax = (1 - 1e-3f) * len(light_pos - from);
    return r;
}

} // namespace rdr
