// envmap.h -- latitude-longitude environment light: lookup, importance sampling, pdf, adjoint.
//
// Behavioural spec: envmap_eval src/envmap.h:62-103, d_envmap_eval :105-203, tent_inv_cdf :205-212,
// envmap_sample :214-253, envmap_pdf :255-306.  Reference quirks that are reproduced are tagged [quirk].
#pragma once
#include "bsdf.h"

namespace rdr {

RDR_FN double safe_acos(double x) {
    if (x >= 1) return 0;
    if (x <= -1) return double(M_PI);
    return acos(x);
}

struct EnvLookup { V3 local_dir; V2 uv, du_dxy, dv_dxy; V3 ld_dx, ld_dy; double du_dx_l, du_dz_l, dv_dy_l; };
RDR_FN EnvLookup env_lookup(const EnvmapD &env, V3 local_dir, const RayDiff &rd) {
    EnvLookup e;
    e.local_dir = local_dir;
    e.uv = v2(atan2(local_dir.x, -local_dir.z) / double(2 * M_PI), safe_acos(local_dir.y) / double(M_PI));
    e.ld_dx = xfm_vector(env.world_to_env, rd.dir_dx);
    e.ld_dy = xfm_vector(env.world_to_env, rd.dir_dy);
    double xz = sq(local_dir.x) + sq(local_dir.z);
    e.du_dx_l = local_dir.x / (double(2 * M_PI) * xz);
    e.du_dz_l = local_dir.z / (double(2 * M_PI) * xz);
    e.du_dxy = v2(e.du_dx_l * e.ld_dx.x + e.du_dz_l * e.ld_dx.z, e.du_dx_l * e.ld_dy.x + e.du_dz_l * e.ld_dy.z);
    e.dv_dy_l = -1 / (double(M_PI) * sqrt(1 - sq(local_dir.y)));
    e.dv_dxy = v2(e.dv_dy_l * e.ld_dx.y, e.dv_dy_l * e.ld_dy.y);
    return e;
}

// Radiance arriving from direction `dir` (world space); `rd` selects the mip level.
RDR_FN V3 envmap_eval(const EnvmapD &env, V3 dir, const RayDiff &rd) {
    V3 local_dir = normalize(xfm_vector(env.world_to_env, dir));
    double out[3];
    if (local_dir.y < 1) {
        EnvLookup e = env_lookup(env, local_dir, rd);
        tex_fetch(env.values, e.uv, e.du_dxy, e.dv_dxy, out);
    } else {      // singularity at the pole: finest level
        V2 uv = v2(atan2(local_dir.x, -local_dir.z) / double(2 * M_PI), safe_acos(local_dir.y) / double(M_PI));
        tex_fetch(env.values, uv, v2(0, 0), v2(0, 0), out);
    }
    return V3{out[0], out[1], out[2]};
}

// Adjoint of envmap_eval: texel and world_to_env gradients, and the direction / differential adjoints.
RDR_FN void adj_envmap_eval(const EnvmapD &env, V3 dir, const RayDiff &rd, V3 o_bar, const GEnvmap *g,
                            V3 &dir_bar, RayDiff &rd_bar) {
    V3 n_local = xfm_vector(env.world_to_env, dir);
    V3 ld = normalize(n_local);
    EnvLookup e = env_lookup(env, ld, rd);       // [quirk] no pole special case in the adjoint
    V2 uv_bar = v2(0, 0), du_bar = v2(0, 0), dv_bar = v2(0, 0);
    double ob[3] = {o_bar.x, o_bar.y, o_bar.z};
    GTex none;
    for (int i = 0; i < kMaxMip; ++i) none.texels[i] = nullptr;
    none.uv_scale = nullptr;
    adj_tex_fetch(env.values, e.uv, e.du_dxy, e.dv_dxy, ob, g ? g->values : none, uv_bar, du_bar, dv_bar);
    double dvdy_bar = dv_bar.x * e.ld_dx.y + dv_bar.y * e.ld_dy.y;
    V3 lddx_bar = V3{0.0, dv_bar.x * e.dv_dy_l, 0.0};
    V3 lddy_bar = V3{0.0, dv_bar.y * e.dv_dy_l, 0.0};
    double one_m = 1 - sq(ld.y);
    V3 ld_bar = V3{0.0, -dvdy_bar * ld.y / (double(M_PI) * sqrt(one_m) * one_m), 0.0};
    double dudx_bar = du_bar.x * e.ld_dx.x + du_bar.y * e.ld_dy.x;
    double dudz_bar = du_bar.x * e.ld_dx.z + du_bar.y * e.ld_dy.z;
    lddx_bar.x += du_bar.x * e.du_dx_l;
    lddx_bar.z += du_bar.x * e.du_dz_l;
    lddy_bar.x += du_bar.y * e.du_dx_l;
    lddy_bar.z += du_bar.y * e.du_dz_l;
    double xz = sq(ld.x) + sq(ld.z);
    ld_bar.z += dudz_bar * (sq(ld.x) - sq(ld.z)) / (double(2 * M_PI) * sq(xz));
    ld_bar.x -= dudz_bar * ld.x * ld.z / (double(2 * M_PI) * sq(xz));
    ld_bar.x += dudx_bar * (sq(ld.z) - sq(ld.x)) / (double(2 * M_PI) * sq(xz));
    ld_bar.z -= dudx_bar * ld.x * ld.z / (double(2 * M_PI) * sq(xz));
    M4 w2e_bar = m4_zero();
    adj_xfm_vector(env.world_to_env, rd.dir_dx, lddx_bar, w2e_bar, rd_bar.dir_dx);
    adj_xfm_vector(env.world_to_env, rd.dir_dy, lddy_bar, w2e_bar, rd_bar.dir_dy);
    if (xz > 0.f) {
        ld_bar.x += (-uv_bar.x * ld.z / (xz * double(2 * M_PI)));
        ld_bar.z += (-uv_bar.x * ld.x / (xz * double(2 * M_PI)));
    }
    if (ld.y < 1.f) ld_bar.y += (-uv_bar.y / (sqrt(1 - sq(ld.y)) * (double(2 * M_PI))));   // [quirk] 2 pi, v = acos(y) / pi
    V3 nl_bar = adj_normalize(n_local, ld_bar);
    adj_xfm_vector(env.world_to_env, dir, nl_bar, w2e_bar, dir_bar);
    if (g && g->world_to_env)
        _Pragma("unroll") for (int r = 0; r < 4; ++r) { _Pragma("unroll") for (int c = 0; c < 4; ++c) accum(g->world_to_env + 4 * r + c, w2e_bar.m[r][c]); }
}

RDR_FN double tent_inv_cdf(double x) {
    if (x < 0.5) return 1 - sqrt(2 * x);
    return sqrt(2 * x - 0.5f) - 1;          // [quirk] as written in the reference
}

RDR_FN int upper_bound_f32(const float *a, int n, double v) {     // first element > v
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (!(v < a[mid])) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Direction (world space, not normalised by the reference either) drawn proportionally to luminance * sin(theta).
RDR_FN V3 envmap_sample(const EnvmapD &env, V2 s) {
    int W = env.values.width[0], H = env.values.height[0];
    const float *cy = env.sample_cdf_ys;
    int y = iclamp(upper_bound_f32(cy, H, s.y) - 1, 0, H - 1);
    if (y < H - 1) s.y = (s.y - cy[y]) / (cy[y + 1] - cy[y]);
    else s.y = (s.y - cy[y]) / (1 - cy[y]);
    const float *cx = env.sample_cdf_xs + (size_t)y * W;
    int x = iclamp(upper_bound_f32(cx, W, s.x) - 1, 0, W - 1);
    if (x < W - 1) s.x = (s.x - cx[x]) / (cx[x + 1] - cx[x]);
    else s.x = (s.x - cx[x]) / (1 - cx[x]);
    V2 uv = v2(x + tent_inv_cdf(s.x), y + tent_inv_cdf(s.y));
    double phi = (2 * double(M_PI) / W) * (uv.x + 0.5f);
    double theta = (double(M_PI) / H) * (uv.y + 0.5f);
    double sp = sin(phi), cp = cos(phi), st = sin(theta), ct = cos(theta);
    return xfm_vector(env.env_to_world, V3{sp * st, ct, -cp * st});
}

RDR_FN float lum_f32(const float *t) { return 0.212671f * t[0] + 0.715160f * t[1] + 0.072169f * t[2]; }

// Solid-angle density of envmap_sample() in direction `dir`.
RDR_FN double envmap_pdf(const EnvmapD &env, V3 dir) {
    V3 ld = xfm_vector(env.world_to_env, dir);           // [quirk] not normalised
    V2 uv = v2(atan2(ld.x, -ld.z) / double(2 * M_PI), safe_acos(ld.y) / double(M_PI));
    int w = env.values.width[0], h = env.values.height[0];
    double x = uv.x * w - 0.5f, y = uv.y * h - 0.5f;
    int xfi = imod((int)floor(x), w), yfi = imod((int)floor(y), h);
    int xci = imod(xfi + 1, w), yci = imod(yfi + 1, h);
    double dx = x - xfi, dy = y - yfi;
    if (dx < 0) dx += w;
    if (dy < 0) dy += h;
    const float *tx = env.values.texels[0];
    double lff = lum_f32(tx + 3 * ((size_t)yfi * w + xfi)), lcf = lum_f32(tx + 3 * ((size_t)yfi * w + xci));
    double lfc = lum_f32(tx + 3 * ((size_t)yci * w + xfi)), lcc = lum_f32(tx + 3 * ((size_t)yci * w + xci));
    double lum_fy = lff * (1.f - dx) * (1.f - dy) + lcf * dx * (1.f - dy);
    double lum_cy = lfc * (1.f - dx) * dy + lcc * dx * dy;
    double st = sqrt(dmax(1 - sq(ld.y), 0.0));
    if (st == 0.f) return 0.f;
    double st_fy = fabs(sin(double(M_PI) * (yfi + 0.5f) / h));
    double st_cy = fabs(sin(double(M_PI) * (yci + 0.5f) / h));
    return env.pdf_norm * fabs(lum_fy * st_fy + lum_cy * st_cy) / st;
}

} // namespace rdr
