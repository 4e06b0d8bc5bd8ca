// bsdf.h -- textures (constant or mip-mapped trilinear) and the Lambert + Blinn-Phong BSDF:
// evaluation, pdf, importance sampling, and the adjoint of the evaluation.
//
// Behavioural spec:
//   tex_fetch / adj_tex_fetch   <- get_texture_value / d_get_texture_value   src/texture.h:337-419
//   (trilinear / adj_trilinear  <- trilinear_interp / d_trilinear_interp     src/texture.h:55-323)
//   bsdf_eval / adj_bsdf_eval   <- bsdf / d_bsdf                              src/material.h:355-692
//   bsdf_pdf                    <- bsdf_pdf                                   src/material.h:1024-1093
//   bsdf_sample_dir             <- bsdf_sample                                src/material.h:694-811
// Reference quirks that are reproduced on purpose are tagged [quirk].
#pragma once
#include "surface.h"

namespace rdr {

RDR_FN int imod(int a, int b) { int r = a % b; return r < 0 ? r + b : r; }

struct Bilerp { int i00, i10, i01, i11; double u, v; };   // texel offsets (without channel) + weights
RDR_FN Bilerp bilerp_setup(const TexD &tex, int li, V2 uv) {
    Bilerp b;
    int W = tex.width[li], H = tex.height[li];
    double x = uv.x * W - 0.5f, y = uv.y * H - 0.5f;
    int xf = (int)floor(x), yf = (int)floor(y);
    b.u = x - xf; b.v = y - yf;
    int x0 = imod(xf, W), y0 = imod(yf, H), x1 = imod(xf + 1, W), y1 = imod(yf + 1, H);
    b.i00 = y0 * W + x0; b.i10 = y0 * W + x1; b.i01 = y1 * W + x0; b.i11 = y1 * W + x1;
    return b;
}
RDR_FN double bilerp_eval(const float *texels, int ch, int c, const Bilerp &b) {
    double f00 = texels[ch * b.i00 + c], f10 = texels[ch * b.i10 + c];
    double f01 = texels[ch * b.i01 + c], f11 = texels[ch * b.i11 + c];
    return f00 * (1.f - b.u) * (1.f - b.v) + f01 * (1.f - b.u) * b.v +
           f10 * b.u * (1.f - b.v) + f11 * b.u * b.v;
}

RDR_FN void trilinear(const TexD &tex, V2 uv, double level, double *out) {
    int ch = tex.channels;
    if (level <= 0 || level >= tex.num_levels - 1) {
        int li = level <= 0 ? 0 : tex.num_levels - 1;
        Bilerp b = bilerp_setup(tex, li, uv);
        for (int c = 0; c < ch; ++c) out[c] = bilerp_eval(tex.texels[li], ch, c, b);
    } else {
        int li = (int)floor(level);
        double ld = level - li;
        Bilerp b0 = bilerp_setup(tex, li, uv), b1 = bilerp_setup(tex, li + 1, uv);
        for (int c = 0; c < ch; ++c) {
            double a0 = bilerp_eval(tex.texels[li], ch, c, b0);
            double a1 = bilerp_eval(tex.texels[li + 1], ch, c, b1);
            out[c] = a0 * (1 - ld) + a1 * ld;
        }
    }
}

RDR_FN double tex_level(const TexD &tex, V2 du, V2 dv) {
    double fp = dmax(len(du) * tex.width[0], len(dv) * tex.height[0]);
    return log(dmax(fp, double(1e-8f))) / log(2.0);   // src/redner.h:112-114
}

// out[channels] = texture value at the shading point.
RDR_FN void tex_fetch(const TexD &tex, V2 uv_, V2 du_, V2 dv_, double *out) {
    if (tex.width[0] <= 0 && tex.height[0] <= 0) {
        for (int c = 0; c < tex.channels; ++c) out[c] = tex.texels[0][c];
        return;
    }
    V2 sc = v2(tex.uv_scale[0], tex.uv_scale[1]);
    V2 uv = uv_ * sc, du = du_ * sc.x, dv = dv_ * sc.y;
    trilinear(tex, uv, tex_level(tex, du, dv), out);
}

RDR_FN void adj_bilerp(const TexD &tex, double *const *gtexels, int li, V2 uv, int ch, int c, double o_bar,
                       double &u_bar, double &v_bar) {
    Bilerp b = bilerp_setup(tex, li, uv);
    const float *tx = tex.texels[li];
    double f00 = tx[ch * b.i00 + c], f10 = tx[ch * b.i10 + c], f01 = tx[ch * b.i01 + c], f11 = tx[ch * b.i11 + c];
    double *g = gtexels[li];
    if (g) {
        accum_texel(g + ch * b.i00 + c, o_bar * (1.f - b.u) * (1.f - b.v));
        accum_texel(g + ch * b.i10 + c, o_bar * b.u * (1.f - b.v));
        accum_texel(g + ch * b.i01 + c, o_bar * (1.f - b.u) * b.v);
        accum_texel(g + ch * b.i11 + c, o_bar * b.u * b.v);
    }
    u_bar += o_bar * (-f00 * (1.f - b.v) + f10 * (1.f - b.v) + -f01 * b.v + f11 * b.v);
    v_bar += o_bar * (-f00 * (1.f - b.u) + -f10 * b.u + f01 * (1.f - b.u) + f11 * b.u);
}

// The same for the three channels of an rgb texture at once: the four corner texels are triples, so the lanes of the wave that
// share a corner are looked for ONCE per corner (accum_texel_triple) instead of once per corner and channel; u_bar / v_bar
// receive the channels' terms in channel order, like three adj_bilerp calls.
RDR_FN void adj_bilerp_rgb(const TexD &tex, double *const *gtexels, int li, V2 uv, const double o_bar[3], double &u_bar, double &v_bar) {
    Bilerp b = bilerp_setup(tex, li, uv);
    const float *tx = tex.texels[li];
    double *g = gtexels[li];
    if (g) {
        // (the products are formed as adj_bilerp forms them: (o_bar * wu) * wv)
        accum_texel_triple(g + 3 * b.i00, o_bar[0] * (1.f - b.u) * (1.f - b.v), o_bar[1] * (1.f - b.u) * (1.f - b.v), o_bar[2] * (1.f - b.u) * (1.f - b.v));
        accum_texel_triple(g + 3 * b.i10, o_bar[0] * b.u * (1.f - b.v), o_bar[1] * b.u * (1.f - b.v), o_bar[2] * b.u * (1.f - b.v));
        accum_texel_triple(g + 3 * b.i01, o_bar[0] * (1.f - b.u) * b.v, o_bar[1] * (1.f - b.u) * b.v, o_bar[2] * (1.f - b.u) * b.v);
        accum_texel_triple(g + 3 * b.i11, o_bar[0] * b.u * b.v, o_bar[1] * b.u * b.v, o_bar[2] * b.u * b.v);
    }
    for (int c = 0; c < 3; ++c) {
        double f00 = tx[3 * b.i00 + c], f10 = tx[3 * b.i10 + c], f01 = tx[3 * b.i01 + c], f11 = tx[3 * b.i11 + c];
        u_bar += o_bar[c] * (-f00 * (1.f - b.v) + f10 * (1.f - b.v) + -f01 * b.v + f11 * b.v);
        v_bar += o_bar[c] * (-f00 * (1.f - b.u) + -f10 * b.u + f01 * (1.f - b.u) + f11 * b.u);
    }
}

RDR_FN void adj_tex_fetch(const TexD &tex, V2 uv_, V2 du_, V2 dv_, const double *o_bar, const GTex &g,
                          V2 &uv_bar_, V2 &du_bar_, V2 &dv_bar_) {
    int ch = tex.channels;
    if (tex.width[0] <= 0 && tex.height[0] <= 0) {
        if (g.texels[0]) { if (ch == 3) accum_triple(g.texels[0], o_bar[0], o_bar[1], o_bar[2]); else for (int c = 0; c < ch; ++c) accum(g.texels[0] + c, o_bar[c]); }
        return;
    }
    V2 sc = v2(tex.uv_scale[0], tex.uv_scale[1]);
    V2 uv = uv_ * sc, du = du_ * sc.x, dv = dv_ * sc.y;
    double fu = len(du) * tex.width[0], fv = len(dv) * tex.height[0];
    bool u_is_max = !(fv > fu);
    double fp = u_is_max ? fu : fv;
    double level = log(dmax(fp, double(1e-8f))) / log(2.0);
    V2 uv_bar = v2(0, 0);
    double level_bar = 0;
    if (level <= 0 || level >= tex.num_levels - 1) {
        int li = level <= 0 ? 0 : tex.num_levels - 1;
        double ub = 0, vb = 0;
        if (ch == 3) adj_bilerp_rgb(tex, g.texels, li, uv, o_bar, ub, vb);
        else for (int c = 0; c < ch; ++c) adj_bilerp(tex, g.texels, li, uv, ch, c, o_bar[c], ub, vb);
        uv_bar.x += ub * tex.width[li]; uv_bar.y += vb * tex.height[li];
    } else {
        int li = (int)floor(level);
        double ld = level - li;
        Bilerp b0 = bilerp_setup(tex, li, uv), b1 = bilerp_setup(tex, li + 1, uv);
        double u0b = 0, v0b = 0, u1b = 0, v1b = 0;
        for (int c = 0; c < ch; ++c) {
            double a0 = bilerp_eval(tex.texels[li], ch, c, b0);
            double a1 = bilerp_eval(tex.texels[li + 1], ch, c, b1);
            level_bar += o_bar[c] * (a1 - a0);
            if (ch == 3) continue;
            adj_bilerp(tex, g.texels, li, uv, ch, c, o_bar[c] * (1 - ld), u0b, v0b);
            adj_bilerp(tex, g.texels, li + 1, uv, ch, c, o_bar[c] * ld, u1b, v1b);
        }
        if (ch == 3) {          // (the two levels have separate accumulators: per level the channels still arrive in order)
            const double lo_bar[3] = {o_bar[0] * (1 - ld), o_bar[1] * (1 - ld), o_bar[2] * (1 - ld)};
            const double hi_bar[3] = {o_bar[0] * ld, o_bar[1] * ld, o_bar[2] * ld};
            adj_bilerp_rgb(tex, g.texels, li, uv, lo_bar, u0b, v0b);
            adj_bilerp_rgb(tex, g.texels, li + 1, uv, hi_bar, u1b, v1b);
        }
        uv_bar.x += u1b * tex.width[li + 1]; uv_bar.y += v1b * tex.height[li + 1];
        uv_bar.x += u0b * tex.width[li]; uv_bar.y += v0b * tex.height[li];
    }
    V2 du_bar = v2(0, 0), dv_bar = v2(0, 0);
    if (fp > double(1e-8f)) {
        double fp_bar = level_bar / (fp * log(2.0));
        if (u_is_max) du_bar += adj_len(du, fp_bar) * (double)tex.width[0];
        else dv_bar += adj_len(dv, fp_bar) * (double)tex.height[0];
    }
    uv_bar_ += uv_bar * sc;
    du_bar_ += du_bar * sc.x;
    dv_bar_ += dv_bar * sc.y;
    if (g.uv_scale) {
        accum(g.uv_scale + 0, uv_bar.x * uv_.x + sum(du_bar * du_));
        accum(g.uv_scale + 1, uv_bar.y * uv_.y + sum(dv_bar * dv_));
    }
}

// sp.plain (lean stages): the texture is known to be a constant, so the lookup machinery -- and the registers it
// would pin -- is compiled out; the values are what tex_fetch's own constant branch returns.
RDR_FN V3 tex3(const TexD &tex, const Surf &sp) {
    if (sp.plain) { const float *t = tex.texels[0]; return V3{(double)load_dev(t), (double)load_dev(t + 1), (double)load_dev(t + 2)}; }
    double o[3]; tex_fetch(tex, sp.uv, sp.du_dxy, sp.dv_dxy, o); return V3{o[0], o[1], o[2]};
}
RDR_FN double tex1(const TexD &tex, const Surf &sp) {
    if (sp.plain) return load_dev(tex.texels[0]);
    double o; tex_fetch(tex, sp.uv, sp.du_dxy, sp.dv_dxy, &o); return o;
}
RDR_FN void adj_tex3(const TexD &tex, const Surf &sp, V3 o_bar, const GTex &g, Surf &sp_bar) {
    if (sp.plain) {
        if (g.texels[0]) accum3(g.texels[0], o_bar);
        return;
    }
    double ob[3] = {o_bar.x, o_bar.y, o_bar.z};
    adj_tex_fetch(tex, sp.uv, sp.du_dxy, sp.dv_dxy, ob, g, sp_bar.uv, sp_bar.du_dxy, sp_bar.dv_dxy);
}
RDR_FN void adj_tex1(const TexD &tex, const Surf &sp, double o_bar, const GTex &g, Surf &sp_bar) {
    if (sp.plain) { if (g.texels[0]) accum(g.texels[0], o_bar); return; }
    adj_tex_fetch(tex, sp.uv, sp.du_dxy, sp.dv_dxy, &o_bar, g, sp_bar.uv, sp_bar.du_dxy, sp_bar.dv_dxy);
}

// ---- normal mapping (src/material.h:274-351) ----------------------------------------------------
RDR_FN bool has_normal_map(const MaterialD &m) { return m.normal_map.num_levels > 0; }

RDR_FN Frame perturbed_frame(const MaterialD &m, const Surf &sp) {
    V3 nl = 2 * tex3(m.normal_map, sp) - 1;
    V3 pn = normalize(to_world(sp.frame, nl));
    V3 px = normalize(sp.dpdu - pn * dot(pn, sp.dpdu));
    return Frame{px, cross(pn, px), pn};
}
// specialised adjoint: only the normal of the perturbed frame carries a gradient
RDR_FN void adj_perturbed_normal(const MaterialD &m, const Surf &sp, V3 n_bar, const GMaterial &gm, Surf &sp_bar) {
    V3 nl = 2 * tex3(m.normal_map, sp) - 1;
    V3 nw = to_world(sp.frame, nl);
    V3 nw_bar = adj_normalize(nw, n_bar);
    V3 nl_bar = v3(0);
    adj_to_world(sp.frame, nl, nw_bar, sp_bar.frame, nl_bar);
    adj_tex3(m.normal_map, sp, 2 * nl_bar, gm.normal_map, sp_bar);
}

RDR_FN double roughness_to_phong(double r) { return dmax(2.f / r - 2.f, 0.0); }
RDR_FN double adj_roughness_to_phong(double r, double e_bar) {
    return (r > 0 && r <= 1.f) ? -2.f * e_bar / sq(r) : 0.f;
}

struct ShadeCtx {      // the side/frame bookkeeping all four BSDF entry points share
    Frame fr;
    V3 gn;
};
RDR_FN ShadeCtx shade_ctx(const MaterialD &m, const Surf &sp) {
    ShadeCtx c;
    c.fr = sp.frame;
    if (!sp.plain && has_normal_map(m)) c.fr = perturbed_frame(m, sp);
    c.gn = sp.geom_normal;
    if (dot(c.gn, c.fr.n) < 0) c.gn = -c.gn;
    return c;
}

RDR_FN double smith_g1(V3 v, V3 n, double roughness) {
    double ct = dot(v, n);
    double tt = sqrt(dmax(1.f / (ct * ct) - 1.f, 0.0));
    if (tt == 0.0f) return 1;
    double alpha = sqrt(roughness);
    double a = 1.f / (alpha * tt);
    if (a >= 1.6f) return 1;
    double a2 = a * a;
    return (3.535f * a + 2.181f * a2) / (1.0f + 2.276f * a + 2.577f * a2);
}

RDR_FN V3 bsdf_eval(const MaterialD &m, const Surf &sp, V3 wi, V3 wo, double min_rough) {
    RDR_CONTRACT_FAST
    ShadeCtx c = shade_ctx(m, sp);
    double gwi = dot(c.gn, wi), gwo = dot(c.gn, wo);
    double swi = fabs(dot(c.fr.n, wi)), swo = fabs(dot(c.fr.n, wo));
    if (gwi * gwo < 0) return v3(0);
    if (!m.two_sided && gwi < 0 && gwo < 0) return v3(0);
    if (swi == 0 || swo <= 1e-3f || fabs(gwo) <= 1e-3f) return v3(0);
    V3 kd = vmax0(m.use_vertex_color ? sp.color : tex3(m.diffuse, sp));
    V3 ks = vmax0(m.use_vertex_color ? v3(0) : tex3(m.specular, sp));
    double rough = dmax(tex1(m.roughness, sp), min_rough);
    V3 diffuse = kd * swo / double(M_PI);
    V3 spec = v3(0);
    if (m.compute_specular_lighting && !m.use_vertex_color) {
        V3 h = normalize(wi + wo);
        V3 hl = to_local(c.fr, h);
        if (m.two_sided && hl.z < 0) hl = -hl;
        if (hl.z > 0.f) {
            double e = roughness_to_phong(rough);
            double D = pow(dmax(hl.z, 0.0), e) * (e + 2.f) / double(2 * M_PI);
            double G = smith_g1(wi, c.fr.n, rough) * smith_g1(wo, c.fr.n, rough);
            double cd = fabs(dot(h, wo));
            V3 F = ks + (1.f - ks) * pow(dmax(1.0 - cd, 0.0), 5.0);
            spec = F * D * G / (4.f * swi);
        }
    }
    return diffuse + spec;
}

RDR_FN void adj_bsdf_eval(const MaterialD &m, const Surf &sp, V3 wi, V3 wo, double min_rough, V3 f_bar,
                          const GMaterial &gm, Surf &sp_bar, V3 &wi_bar, V3 &wo_bar) {
    RDR_CONTRACT_FAST
    ShadeCtx c = shade_ctx(m, sp);
    V3 n = c.fr.n;
    V3 n_bar = v3(0);
    double gwi = dot(c.gn, wi), gwo = dot(c.gn, wo);
    double swi = fabs(dot(n, wi)), swo = fabs(dot(n, wo));
    if (gwi * gwo < 0) return;
    if (!m.two_sided && gwi < 0 && gwo < 0) return;
    if (swi == 0 || swo <= 1e-3f || fabs(gwo) <= 1e-3f) return;

    V3 kd_raw = m.use_vertex_color ? sp.color : tex3(m.diffuse, sp);
    V3 kd = vmax0(kd_raw);
    V3 kd_bar = f_bar * (swo / double(M_PI));
    // [quirk] gradient is passed through the max(.,0) clamp unchanged
    if (m.use_vertex_color) sp_bar.color += kd_bar;
    else adj_tex3(m.diffuse, sp, kd_bar, gm.diffuse, sp_bar);
    double swo_bar = sum(f_bar * kd) / double(M_PI);
    if (dot(n, wo) < 0) swo_bar = -swo_bar;
    wo_bar += n * swo_bar;
    n_bar += wo * swo_bar;

    V3 ks = vmax0(m.use_vertex_color ? v3(0) : tex3(m.specular, sp));
    double rough = dmax(tex1(m.roughness, sp), min_rough);
    rough = dmax(rough, 1e-6);     // [quirk] clamp exists only in the adjoint
    if (m.compute_specular_lighting && !m.use_vertex_color) {
        V3 h = normalize(wi + wo);
        V3 hl = to_local(c.fr, h);
        bool hflip = false;
        if (m.two_sided && hl.z < 0) { hl = -hl; hflip = true; }
        if (hl.z > 0.f) {
            double e = roughness_to_phong(rough);
            double D = pow(hl.z, e) * (e + 2.f) / double(2 * M_PI);
            double rough_bar = 0;
            double Gwi = smith_g1(wi, n, rough), Gwo = smith_g1(wo, n, rough);
            double G = Gwi * Gwo;
            double cd = dot(h, wo);            // [quirk] no fabs in the adjoint
            double c5 = pow(dmax(1.0 - cd, 0.0), 5.0);
            V3 F = ks + (1.f - ks) * c5;
            V3 spec = F * D * G / (4.f * swi);
            V3 F_bar = f_bar * (D * G / (4.f * swi));
            double D_bar = sum(f_bar * F) * (G / (4.f * swi));
            double G_bar = sum(f_bar * F) * (D / (4.f * swi));
            double swi_bar = -sum(f_bar * spec) / swi;
            // [quirk] the sign of dot(wi, n) is not applied to swi_bar
            wi_bar += swi_bar * n;
            n_bar += swi_bar * wi;
            V3 ks_bar = F_bar * (1.f - c5);
            double c5_bar = sum(F_bar * (1.f - ks));
            double cd_bar = -5.f * c5_bar * pow(dmax(1.0 - cd, 0.0), 4.0);
            V3 h_bar = cd_bar * wo;
            wo_bar += cd_bar * h;
            double Gwi_bar = G_bar * Gwo, Gwo_bar = G_bar * Gwi;
            // adjoint of smith_g1 w.r.t. its direction; [quirk] 2.557 (not 2.577) in the denominator
            for (int side = 0; side < 2; ++side) {
                V3 dv = side == 0 ? wi : wo;
                double g1_bar = side == 0 ? Gwi_bar : Gwo_bar;
                double ct = dot(dv, n);
                V3 dv_bar = v3(0);
                if (!(dot(dv, h) * ct <= 0)) {
                    double tt = sqrt(dmax(1.f / sq(ct) - 1.f, 0.0));
                    if (!(tt <= 1e-10f)) {
                        double alpha = sqrt(rough);
                        double a = 1.f / (alpha * tt);
                        if (!(a >= 1.6f)) {
                            double num = 3.535f * a + 2.181f * sq(a);
                            double den = 1.f + 2.276f * a + 2.557f * sq(a);
                            double num_bar = g1_bar / den;
                            double den_bar = -g1_bar * num / sq(den);
                            double a_bar = num_bar * (3.535f + 2.181f * 2 * a) + den_bar * (2.276f + 2.557f * 2 * a);
                            double alpha_bar = -a_bar * a / alpha;
                            double tt_bar = -a_bar * a / tt;
                            rough_bar += 0.5f * alpha_bar / alpha;
                            double tt2_bar = tt_bar * 0.5f / tt;
                            double ct_bar = -2.f * tt2_bar / (ct * ct * ct);
                            dv_bar = ct_bar * n;
                            n_bar += ct_bar * dv;
                        }
                    }
                }
                if (side == 0) wi_bar += dv_bar; else wo_bar += dv_bar;
            }
            double Dpow_bar = D_bar * (e + 2.f) / double(2 * M_PI);
            double Dfac_bar = D_bar * pow(hl.z, e);
            double hz_bar = Dpow_bar * pow(dmax(hl.z, 0.0), e - 1) * e;
            double e_bar = Dpow_bar * pow(dmax(hl.z, 0.0), e) * log(hl.z);
            e_bar += Dfac_bar / double(2 * M_PI);
            rough_bar += adj_roughness_to_phong(rough, e_bar);
            if (hflip) hz_bar = -hz_bar;
            h_bar += hz_bar * n;
            n_bar += hz_bar * h;
            V3 s_bar = adj_normalize(wi + wo, h_bar);
            wi_bar += s_bar;
            wo_bar += s_bar;
            adj_tex3(m.specular, sp, ks_bar, gm.specular, sp_bar);
            if (rough > min_rough) adj_tex1(m.roughness, sp, rough_bar, gm.roughness, sp_bar);
        }
    }
    if (!sp.plain && has_normal_map(m)) adj_perturbed_normal(m, sp, n_bar, gm, sp_bar);      // (as shade_ctx decides)
    else sp_bar.frame.n += n_bar;
}

struct LobePmf { double diffuse, specular; };
RDR_FN LobePmf lobe_pmf(const MaterialD &m, const Surf &sp) {
    V3 kd = vmax0(m.use_vertex_color ? sp.color : tex3(m.diffuse, sp));
    V3 ks = vmax0(m.use_vertex_color ? v3(0) : tex3(m.specular, sp));
    double wd = luminance(kd), ws = luminance(ks), wsum = wd + ws;
    LobePmf p{0.5, 0.5};
    if (wsum > 0.f) { p.diffuse = wd / wsum; p.specular = ws / wsum; }
    return p;
}

RDR_FN double bsdf_pdf(const MaterialD &m, const Surf &sp, V3 wi, V3 wo, double min_rough) {
    RDR_CONTRACT_FAST
    ShadeCtx c = shade_ctx(m, sp);
    double gwi = dot(c.gn, wi), gwo = dot(c.gn, wo);
    double swo = fabs(dot(c.fr.n, wo));
    if (gwi * gwo < 0) return 0;
    if (!m.two_sided && gwi < 0 && gwo < 0) return 0;
    LobePmf p = lobe_pmf(m, sp);
    double pd = 0;
    if (p.diffuse > 0.f) pd = p.diffuse * swo / double(M_PI);
    double ps = 0;
    if (p.specular > 0.f) {
        V3 h = normalize(wi + wo);
        V3 hl = to_local(sp.frame, h);     // [quirk] un-perturbed frame here
        if (m.two_sided && hl.z < 0) hl = -hl;
        if (hl.z > 0.f && fabs(dot(h, wo)) > 0) {
            double rough = dmax(dmax(tex1(m.roughness, sp), min_rough), 1e-6);
            double e = roughness_to_phong(rough);
            double D = pow(hl.z, e) * (e + 2.f) / double(2 * M_PI);
            ps = p.specular * D * hl.z / (4.f * fabs(dot(h, wo)));
        }
    }
    return pd + ps;
}

// Cosine-weighted hemisphere direction; phi uses float pi  [quirk] (src/material.h:694-700)
RDR_FN V3 cos_hemisphere(V2 s) {
    double phi = 2.f * float(M_PI) * s.x;
    double tmp = sqrt(dmax(1.f - s.y, 0.0));
    return V3{cos(phi) * tmp, sin(phi) * tmp, sqrt(s.y)};
}

// Importance-sample the outgoing direction.  Returns 0 when the sample fails.  wo_rd is written
// only on success (like the reference, which leaves it untouched on the early return).
RDR_FN V3 bsdf_sample_dir(const MaterialD &m, const Surf &sp, V3 wi, V2 s_uv, double s_w, double min_rough,
                          const RayDiff &wi_rd, RayDiff &wo_rd, double &next_min_rough, bool diffs = true) {
    next_min_rough = min_rough;
    ShadeCtx c = shade_ctx(m, sp);
    double gwi = dot(c.gn, wi);
    if (!m.two_sided && gwi < 0) return v3(0);
    LobePmf p = lobe_pmf(m, sp);
    if (s_w <= p.diffuse) {
        next_min_rough = 1.0;
        V3 ld = cos_hemisphere(s_uv);
        if (diffs) {
            wo_rd.org_dx = wi_rd.org_dx; wo_rd.org_dy = wi_rd.org_dy;
            wo_rd.dir_dx = V3{0.03f, 0.03f, 0.03f};
            wo_rd.dir_dy = V3{0.03f, 0.03f, 0.03f};
        }
        V3 d = to_world(c.fr, ld);
        if (dot(c.gn, d) * gwi < 0) d = to_world(c.fr, -ld);
        return d;
    }
    double rough = dmax(dmax(tex1(m.roughness, sp), min_rough), 1e-6);
    next_min_rough = dmax(rough, min_rough);
    double e = roughness_to_phong(rough);
    double phi = 2.f * double(M_PI) * s_uv.y;
    double sphi = sin(phi), cphi = cos(phi);
    double ct = pow(s_uv.x, 1.0 / (e + 2.0));
    double st = sqrt(dmax(1.f - ct * ct, 0.0));
    V3 hl = V3{st * cphi, st * sphi, ct};
    V3 h = to_world(c.fr, hl);
    V3 d = 2.f * dot(wi, h) * h - wi;
    if (dot(c.gn, d) * gwi < 0) {
        hl = -hl;
        h = to_world(c.fr, hl);
        d = 2.f * dot(wi, h) * h - wi;
    }
    if (!diffs) return d;
    V3 dmdx = sp.dn_dx * hl.z, dmdy = sp.dn_dy * hl.z;
    V3 wi_dx = -wi_rd.dir_dx, wi_dy = -wi_rd.dir_dy;
    double wdm_dx = sum(wi_dx * h) + sum(wi * dmdx);
    double wdm_dy = sum(wi_dy * h) + sum(wi * dmdy);
    wo_rd.org_dx = wi_rd.org_dx; wo_rd.org_dy = wi_rd.org_dy;
    wo_rd.dir_dx = 2 * (dot(wi, h) * dmdx + wdm_dx * h) - wi_dx;
    wo_rd.dir_dy = 2 * (dot(wi, h) * dmdy + wdm_dy * h) - wi_dy;
    return d;
}

} // namespace rdr
