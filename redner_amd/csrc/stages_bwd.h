// stages_bwd.h -- adjoint stages of the camera path (the "continuous" part of the gradient).
//
// Behavioural spec:
//   AdjBounce   <- d_path_contribs_accumulator          src/path_contribution.cpp:156-626
//   AdjPrimary  <- d_primary_contribs_accumulator (radiance)  src/primary_contribution.cpp:438-480
//                  + d_primary_intersector               src/primary_intersection.cpp:5-130
// Walking the path backwards, each vertex hands three adjoints to its predecessor: throughput,
// incoming-ray direction and shading point (the reference's d_throughputs / d_rays / d_points;
// its d_ray_differentials are identically zero because the BSDF-sampling adjoint is disabled,
// src/path_contribution.cpp:458-474, so they are not materialised here).  One buffer set is
// updated in place: lane p reads its successor's adjoints and overwrites them with its own.
//
// Like the reference this estimator deliberately ignores d(MIS), d(pdf_bsdf) and d(BSDF
// sampling) (:369-376, :458-474) and drops the position gradient of diffuse inter-reflection
// unless min_roughness > 0.01 (:447-455).
#pragma once
#include "stages_fwd.h"

namespace rdr {

constexpr int kAdjPointDoubles = 24;   // position 3, frame 9, dpdu 3, uv 2, du_dxy 2, dv_dxy 2, color 3
constexpr int kAdjPointDoublesLean = 6;

// The lean stages (plain scenes: constant reflectances, no normal map, radiance only) keep SIX components of the shading-point
// adjoint: position and shading normal.  Nothing writes the others there: the BSDF adjoint of an untextured material without a
// normal map adds to frame.n only (bsdf.h: adj_bsdf_eval), the edge estimators and the replay add to the position
// (stages_edge.h), the first-hit channels do not exist -- the tangent, bitangent, dp/du, uv and colour adjoints a record would
// carry are zeros from the first vertex to the camera, and adding a zero changes no sum.  (Round 6: 15 -> 6 doubles read and
// written per lane and stage, and the surface adjoint of the lean form loses the terms those zeros fed.)
struct AdjState {      // SoA, stride n, indexed by lane id
    double *thr;       // 3 x n
    double *ray_dir;   // 3 x n   (adjoint of the incoming ray's direction; the origin's is always 0)
    double *point;     // 24 x n  (non-zero part of the shading point adjoint); lean: 6 x n = position, frame.n
    int n;
    int plain;         // lean stages (see above)
    // per lane, or null (not tracked): 0 = the lane's record is all zeros -- cleared, and none of the stages that write records
    // (AdjBounceScatter, adj_record_add, SecondaryEdgeDerivatives) has come by since.  The continuation half of the next vertex up
    // the path skips such a lane unless its ray reached an emitter (render.cpp: adj_scatter).
    unsigned char *carries;
};

RDR_FN Surf load_adj_point(const AdjState &a, int p) {
    Surf s = surf_zero();
    s.position = ld3(a.point, a.n, p, 0);
    if (a.plain) { s.frame.n = ld3(a.point, a.n, p, 3); return s; }       // the lean record: position, shading normal (see AdjState)
    s.frame.x = ld3(a.point, a.n, p, 3); s.frame.y = ld3(a.point, a.n, p, 6); s.frame.n = ld3(a.point, a.n, p, 9);
    s.dpdu = ld3(a.point, a.n, p, 12);
    s.uv = v2(a.point[(size_t)15 * a.n + p], a.point[(size_t)16 * a.n + p]);
    s.du_dxy = v2(a.point[(size_t)17 * a.n + p], a.point[(size_t)18 * a.n + p]);
    s.dv_dxy = v2(a.point[(size_t)19 * a.n + p], a.point[(size_t)20 * a.n + p]);
    s.color = ld3(a.point, a.n, p, 21);
    return s;
}
RDR_FN void store_adj_point(const AdjState &a, int p, const Surf &s, bool with_position = true) {
    if (with_position) st3(a.point, a.n, p, 0, s.position);
    if (a.plain) { st3(a.point, a.n, p, 3, s.frame.n); return; }
    st3(a.point, a.n, p, 3, s.frame.x); st3(a.point, a.n, p, 6, s.frame.y); st3(a.point, a.n, p, 9, s.frame.n);
    st3(a.point, a.n, p, 12, s.dpdu);
    a.point[(size_t)15 * a.n + p] = s.uv.x; a.point[(size_t)16 * a.n + p] = s.uv.y;
    a.point[(size_t)17 * a.n + p] = s.du_dxy.x; a.point[(size_t)18 * a.n + p] = s.du_dxy.y;
    a.point[(size_t)19 * a.n + p] = s.dv_dxy.x; a.point[(size_t)20 * a.n + p] = s.dv_dxy.y;
    st3(a.point, a.n, p, 21, s.color);
}

RDR_FN V3 image_grad(const float *d_image, int nd, int radiance_dim, int pixel) {
    if (radiance_dim < 0) return v3(0);
    const float *g = d_image + (size_t)nd * pixel + radiance_dim;
    return V3{(double)g[0], (double)g[1], (double)g[2]};
}

// ---- adjoint of one bounce -------------------------------------------------------------------------
// Two stages per vertex so that each fits the register file without scratch:
//   AdjBounceScatter : the BSDF-sampled continuation.  Consumes the successor's adjoints and
//                      OVERWRITES the lane's adjoint record with this vertex's (partial) adjoints.
//   AdjBounceNee     : next-event estimation; ADDS its share to the record.
// (The reference does both in one functor, src/path_contribution.cpp:156-626; the sums are the same.)
struct AdjBounceArgs {
    SceneD sc; GScene g; SamplerD rng; int dim;
    const int *active; VSlice v, vn;
    const float *d_image; int nd, radiance_dim; double weight;
    AdjState adj;
};

struct AdjBounceScatter {
    AdjBounceArgs a;
    static constexpr int kMidBlocksPerCU = 2;          // 320 -> 256 registers (256 B of scratch), two waves per SIMD
    static constexpr int kMinBlocksPerCU = 2;          // general form
#ifndef RDR_ADJ_SCATTER_LEAN_BLOCKS          // (variant builds: tools/build_render_variant.sh x -DRDR_ADJ_SCATTER_LEAN_BLOCKS=4)
#define RDR_ADJ_SCATTER_LEAN_BLOCKS 3
#endif
    static constexpr int kLeanBlocksPerCU = RDR_ADJ_SCATTER_LEAN_BLOCKS;         // 190 -> 168 registers (100 B of scratch), three waves per SIMD: +2 % on the benchmark
    RDR_FN void make_lean() { lean_scene(a.sc); lean_slice(a.v); lean_slice(a.vn); a.nd = 3; a.radiance_dim = 0; a.adj.plain = 1; }
    RDR_FN void make_mid() { mid_scene(a.sc); a.nd = 3; a.radiance_dim = 0; }
    RDR_FN void operator()(int idx) const {
        const SceneD &sc = a.sc; const GScene &g = a.g; const VSlice &v = a.v, &vn = a.vn; const AdjState &adj = a.adj;
        int p = a.active[idx];
        VertexCtx c = load_vertex(sc, v, p);
        const GMaterial &gm = g.materials[c.shape->material_id];
        V3 thr = ld3(v.thr, v.n, p, 0);
        V3 pc_bar = a.weight * image_grad(a.d_image, a.nd, a.radiance_dim, p);   // adjoint of path_contrib
        V3 thr_bar = v3(0), in_dir_bar = v3(0);
        Surf sp_bar = surf_zero();
        V3 pos = c.sp.position;
        int bshape = vn.shape[p];
        TriGrad tg = trigrad_zero();          // gradient of the triangle hit by the continuation, scattered at the end
        int tg_shape = -1, tg_tri = -1;
        // Order of the two halves (register pressure: the Surf adjoints are 35 doubles each): first everything that needs this
        // vertex's surface point and accumulates into its adjoint sp_bar (the BSDF's adjoint); then sp_bar goes to the adjoint
        // record -- all but its position, which the second half still changes -- and only then is the successor's adjoint point
        // loaded (same record: read before it is overwritten) and pushed through the hit triangle's surface adjoint.
        V3 dir_bar = v3(0);
        bool through_hit = false;
        double d2 = 0;
        V3 wo = v3(0);
        if (bshape >= 0) {
            const ShapeD &bsh = sc.shapes[bshape];
            int btri = vn.tri[p];
            RayDiff wo_rd = load_rdiff(vn, p);
            RayDiff tmp;
            Surf bp = surf_at(bsh, btri, load_ray(vn, p), wo_rd, tmp, !sc.no_diffs);
            V3 next_thr_bar = ld3(adj.thr, adj.n, p, 0);
            V3 next_dir_bar = ld3(adj.ray_dir, adj.n, p, 0);
            V3 dir = bp.position - pos;
            d2 = len_sq(dir);
            wo = dir / sqrt(d2);
            double pdf_b = bsdf_pdf(*c.mat, c.sp, c.wi, wo, c.mrough);
            if (pdf_b > 0) {
                V3 f = bsdf_eval(*c.mat, c.sp, c.wi, wo, c.mrough);
                V3 sb = f / pdf_b;
                thr_bar += next_thr_bar * sb;
                V3 sb_bar = next_thr_bar * thr;
                V3 f_bar = sb_bar / pdf_b;
                if (bsh.light_id >= 0) {
                    const LightD &l = sc.lights[bsh.light_id];
                    if (l.two_sided || dot(-wo, bp.frame.n) > 0) {
                        double geo = fabs(dot(wo, bp.geom_normal)) / d2;
                        V3 Le = v3f(l.intensity);
                        double pdf_nee = (sc.light_pmf[bsh.light_id] * (1 / sc.light_areas[bsh.light_id])) / geo;
                        double mis = 1 / (1 + sq(pdf_nee / pdf_b));
                        V3 sc_contrib = (mis / pdf_b) * f * Le;
                        V3 scb = pc_bar * thr;
                        thr_bar += pc_bar * sc_contrib;
                        double w = mis / pdf_b;
                        f_bar += w * scb * Le;
                        if (g.light_intensity) accum3(g.light_intensity + 3 * bsh.light_id, w * scb * f);
                    }
                }
                V3 wi_bar = v3(0);
                V3 wo_bar = next_dir_bar;
                adj_bsdf_eval(*c.mat, c.sp, c.wi, wo, c.mrough, f_bar, gm, sp_bar, wi_bar, wo_bar);
                dir_bar = wo_bar / sqrt(d2);
                double sd_bar = -sum(wo_bar * dir) / d2;
                double d2_bar = 0.5f * sd_bar / sqrt(d2);
                dir_bar += adj_len_sq(dir, d2_bar);
                in_dir_bar -= wi_bar;
                through_hit = true;
            }
        } else if (sc.envmap != nullptr) {
            // the BSDF ray reached the environment light (src/path_contribution.cpp:520-600); MIS weight and
            // pdf are treated as constants, nothing flows into the sampling procedure
            V3 wo_e = load_ray(vn, p).dir;
            double pdf_b = bsdf_pdf(*c.mat, c.sp, c.wi, wo_e, c.mrough);
            if (len_sq(wo_e) > 0 && pdf_b > 0) {
                V3 f = bsdf_eval(*c.mat, c.sp, c.wi, wo_e, c.mrough);
                V3 Le = envmap_eval(*sc.envmap, wo_e, raydiff_zero());
                double pdf_nee = envmap_pdf(*sc.envmap, wo_e) * sc.light_pmf[sc.num_lights - 1];
                double mis = 1 / (1 + sq(pdf_nee / pdf_b));
                V3 sc_contrib = (mis / pdf_b) * f * Le;
                V3 scb = pc_bar * thr;
                thr_bar += pc_bar * sc_contrib;
                double w = mis / pdf_b;
                V3 f_bar = w * scb * Le, Le_bar = w * scb * f;
                V3 wo_bar = v3(0), wi_bar = v3(0);
                RayDiff rd_bar = raydiff_zero();
                adj_envmap_eval(*sc.envmap, wo_e, raydiff_zero(), Le_bar, g.envmap, wo_bar, rd_bar);
                adj_bsdf_eval(*c.mat, c.sp, c.wi, wo_e, c.mrough, f_bar, gm, sp_bar, wi_bar, wo_bar);
                in_dir_bar -= wi_bar;
            }
        }
        st3(adj.thr, adj.n, p, 0, thr_bar);
        st3(adj.ray_dir, adj.n, p, 0, in_dir_bar);
        V3 pos_bar = sp_bar.position;
        Surf next_pt_bar = surf_zero();
        if (through_hit) next_pt_bar = load_adj_point(adj, p);
        store_adj_point(adj, p, sp_bar, false);          // all but the position: the second half still changes it
        if (through_hit) {
            const ShapeD &bsh = sc.shapes[bshape];
            const int btri = vn.tri[p];
            Surf bp_bar = next_pt_bar;
            bp_bar.position += dir_bar;
            DRay r_bar = dray_zero();
            RayDiff rd_bar = raydiff_zero();
            Ray br = make_ray(pos, wo);
            adj_surf_at(bsh, btri, br, load_rdiff(vn, p), bp_bar, raydiff_zero(), r_bar, rd_bar, tg, !sc.no_diffs, sc.plain_materials != 0);
            tg_shape = bshape; tg_tri = btri;
            if (c.mrough > 0.01f) {
                pos_bar -= dir_bar;
                pos_bar += r_bar.org;
            }
        }
        st3(adj.point, adj.n, p, 0, pos_bar);
        if (adj.carries) adj.carries[p] = 1;
        scatter_trigrad_wave(sc.shapes, g.shapes, tg_shape, tg_tri, tg, sc.plain_materials != 0);   // every lane gets here
    }
};

// Adds one estimator's share to the adjoint record written by AdjBounceScatter.
RDR_FN void adj_record_add(const AdjState &adj, int p, V3 thr_bar, V3 in_dir_bar, const Surf &sp_bar) {
    st3(adj.thr, adj.n, p, 0, ld3(adj.thr, adj.n, p, 0) + thr_bar);
    st3(adj.ray_dir, adj.n, p, 0, ld3(adj.ray_dir, adj.n, p, 0) + in_dir_bar);
    Surf cur = load_adj_point(adj, p);
    cur.position += sp_bar.position;
    cur.frame.n += sp_bar.frame.n;
    if (!adj.plain) {
        cur.frame.x += sp_bar.frame.x; cur.frame.y += sp_bar.frame.y;
        cur.dpdu += sp_bar.dpdu; cur.uv += sp_bar.uv; cur.du_dxy += sp_bar.du_dxy; cur.dv_dxy += sp_bar.dv_dxy;
        cur.color += sp_bar.color;
    }
    store_adj_point(adj, p, cur);
    if (adj.carries) adj.carries[p] = 1;
}

struct AdjBounceNee {
    // mid specialisation at two waves per SIMD: 256 VGPR + 240 B of spills instead of 256 + 66 AGPR at one wave (0.89 -> 0.71 ms per
    // launch on the config-5 stand-in); the same cap costs AdjBounceScatter 650 B of spills and 1.8 -> 3.0 ms, so it keeps one wave
    static constexpr int kMinBlocksPerCU = 2;
    static constexpr int kMidBlocksPerCU = 2;
#ifdef RDR_ADJ_NEE_LEAN_BLOCKS              // (variant builds only; the lean form runs at 231 registers, two waves per SIMD)
    static constexpr int kLeanBlocksPerCU = RDR_ADJ_NEE_LEAN_BLOCKS;
#endif
    AdjBounceArgs a;
    RDR_FN void make_lean() { lean_scene(a.sc); lean_slice(a.v); lean_slice(a.vn); a.nd = 3; a.radiance_dim = 0; a.adj.plain = 1; }
    RDR_FN void make_mid() { mid_scene(a.sc); a.nd = 3; a.radiance_dim = 0; }
    // next-event estimation towards the environment light (src/path_contribution.cpp:295-338)
    RDR_FN void envmap_nee(int p, const LightDraw &ld) const {
        const SceneD &sc = a.sc; const GScene &g = a.g; const VSlice &v = a.v;
        V3 wo = envmap_sample(*sc.envmap, ld.uv);
        double pdf_nee = envmap_pdf(*sc.envmap, wo) * sc.light_pmf[sc.num_lights - 1];
        if (!(pdf_nee > 0)) return;
        VertexCtx c = load_vertex(sc, v, p);
        const GMaterial &gm = g.materials[c.shape->material_id];
        V3 thr = ld3(v.thr, v.n, p, 0);
        V3 pc_bar = a.weight * image_grad(a.d_image, a.nd, a.radiance_dim, p);
        V3 f = bsdf_eval(*c.mat, c.sp, c.wi, wo, c.mrough);
        V3 Le = envmap_eval(*sc.envmap, wo, raydiff_zero());
        double pdf_b = bsdf_pdf(*c.mat, c.sp, c.wi, wo, c.mrough);
        double mis = 1 / (1 + sq(pdf_b / pdf_nee));
        V3 nee = (mis / pdf_nee) * f * Le;
        V3 nee_bar = pc_bar * thr;
        V3 thr_bar = pc_bar * nee;
        double w = mis / pdf_nee;
        V3 f_bar = w * nee_bar * Le, Le_bar = w * nee_bar * f;
        V3 wo_bar = v3(0), wi_bar = v3(0);
        RayDiff rd_bar = raydiff_zero();
        adj_envmap_eval(*sc.envmap, wo, raydiff_zero(), Le_bar, g.envmap, wo_bar, rd_bar);
        Surf sp_bar = surf_zero();
        adj_bsdf_eval(*c.mat, c.sp, c.wi, wo, c.mrough, f_bar, gm, sp_bar, wi_bar, wo_bar);
        adj_record_add(a.adj, p, thr_bar, -wi_bar, sp_bar);
    }
    RDR_FN void operator()(int idx) const {
        V3 lv_bar[3] = {v3(0), v3(0), v3(0)};      // gradient of the sampled light triangle, scattered by the whole wave
        int l_shape = -1, l_tri = -1;
        RDR_INLINE_CALL light_vertex(idx, lv_bar, l_shape, l_tri);
        scatter_positions_wave(a.sc.shapes, a.g.shapes, l_shape, l_tri, lv_bar);
    }
    RDR_FN void light_vertex(int idx, V3 (&lv_bar)[3], int &l_shape, int &l_tri) const {
        const SceneD &sc = a.sc; const GScene &g = a.g; const VSlice &v = a.v; const AdjState &adj = a.adj;
        int p = a.active[idx];
        if (v.occl[p]) return;
        LightDraw ld = draw_light(a.rng, p, a.dim);
        LightPick pk = pick_light(sc, ld.light_sel, ld.tri_sel);
        if (pk.shape_id < 0) {
            if (sc.envmap != nullptr) envmap_nee(p, ld);
            return;
        }
        const ShapeD &lsh = sc.shapes[pk.shape_id];
        if (lsh.light_id < 0) return;
        VertexCtx c = load_vertex(sc, v, p);
        V3 pos = c.sp.position;
        Surf lp = sample_tri(lsh, pk.tri_id, ld.uv);
        V3 dir = lp.position - pos;
        double d2 = len_sq(dir);
        V3 wo = dir / sqrt(d2);
        const LightD &l = sc.lights[lsh.light_id];
        // light facing away / bsdf_eval's geometric early-outs: value and every adjoint below are exactly zero then.  (The
        // forward pass has folded the same test into the occlusion byte and the list this stage runs over holds only lanes
        // whose byte is clear: kept for callers that hand over the full list.)
        if (nee_is_geometrically_zero(sc, c, pk, lp)) return;
        const GMaterial &gm = g.materials[c.shape->material_id];
        V3 thr = ld3(v.thr, v.n, p, 0);
        V3 pc_bar = a.weight * image_grad(a.d_image, a.nd, a.radiance_dim, p);
        V3 thr_bar = v3(0), in_dir_bar = v3(0);
        Surf sp_bar = surf_zero();
        V3 f = bsdf_eval(*c.mat, c.sp, c.wi, wo, c.mrough);
        double cl = dot(wo, lp.geom_normal);
        double geo = fabs(cl) / d2;
        V3 Le = v3f(l.intensity);
        double pdf_nee = sc.light_pmf[lsh.light_id] * (1 / sc.light_areas[lsh.light_id]);
        double pdf_b = bsdf_pdf(*c.mat, c.sp, c.wi, wo, c.mrough) * geo;
        double mis = 1 / (1 + sq(pdf_b / pdf_nee));
        V3 nee = (mis * geo / pdf_nee) * f * Le;
        V3 nee_bar = pc_bar * thr;
        thr_bar += pc_bar * nee;
        double w = mis / pdf_nee;
        double w_bar = geo * sum(nee_bar * f * Le);
        double pdfnee_bar = -w_bar * w / pdf_nee;
        double geo_bar = w * sum(nee_bar * f * Le);
        V3 f_bar = w * nee_bar * geo * Le;
        V3 Le_bar = w * nee_bar * geo * f;
        // pdf_nee depends on the sampled triangle's area
        double area_bar = -pdfnee_bar * pdf_nee / tri_area(lsh, pk.tri_id);
        adj_tri_area(lsh, pk.tri_id, area_bar, lv_bar);
        if (g.light_intensity) accum3(g.light_intensity + 3 * lsh.light_id, Le_bar);
        double cl_bar = cl > 0 ? geo_bar / d2 : -geo_bar / d2;
        double d2_bar = -geo_bar * geo / d2;
        V3 wo_bar = cl_bar * lp.geom_normal;
        Surf lp_bar = surf_zero();
        lp_bar.geom_normal = cl_bar * wo;
        V3 wi_bar = v3(0);
        adj_bsdf_eval(*c.mat, c.sp, c.wi, wo, c.mrough, f_bar, gm, sp_bar, wi_bar, wo_bar);
        V3 dir_bar = wo_bar / sqrt(d2);
        double sd_bar = -sum(wo_bar * dir) / d2;
        d2_bar += (0.5f * sd_bar / sqrt(d2));
        dir_bar += adj_len_sq(dir, d2_bar);
        lp_bar.position += dir_bar;
        sp_bar.position -= dir_bar;
        in_dir_bar -= wi_bar;
        adj_sample_tri(lsh, pk.tri_id, ld.uv, lp_bar, lv_bar);
        l_shape = pk.shape_id; l_tri = pk.tri_id;
        adj_record_add(adj, p, thr_bar, in_dir_bar, sp_bar);
    }
};

// ---- adjoint of the camera vertex ------------------------------------------------------------------
// Adjoint of the non-radiance G-buffer channels at the first hit: adds to the shading-point adjoint,
// the ray-origin adjoint (depth) and the texture gradients (src/primary_contribution.cpp:487-700).
RDR_FN void adj_first_hit_channels(const SceneD &sc, const GScene &g, const ChannelsD &ch, const float *d_image,
                                   double weight, int p, int shape, const Surf &sp, const Ray &ray,
                                   Surf &pt_bar, V3 &org_bar) {
    const ShapeD &sh = sc.shapes[shape];
    const MaterialD &m = sc.materials[sh.material_id];
    const GMaterial &gm = g.materials[sh.material_id];
    const float *gi = d_image + (size_t)ch.nd * p;
    int d = 0;
    for (int k = 0; k < ch.n; ++k) {
        int id = ch.id[k];
        int width = channel_width(id, ch.max_generic);
        switch (id) {
            case 2: {     // depth = |position - org|
                double dist_bar = (double)gi[d] * weight;
                V3 diff_bar = adj_len(sp.position - ray.org, dist_bar);
                org_bar -= diff_bar; pt_bar.position += diff_bar;
            } break;
            case 3: pt_bar.position += weight * V3{(double)gi[d], (double)gi[d + 1], (double)gi[d + 2]}; break;
            case 4: pt_bar.geom_normal += weight * V3{(double)gi[d], (double)gi[d + 1], (double)gi[d + 2]}; break;
            case 5: pt_bar.frame.n += weight * V3{(double)gi[d], (double)gi[d + 1], (double)gi[d + 2]}; break;
            case 6: pt_bar.uv += weight * V2{(double)gi[d], (double)gi[d + 1]}; break;
            case 7: pt_bar.bary += weight * V2{(double)gi[d], (double)gi[d + 1]}; break;
            case 8: {
                V3 r_bar = weight * V3{(double)gi[d], (double)gi[d + 1], (double)gi[d + 2]};
                if (m.use_vertex_color) pt_bar.color += r_bar; else adj_tex3(m.diffuse, sp, r_bar, gm.diffuse, pt_bar);
            } break;
            case 9: adj_tex3(m.specular, sp, weight * V3{(double)gi[d], (double)gi[d + 1], (double)gi[d + 2]}, gm.specular, pt_bar); break;
            case 10: adj_tex1(m.roughness, sp, weight * (double)gi[d], gm.roughness, pt_bar); break;
            case 11: {
                if (m.generic.num_levels > 0) {
                    double ob[kMaxGeneric];
                    for (int j = 0; j < m.generic.channels; ++j) ob[j] = weight * (double)gi[d + j];
                    adj_tex_fetch(m.generic, sp.uv, sp.du_dxy, sp.dv_dxy, ob, gm.generic, pt_bar.uv, pt_bar.du_dxy, pt_bar.dv_dxy);
                }
            } break;
            case 12: pt_bar.color += weight * V3{(double)gi[d], (double)gi[d + 1], (double)gi[d + 2]}; break;
            default: break;      // radiance handled by the caller; alpha and ids carry no gradient
        }
        d += width;
    }
}

struct AdjPrimary {
    static constexpr int kMinBlocksPerCU = 2;
    static constexpr int kMidBlocksPerCU = 2;
    SceneD sc; GScene g; SamplerD rng; int sample_center;
    VSlice v0; const float *d_image; int nd, radiance_dim; double weight;
    AdjState adj; float *screen_grad; ChannelsD ch;
    RDR_FN void make_lean() { lean_scene(sc); lean_slice(v0); lean_channels(ch); nd = 3; radiance_dim = 0; adj.plain = 1; }
    RDR_FN void make_mid() { mid_scene(sc); lean_channels(ch); nd = 3; radiance_dim = 0; }
    RDR_FN void operator()(int p) const {
        TriGrad tg = trigrad_zero();          // gradient of the first-hit triangle, scattered by the whole wave at the end
        int tg_shape = -1, tg_tri = -1;
        RDR_INLINE_CALL camera_vertex(p, tg, tg_shape, tg_tri);
        scatter_trigrad_wave(sc.shapes, g.shapes, tg_shape, tg_tri, tg, sc.plain_materials != 0);
    }
    RDR_FN void camera_vertex(int p, TriGrad &tg, int &tg_shape, int &tg_tri) const {
        int shape = v0.shape[p];
        Ray ray = load_ray(v0, p);
        RayDiff rd = load_rdiff(v0, p);
        // radiance channel: only the light intensity receives a gradient here
        if (shape >= 0 && radiance_dim >= 0) {
            const ShapeD &sh = sc.shapes[shape];
            if (sh.light_id >= 0 && g.light_intensity) {
                RayDiff tmp;
                Surf sp = surf_at(sh, v0.tri[p], ray, rd, tmp, !sc.no_diffs);
                const LightD &l = sc.lights[sh.light_id];
                if (dot(-ray.dir, sp.frame.n) > 0 && l.directly_visible) {
                    V3 e_bar = weight * ld3(v0.thr, v0.n, p, 0) * image_grad(d_image, nd, ch.radiance_off, p);
                    accum3(g.light_intensity + 3 * sh.light_id, e_bar);
                }
            }
        }
        if (shape < 0) {
            // a miss only carries an adjoint when it looks at the environment light (primary_contribution.cpp:469-485);
            // the ray-differential adjoint of the lookup is dropped there (primary_intersection.cpp consumes it on hits only)
            if (sc.envmap == nullptr || !sc.envmap->directly_visible || radiance_dim < 0 || len_sq(ray.dir) <= 0) return;
            V3 e_bar = weight * ld3(v0.thr, v0.n, p, 0) * image_grad(d_image, nd, ch.radiance_off, p);
            DRay mr_bar = dray_zero();
            RayDiff mrd_bar = raydiff_zero();
            adj_envmap_eval(*sc.envmap, ray.dir, rd, e_bar, g.envmap, mr_bar.dir, mrd_bar);
            V2 ms = v2(0.5, 0.5);
            if (!sample_center) { const SamplerD::Lane ln = rng.lane(p); ms = v2(rng.draw(ln, 0), rng.draw(ln, 1)); }
            V2 mscr_bar = v2(0, 0);
            RDR_INLINE_CALL adj_primary_ray(sc.cam, pixel_to_screen(sc.cam, p, ms), mr_bar, g.cam, screen_grad != nullptr, mscr_bar);
            if (screen_grad) { screen_grad[2 * p] += (float)mscr_bar.x; screen_grad[2 * p + 1] += (float)mscr_bar.y; }
            return;
        }
        DRay r_bar = dray_zero();
        r_bar.dir = ld3(adj.ray_dir, adj.n, p, 0);
        RayDiff prd_bar = raydiff_zero();
        {
            Surf pt_bar = load_adj_point(adj, p);
            if (!ch.radiance_only) {
                RayDiff tmp;
                Surf sp = surf_at(sc.shapes[shape], v0.tri[p], ray, rd, tmp, !sc.no_diffs);
                adj_first_hit_channels(sc, g, ch, d_image, weight, p, shape, sp, ray, pt_bar, r_bar.org);
            }
            adj_surf_at(sc.shapes[shape], v0.tri[p], ray, rd, pt_bar, raydiff_zero(), r_bar, prd_bar, tg, !sc.no_diffs, sc.plain_materials != 0);
            tg_shape = shape; tg_tri = v0.tri[p];
        }
        V2 s = v2(0.5, 0.5);
        if (!sample_center) { const SamplerD::Lane ln = rng.lane(p); s = v2(rng.draw(ln, 0), rng.draw(ln, 1)); }
        V2 screen = pixel_to_screen(sc.cam, p, s);
        double delta = 1e-3;
        double sx = 0.5 / sc.cam.width, sy = 0.5 / sc.cam.height;
        V2 scr_bar = v2(0, 0);
        const bool sb = screen_grad != nullptr;
        if (!sc.no_diffs) {
            // the finite-difference ray differential: two more primary rays carry its adjoint
            DRay rx_bar{prd_bar.org_dx * sx / delta, prd_bar.dir_dx * sx / delta};
            DRay ry_bar{prd_bar.org_dy * sy / delta, prd_bar.dir_dy * sy / delta};
            r_bar.org += (prd_bar.org_dx * -sx + prd_bar.org_dy * -sy) / delta;
            r_bar.dir += (prd_bar.dir_dx * -sx + prd_bar.dir_dy * -sy) / delta;
            RDR_INLINE_CALL adj_primary_ray(sc.cam, screen + v2(delta, 0), rx_bar, g.cam, sb, scr_bar);
            RDR_INLINE_CALL adj_primary_ray(sc.cam, screen + v2(0, delta), ry_bar, g.cam, sb, scr_bar);
        }
        RDR_INLINE_CALL adj_primary_ray(sc.cam, screen, r_bar, g.cam, sb, scr_bar);
        if (screen_grad) {
            screen_grad[2 * p] += (float)scr_bar.x;
            screen_grad[2 * p + 1] += (float)scr_bar.y;
        }
    }
};

// The adjoint stages restricted to the lanes marked in `live` (the replay of recorded stale reads, stages_edge.h: HitPosView):
// run with weight 0 they carry an adjoint that was put into the records down to the camera and into the gradient buffers --
// the stages are linear in (adjoint record, upstream image gradient) -- and do nothing for the other lanes.
struct AdjBounceScatterLive {
    static constexpr int kMidBlocksPerCU = AdjBounceScatter::kMidBlocksPerCU, kMinBlocksPerCU = AdjBounceScatter::kMinBlocksPerCU,
                         kLeanBlocksPerCU = AdjBounceScatter::kLeanBlocksPerCU;
    AdjBounceScatter f; const unsigned char *live;
    RDR_FN void make_lean() { f.make_lean(); }
    RDR_FN void make_mid() { f.make_mid(); }
    RDR_FN void operator()(int idx) const { if (!live[f.a.active[idx]]) return; RDR_INLINE_CALL f(idx); }
};
struct AdjPrimaryLive {
    static constexpr int kMidBlocksPerCU = AdjPrimary::kMidBlocksPerCU, kMinBlocksPerCU = AdjPrimary::kMinBlocksPerCU;
    AdjPrimary f; const unsigned char *live;
    RDR_FN void make_lean() { f.make_lean(); }
    RDR_FN void make_mid() { f.make_mid(); }
    RDR_FN void operator()(int p) const { if (!live[p]) return; RDR_INLINE_CALL f(p); }
};

// fp64 accumulators -> the caller's fp32 gradient tensors (+=), all tensors in one launch: lane i owns element i of the
// replicated block, sums its replicas in fixed order and finds the tensor it belongs to in the sorted segment table.
struct KeepNeeLive {    // lanes whose next-event estimate has something to differentiate (the slice's occlusion byte is clear)
    const unsigned char *occl;
    RDR_FN bool operator()(int p) const { return occl[p] == 0; }
};

struct KeepLitContinuation {    // lanes whose continuation ray reached an emitter (or, without a hit, the environment light)
    const int *next_shape; const ShapeD *shapes; bool envmap;
    RDR_FN bool operator()(int p) const { const int s = next_shape[p]; return s >= 0 ? shapes[s].light_id >= 0 : envmap; }
};
struct KeepCarryingContinuation {    // lanes (with a continuation hit) whose successor left a non-zero record, or whose ray reached an emitter
    const unsigned char *carries; const int *next_shape; const ShapeD *shapes;
    RDR_FN bool operator()(int p) const { return carries[p] != 0 || shapes[next_shape[p]].light_id >= 0; }
};

struct FlushSegment { size_t begin, count; float *out; };      // elements [begin, begin + count) of the block
struct FlushGrad {
    const double *block; size_t stride; int replicas;
    const FlushSegment *segments; int num_segments;
    RDR_FN void operator()(int i) const {
        int lo = 0, hi = num_segments;                 // last segment with begin <= i
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (segments[mid].begin <= (size_t)i) lo = mid; else hi = mid;
        }
        const FlushSegment sg = segments[lo];
        if ((size_t)i < sg.begin || (size_t)i >= sg.begin + sg.count) return;       // padding between tensors
        double s = 0;
        for (int r = 0; r < replicas; ++r) s += block[(size_t)r * stride + i];
        sg.out[(size_t)i - sg.begin] += (float)s;
    }
};

} // namespace rdr
