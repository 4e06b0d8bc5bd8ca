// stages_edge.h -- the edge-sampling estimators (visibility gradients).
//
// Behavioural spec:
//   SamplePrimaryEdges        <- primary_edge_sampler               src/edge.cpp:385-625
//   PrimaryEdgeDerivatives    <- primary_edge_derivatives_computer  src/edge.cpp:700-783
//   SampleSecondaryEdges      <- secondary_edge_sampler             src/edge.cpp:826-1773
//        (ltc_bound :838-883, importance :885-928, leaf importance :942-1083, hierarchical pick
//         sample_edge_h :1115-1237, NEE-billboard pick sample_edge_l :1239-1364)
//   SecondaryEdgeWeights      <- secondary_edge_weights_updater     src/edge.cpp:1856-1981
//   SecondaryEdgeDerivatives  <- secondary_edge_derivatives_accumulator  src/edge.cpp:2001-2053
// Edge sub-paths live in lanes 2*slot (upper side) and 2*slot+1 (lower side) of a pair of
// ping-pong vertex slices; both lanes of a slot draw the same random numbers.
#pragma once
#include "edges.h"
#include "stages_bwd.h"

namespace rdr {

// ---- small 3x3 helpers (cofactor inverse in the reference's operation order, src/matrix.h:222-244)
RDR_FN M3 m3_from_frame(const Frame &f) {
    M3 r;
    r.m[0][0] = f.x.x; r.m[0][1] = f.x.y; r.m[0][2] = f.x.z;
    r.m[1][0] = f.y.x; r.m[1][1] = f.y.y; r.m[1][2] = f.y.z;
    r.m[2][0] = f.n.x; r.m[2][1] = f.n.y; r.m[2][2] = f.n.z;
    return r;
}
RDR_FN M3 m3_inverse(const M3 &a) {
    const double (*m)[3] = a.m;
    double det = m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                 m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    double id = 1 / det;
    M3 r;
    r.m[0][0] = (m[1][1] * m[2][2] - m[2][1] * m[1][2]) * id;
    r.m[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id;
    r.m[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
    r.m[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) * id;
    r.m[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id;
    r.m[1][2] = (m[1][0] * m[0][2] - m[0][0] * m[1][2]) * id;
    r.m[2][0] = (m[1][0] * m[2][1] - m[2][0] * m[1][1]) * id;
    r.m[2][1] = (m[2][0] * m[0][1] - m[0][0] * m[2][1]) * id;
    r.m[2][2] = (m[0][0] * m[1][1] - m[1][0] * m[0][1]) * id;
    return r;
}
RDR_FN M3 m3_mul(const M3 &a, const M3 &b) {
    M3 r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += a.m[i][k] * b.m[k][j];
        r.m[i][j] = s;
    }
    return r;
}
RDR_FN V3 m3_apply(const M3 &a, V3 v) {     // sums start from 0.f, accumulate left to right
    V3 r;
    r.x = ((0.0 + a.m[0][0] * v.x) + a.m[0][1] * v.y) + a.m[0][2] * v.z;
    r.y = ((0.0 + a.m[1][0] * v.x) + a.m[1][1] * v.y) + a.m[1][2] * v.z;
    r.z = ((0.0 + a.m[2][0] * v.x) + a.m[2][1] * v.y) + a.m[2][2] * v.z;
    return r;
}

struct PrimaryEdgeRec { EdgeD edge; V2 edge_pt; };
struct SecondaryEdgeRec { EdgeD edge; V3 edge_pt, mwt, sp_pos; int use_nee_ray, diffuse_or_glossy; };

// Emit queue records for the live lanes of an edge slice (optional per-lane tmin).
struct QueueRays {
    const int *active; VSlice v; const double *tmin; rt::RayRec *q;
    RDR_FN void operator()(int idx) const {
        int p = active[idx];
        Ray r = load_ray(v, p);
        if (tmin) r.tmin = tmin[p];
        put_ray(q, idx, r, len_sq(r.dir) <= 1e-3f);
    }
};
struct RecordHits {
    const int *active; VSlice v; const rt::HitRec *hits;
    RDR_FN void operator()(int idx) const {
        int p = active[idx];
        rt::HitRec h = hits[idx];
        v.shape[p] = h.shape; v.tri[p] = h.shape >= 0 ? h.prim : -1;
    }
};
// First-hit emission for lanes whose hit ids are already recorded.
struct ShadeRecorded {
    SceneD sc; const int *active; VSlice v; Sink sink;
    RDR_FN void make_lean() { lean_scene(sc); lean_slice(v); lean_channels(sink.ch); sink.multipliers = nullptr; }
    RDR_FN void make_mid() { mid_scene(sc); lean_channels(sink.ch); sink.multipliers = nullptr; }
    RDR_FN void operator()(int idx) const {
        int p = active[idx];
        shade_first_hit(sc, sink, v, p, v.shape[p], v.tri[p]);
    }
};
// Mirrors of the reference's in-place differential buffer (VSlice::erd): transfer onto the surface
// at a first hit (src/scene.cpp:585-591), and the per-lane read of the primary edge pass.
struct MirrorSurfDiff {
    SceneD sc; const int *active; VSlice v;
    RDR_FN void operator()(int idx) const {
        int p = active[idx];
        if (v.shape[p] < 0) return;
        RayDiff out;
        surf_at(sc.shapes[v.shape[p]], v.tri[p], load_ray(v, p), load_rdiff(v, p), out);
        st_erd(v, p, out);
    }
};
// `chain` (sample batches in chain mode, the sequential part): the lanes read the state the previous sample left -- entry
// p % lanes_per_sample of the chain buffer -- instead of their own sample's entry.
struct LoadLaneDiff {
    const int *active; VSlice v; const double *chain; int chain_n, lanes_per_sample;
    RDR_FN void operator()(int idx) const {
        int p = active[idx];
        store_rdiff(v, p, chain ? ld_rdiff(chain, chain_n, p % lanes_per_sample) : ld_erd(v, p));
    }
};
// Chain mode: after sample s of a batch, the state the next sample finds: entry l of the sample's copy where the sample
// wrote it, else what was there before.
struct ChainAdvance {
    double *chain; int chain_n; const double *erd; int erd_n; const unsigned char *touched; int sample;
    RDR_FN void operator()(int l) const {
        const int i = sample * chain_n + l;
        if (!touched[i]) return;
        for (int k = 0; k < 12; ++k) chain[(size_t)k * chain_n + l] = erd[(size_t)k * erd_n + i];
    }
};
// Chain mode: primary-edge lanes whose differential entry was written during this batch (by the sample's own secondary passes
// or by the primary sampler's slot-indexed write) / the others
struct KeepTouched { const unsigned char *touched; int want; RDR_FN bool operator()(int p) const { return (touched[p] != 0) == (want != 0); } };
// dst[i] = src[seg[s] + i] for the lanes of sample s; the count lands in *count_out
struct ExtractSegment {
    const int *src; const int *seg; int s; int *dst; int *count_out;
    RDR_FN void operator()(int i) const {
        const int b = seg[s], e = seg[s + 1];
        if (i == 0) *count_out = e - b;
        if (i < e - b) dst[i] = src[b + i];
    }
};
struct FillDouble { double *p; double value; RDR_FN void operator()(int i) const { p[i] = value; } };

// ===================================== primary edges ================================================
struct SamplePrimaryEdges {
    SceneD sc; EdgeSceneD es; SamplerD rng; int dim;
    const float *d_image; int nd, radiance_dim;
    PrimaryEdgeRec *recs; VSlice v;       // lanes 2*slot, 2*slot+1
    double *multipliers;                  // [2P x nd] per-channel weights of the two rays, or null
    int batch_P0 = 0;                     // sample batches: slots per sample (0: one sample)
    RDR_FN void make_lean() { lean_scene(sc); lean_slice(v); multipliers = nullptr; nd = 3; radiance_dim = 0; }
    RDR_FN void make_mid() { mid_scene(sc); multipliers = nullptr; nd = 3; radiance_dim = 0; }
    RDR_FN void operator()(int slot) const {
        int l0 = 2 * slot, l1 = 2 * slot + 1;
        if (multipliers) for (int d = 0; d < 2 * nd; ++d) multipliers[(size_t)nd * l0 + d] = 0;
        PrimaryEdgeRec rec;
        rec.edge = EdgeD{-1, 0, 0, 0, 0};
        rec.edge_pt = v2(0, 0);
        recs[slot] = rec;
        for (int l = l0; l <= l1; ++l) {
            st3(v.thr, v.n, l, 0, v3(0));
            store_ray(v, l, v3(0), v3(0));
            v.shape[l] = -1; v.tri[l] = -1;
            v.mrough[l] = 0;
        }
        const SamplerD::Lane ln = rng.lane(slot);
        double edge_sel = rng.draw(ln, dim), t = rng.draw(ln, dim + 1);
        int eid = iclamp(upper_bound_idx(es.primary_cdf, es.num_edges, edge_sel) - 1, 0, es.num_edges - 1);
        const EdgeD &edge = es.edges[eid];
        V3 a = edge_v0(sc.shapes, edge), b = edge_v1(sc.shapes, edge);
        V2 a_ss, b_ss;
        // both rays of a slot carry the differential of the sampled screen position; see DESIGN.md
        // "edge-ray differentials" for how this differs from the reference's slot-indexed buffer
        RayDiff rd = raydiff_zero();
        if (!project_segment(sc.cam, a, b, a_ss, b_ss) || es.primary_pmf[eid] <= 0.f) {
            store_rdiff(v, l0, raydiff_zero()); store_rdiff(v, l1, raydiff_zero());
            return;
        }
        const bool linear = linear_projection(sc.cam);
        V2 pt;
        V3 a_dir = v3(0), b_dir = v3(0), ab_dir = v3(0), pt3 = v3(0);
        if (linear) {
            pt = a_ss + t * (b_ss - a_ss);
        } else {
            // fisheye / panorama / distorted lenses: sample on the camera-space film, where the edge is straight (:486-512)
            a_dir = screen_to_camera(sc.cam, a_ss); b_dir = screen_to_camera(sc.cam, b_ss);
            ab_dir = b_dir - a_dir;
            pt3 = a_dir + t * ab_dir;
            pt = camera_to_screen(sc.cam, pt3);
        }
        if (!in_screen(sc.cam, pt)) {
            store_rdiff(v, l0, raydiff_zero()); store_rdiff(v, l1, raydiff_zero());
            return;
        }
        rec.edge = edge; rec.edge_pt = pt;
        recs[slot] = rec;
        Ray up, lo;
        double jacobian = 1;
        if (linear) {
            V2 dn = normalize(a_ss - b_ss);
            V2 hn = v2(dn.y, -dn.x);
            double off = 1e-6f;
            up = primary_ray(sc.cam, pt + hn * off); lo = primary_ray(sc.cam, pt - hn * off);
        } else {
            V3 hn = normalize(cross(a_dir, b_dir));
            V3 a_loc = xfm_point(sc.cam.world_to_cam, a), b_loc = xfm_point(sc.cam.world_to_cam, b);
            V3 e_loc = a_loc + t * b_loc;                                 // [quirk] not a point of the edge (:529)
            double off = 1e-5f / len(e_loc);
            up = primary_ray(sc.cam, camera_to_screen(sc.cam, normalize(pt3 + off * hn)));
            lo = primary_ray(sc.cam, camera_to_screen(sc.cam, normalize(pt3 - off * hn)));
            // |d alpha / d screen|^-1 and the stretch of the line parameterisation, by finite difference (:562-576)
            V2 pt_bar = v2(0, 0);
            adj_screen_to_camera(sc.cam, pt, cross(a_dir, b_dir), pt_bar);
            double dirac_jacobian = 1.f / sqrt(pt_bar.x * pt_bar.x + pt_bar.y * pt_bar.y);
            double jac_offset = 1e-6;
            V3 pt3_delta = a_dir + (t + jac_offset) * ab_dir;
            V2 pt_delta = camera_to_screen(sc.cam, pt3_delta);
            double line_jacobian = len((pt_delta - pt) / off);
            jacobian = line_jacobian * dirac_jacobian;
        }
        store_ray(v, l0, up.org, up.dir);
        store_ray(v, l1, lo.org, lo.dir);
        int vw = sc.cam.vp_x1 - sc.cam.vp_x0, vh = sc.cam.vp_y1 - sc.cam.vp_y0;
        int xi = iclamp(int(pt.x * sc.cam.width - sc.cam.vp_x0), 0, vw);
        int yi = iclamp(int(pt.y * sc.cam.height - sc.cam.vp_y0), 0, vh);
        // [quirk] the reference clamps to [0, vw] x [0, vh] INCLUSIVE (src/edge.cpp:447-450, 541-544): a point on the right
        // border reads the first pixel of the next row (reproduced: same linear index); one on the bottom border, or in the
        // bottom-right corner, reads past the image there -- here it reads the last pixel of its row / column instead
        int pix = yi * vw + xi;
        if (pix >= vw * vh) pix = (yi < vh ? yi : vh - 1) * vw + (xi < vw ? xi : vw - 1);
        V3 dc = radiance_dim >= 0 ? image_grad(d_image, nd, radiance_dim, pix) : v3(0);
        double pmf = es.primary_pmf[eid];
        if (multipliers) {
            for (int d = 0; d < nd; ++d) {
                double dch = d_image[(size_t)nd * pix + d];
                multipliers[(size_t)nd * l0 + d] = linear ? dch / pmf : dch * jacobian / pmf;
                multipliers[(size_t)nd * l1 + d] = linear ? -dch / pmf : -dch * jacobian / pmf;
            }
        }
        V3 w_up = dc / pmf, w_lo = -dc / pmf;
        if (!linear) { w_up = w_up * jacobian; w_lo = w_lo * jacobian; }
        st3(v.thr, v.n, l0, 0, w_up);
        st3(v.thr, v.n, l1, 0, w_lo);
        if (!sc.no_diffs) primary_ray_with_diff(sc.cam, pt, rd);
        // [quirk] slot-indexed; lanes read theirs in LoadLaneDiff.  (A batch: slot s * P0 + k is entry k of sample s's copy.)
        if (v.erd) st_erd(v, batch_P0 > 0 ? slot + (slot / batch_P0) * batch_P0 : slot, rd);
        else { store_rdiff(v, l0, rd); store_rdiff(v, l1, rd); }
    }
};

struct PrimaryEdgeDerivatives {
    SceneD sc; GScene g; const PrimaryEdgeRec *recs; const double *edge_contrib; float *screen_grad;
    RDR_FN void make_lean() { lean_scene(sc); }
    RDR_FN void make_mid() { mid_scene(sc); }
    RDR_FN void operator()(int slot) const {
        const PrimaryEdgeRec &rec = recs[slot];
        if (rec.edge.shape_id < 0) return;
        double contrib = edge_contrib[2 * slot] + edge_contrib[2 * slot + 1];
        if (contrib == 0) return;                  // every term below is a product with it
        V3 a = edge_v0(sc.shapes, rec.edge), b = edge_v1(sc.shapes, rec.edge);
        V2 a_ss, b_ss;
        if (!project_segment(sc.cam, a, b, a_ss, b_ss)) return;
        V2 pt = rec.edge_pt;
        V2 a_bar = v2(0, 0), b_bar = v2(0, 0), pt_bar = v2(0, 0);
        if (linear_projection(sc.cam)) {
            // Eq. 8 of the paper: gradient of the edge equation w.r.t. its screen-space end points
            a_bar = v2(b_ss.y - pt.y, pt.x - b_ss.x);
            b_bar = v2(pt.y - a_ss.y, a_ss.x - pt.x);
            pt_bar = v2(a_ss.y - b_ss.y, b_ss.x - a_ss.x);
        } else {
            // alpha(p) = dot(p, cross(a_dir, b_dir)) on the camera-space film (:734-749)
            V3 a_dir = screen_to_camera(sc.cam, a_ss), b_dir = screen_to_camera(sc.cam, b_ss);
            V3 e_dir = screen_to_camera(sc.cam, pt);
            adj_screen_to_camera(sc.cam, a_ss, cross(b_dir, e_dir), a_bar);
            adj_screen_to_camera(sc.cam, b_ss, cross(e_dir, a_dir), b_bar);
            adj_screen_to_camera(sc.cam, b_ss, cross(a_dir, b_dir), pt_bar);     // [quirk] evaluated at b_ss, not at pt (:748)
        }
        a_bar = a_bar * contrib; b_bar = b_bar * contrib; pt_bar = pt_bar * contrib;
        V3 pa_bar = v3(0), pb_bar = v3(0);
        adj_project_segment(sc.cam, a, b, a_bar, b_bar, g.cam, pa_bar, pb_bar);
        double *gv = g.shapes[rec.edge.shape_id].vertices;
        accum3(gv + 3 * rec.edge.v0, pa_bar);
        accum3(gv + 3 * rec.edge.v1, pb_bar);
        if (screen_grad) {
            int vw = sc.cam.vp_x1 - sc.cam.vp_x0, vh = sc.cam.vp_y1 - sc.cam.vp_y0;
            // the reference clamps to [0, vw] x [0, vh] inclusive and writes past the image for a point on the right /
            // bottom border (src/edge.cpp:765-773); clamped to the last pixel here instead of reproducing the overflow
            int xi = iclamp(int(pt.x * sc.cam.width - sc.cam.vp_x0), 0, vw - 1);
            int yi = iclamp(int(pt.y * sc.cam.height - sc.cam.vp_y0), 0, vh - 1);
            accum_f32(screen_grad + 2 * (yi * vw + xi), (float)pt_bar.x);
            accum_f32(screen_grad + 2 * (yi * vw + xi) + 1, (float)pt_bar.y);
        }
    }
};

// ==================================== secondary edges ===============================================
struct LtcCtx { V3 pos; M3 m, m_inv; };

RDR_FN bool box_contains(V3 lo, V3 hi, V3 p) {
    return p.x >= lo.x && p.x <= hi.x && p.y >= lo.y && p.y <= hi.y && p.z >= lo.z && p.z <= hi.z;
}
RDR_FN double min_abs_bound(double lo, double hi) {
    if (lo <= 0.f && hi >= 0.f) return 0;
    if (lo <= 0.f && hi <= 0.f) return hi;
    return lo;
}
// Upper bound of the LTC-transformed clamped cosine over a box (src/edge.cpp:838-876).
RDR_FN double ltc_bound(V3 lo, V3 hi, const LtcCtx &c) {
    V3 dir = V3{0, 0, 1};
    if (!box_contains(lo, hi, c.pos)) {
        // Bounds of the eight transformed corners m_inv (corner - pos).  A coordinate of a transformed corner is
        // m3_apply's left-to-right sum of three products, one per box axis, each with the low or the high face; every
        // rounding in that sum is monotone in its operands and the corners are all eight combinations, so the
        // minimum (maximum) over the corners is the same sum of the per-axis minima (maxima) -- the identical
        // double, a third of the arithmetic of transforming eight corners.
        const V3 fl = lo - c.pos, fh = hi - c.pos;
        double blv[3], bhv[3];
        _Pragma("unroll") for (int r = 0; r < 3; ++r) {
            const double xl = c.m_inv.m[r][0] * fl.x, xh = c.m_inv.m[r][0] * fh.x;
            const double yl = c.m_inv.m[r][1] * fl.y, yh = c.m_inv.m[r][1] * fh.y;
            const double zl = c.m_inv.m[r][2] * fl.z, zh = c.m_inv.m[r][2] * fh.z;
            blv[r] = ((0.0 + dmin(xl, xh)) + dmin(yl, yh)) + dmin(zl, zh);
            bhv[r] = ((0.0 + dmax(xl, xh)) + dmax(yl, yh)) + dmax(zl, zh);
        }
        const V3 bl = V3{blv[0], blv[1], blv[2]}, bh = V3{bhv[0], bhv[1], bhv[2]};
        if (bh.z < 0) return 0;
        dir.x = min_abs_bound(bl.x, bh.x);
        dir.y = min_abs_bound(bl.y, bh.y);
        dir.z = bh.z;
        double dl = len(dir);
        if (dl <= 0) dir = V3{0, 0, 1}; else dir = dir / dl;
    }
    V3 md = normalize(m3_apply(c.m, dir));
    V3 ml = m3_apply(c.m_inv, md);
    if (ml.z <= 0) return 0;
    double nrm = sq(len_sq(ml));
    return ml.z / nrm;
}
// Sphere/box overlap (Arvo), early-out form of src/aabb.h:158-174.
// Only the x axis can decide (see EdgeNodeC): the first partial sum is the smallest one.
// Silhouette query of one point p against Hough intervals: centre x and squared radius of the sphere
// (0.5 (p - cam), 0.5 |cam - p|), computed once per walk instead of at every node.
struct SilQuery { double cx, r2; };
RDR_FN SilQuery sil_query(const EdgeSceneD &es, V3 p) {
    return SilQuery{(0.5f * (p - es.cam_org)).x, sq(0.5f * len(es.cam_org - p))};
}
RDR_FN bool sphere_box_x(const SilQuery &q, double lo_x, double hi_x) {
    double dmin_ = 0;
    if (q.cx < lo_x) dmin_ += sq(q.cx - lo_x);
    else if (q.cx > hi_x) dmin_ += sq(q.cx - hi_x);
    return dmin_ <= q.r2;
}
RDR_FN bool sphere_box_x(double cx, double radius, double lo_x, double hi_x) {
    double dmin_ = 0, r2 = sq(radius);
    if (cx < lo_x) dmin_ += sq(cx - lo_x);
    else if (cx > hi_x) dmin_ += sq(cx - hi_x);
    return dmin_ <= r2;
}
// Can the subtree with Hough x-interval [dx_min, dx_max] hold an edge that is a silhouette seen from p?
// (3-D tree: always; src/edge.cpp contains_silhouette)
RDR_FN bool may_hold_silhouette(bool tree3d, double dx_min, double dx_max, const SilQuery &q) {
    if (tree3d) return true;
    return sphere_box_x(q, dx_min, dx_max);
}
// Importance of child k of `nd` (src/edge.cpp:885-928).
RDR_FN double node_importance(const EdgeNodeP &nd, int k, bool tree3d, const LtcCtx &c, const SilQuery &q) {
    if (!tree3d) {
        if (!sphere_box_x(q, nd.c_dx_min[k], nd.c_dx_max[k])) return 0;
    }
    V3 lo = v3_of(nd.c_pmin[k]), hi = v3_of(nd.c_pmax[k]);
    double brdf = ltc_bound(lo, hi, c);
    V3 ctr = 0.5f * (lo + hi);
    return brdf * nd.c_wlen[k] / dmax(len(c.pos - ctr), 1e-3);
}

// LTC line integral of an edge seen from the shading point (clipped to the tangent plane).
struct LineSetup { bool ok; V3 wt, vo; double d, l0, l1; };
RDR_FN double line_I(const LineSetup &s, double l) {
    double d = s.d;
    return (l / (d * (d * d + l * l)) + atan(l / d) / (d * d)) * s.vo.z + (l * l / (d * (d * d + l * l))) * s.wt.z;
}
RDR_FN LineSetup line_setup(V3 v0o, V3 v1o) {
    LineSetup s;
    s.ok = false;
    if (v0o.z < 0.f) v0o = (v0o * v1o.z - v1o * v0o.z) / (v1o.z - v0o.z);
    if (v1o.z < 0.f) v1o = (v0o * v1o.z - v1o * v0o.z) / (v1o.z - v0o.z);
    V3 dirv = v1o - v0o;
    s.wt = normalize(dirv);
    s.l0 = dot(v0o, s.wt);
    s.l1 = dot(v1o, s.wt);
    s.vo = v0o - s.l0 * s.wt;
    s.d = len(s.vo);
    s.ok = true;
    return s;
}
RDR_FN double edge_line_importance(V3 a, V3 b, const LtcCtx &c) {
    if (len_sq(b - a) > 1e-10f) {
        V3 ao = m3_apply(c.m_inv, a - c.pos), bo = m3_apply(c.m_inv, b - c.pos);
        if (ao.z > 0.f || bo.z > 0.f) {
            LineSetup s = line_setup(ao, bo);
            return dmax(line_I(s, s.l1) - line_I(s, s.l0), 0.0);
        }
    }
    return 0;
}
RDR_FN double leaf_importance_h(const SceneD &sc, const EdgeSceneD &es, int eid, const LtcCtx &c) {
    const EdgeGeom &g = es.geom[eid];
    if (!edge_is_silhouette_g(g, c.pos)) return 0;
    return edge_line_importance(v3_of(g.v0), v3_of(g.v1), c);
}
// Leaf weight for the NEE-billboard mode: the edge must be a silhouette from both ends of the NEE
// segment and the NEE ray must pass within `edge_bounds_expand` of it.
RDR_FN double leaf_importance_l(const SceneD &sc, const EdgeSceneD &es, int eid, const LtcCtx &c,
                                const Ray &nee, bool nee_valid) {
    const EdgeGeom &g = es.geom[eid];
    if (!edge_is_silhouette_g(g, c.pos)) return 0;
    if (nee_valid) {
        if (!edge_is_silhouette_g(g, nee.org + nee.tmax * nee.dir)) return 0;
    } else {
        if (!edge_is_silhouette_g(g, nee.dir)) return 0;
    }
    V3 a = v3_of(g.v0), b = v3_of(g.v1);
    V3 pn = nee.dir;
    double t = -(dot(nee.org, pn) - dot(a, pn)) / dot(nee.dir, pn);
    V3 ip = nee.org + nee.dir * t;
    V3 ap = a - ip;
    V3 ab = normalize(b - a);
    V3 ept = ip + ap - (dot(ap, ab)) * ab;
    if (len_sq(ip - ept) > sq(es.edge_bounds_expand)) return 0;
    return edge_line_importance(a, b, c);
}

// leaf_importance_l on a GatherLeaf record: the same four conditions and the same value, with the cheapest and most
// selective condition first -- the distance of the NEE segment to the edge (needs only the end points, rejects 61 % of
// the candidates on the benchmark scene), then the two silhouette tests (79 % / 10 % in the reference's order), and the LTC
// line integral only for what is left (2.5 %).  The function is pure, so the order of the tests cannot change its result.
RDR_FN double leaf_importance_gathered(const GatherLeaf &gl, double expand, const LtcCtx &c, const Ray &nee, bool nee_valid) {
    const V3 a = v3_of(gl.v0), b = v3_of(gl.v1);
    {
        V3 pn = nee.dir;
        double t = -(dot(nee.org, pn) - dot(a, pn)) / dot(nee.dir, pn);
        V3 ip = nee.org + nee.dir * t;
        V3 ap = a - ip;
        V3 ab = normalize(b - a);
        V3 ept = ip + ap - (dot(ap, ab)) * ab;
        if (len_sq(ip - ept) > sq(expand)) return 0;
    }
    EdgeGeom g;
    for (int k = 0; k < 3; ++k) { g.v0[k] = gl.v0[k]; g.v1[k] = gl.v1[k]; g.o0[k] = gl.o0[k]; g.o1[k] = gl.o1[k]; }
    g.f0 = gl.f0; g.f1 = gl.f1; g.has_normals = gl.has_normals; g.pad = 0;
    if (!edge_is_silhouette_g(g, c.pos)) return 0;
    if (nee_valid) {
        if (!edge_is_silhouette_g(g, nee.org + nee.tmax * nee.dir)) return 0;
    } else {
        if (!edge_is_silhouette_g(g, nee.dir)) return 0;
    }
    return edge_line_importance(a, b, c);
}

// Slab test of the reference's edge-tree traversal (src/aabb.h:176-200), boxes grown by `expand`.
// `inv_dir` = 1 / r.dir per axis, computed once per walk (the reference divides at every node; same values).
RDR_FN bool ray_box_expand(V3 lo, V3 hi, const Ray &r, V3 inv_dir, double expand) {
    double t0 = r.tmin, t1 = r.tmax;
    for (int i = 0; i < 3; ++i) {
        double inv = comp(inv_dir, i);
        double tn = (comp(lo, i) - expand - comp(r.org, i)) * inv;
        double tf = (comp(hi, i) + expand - comp(r.org, i)) * inv;
        if (tn > tf) { double t = tn; tn = tf; tf = t; }
        tf *= (1 + 1e-6f);
        t0 = tn > t0 ? tn : t0;
        t1 = tf < t1 ? tf : t1;
        if (t0 > t1) return false;
    }
    return true;
}

constexpr int kHSamples = 16;
// hierarchical pick: every stack entry carries >= 1 of the 16 samples, so <= 16 entries are live
// (three LDS columns per lane -- int, byte, double -- 13 B per entry: 3 workgroups fit in a CU's 160 KB)
constexpr int kHStack = 16;
// NEE pick: depth-first with both children pushed: <= tree depth + 1 entries.  The kernel is instantiated for a
// few stack sizes and the host picks the smallest that covers the scene's trees (EdgeSceneD::max_stack).
constexpr int kNStackMax = 64;

struct HItem { int ref, num; double pmf; };

// Stochastic top-down pick of one edge, splitting kHSamples "samples" by LTC-bounded importance
// and keeping one leaf by reservoir sampling.  Returns the edge id or -1; weight = 1/pmf.
RDR_DEV_FN int pick_edge_hierarchical(const SceneD &sc, const EdgeSceneD &es, const LtcCtx &c,
                                      double sample, double resample, double &weight) {
    const SilQuery q_pos = sil_query(es, c.pos);
    RDR_STACK_DECL(int, st_ref, kHStack);
    RDR_STACK_DECL(unsigned char, st_num, kHStack);
    RDR_STACK_DECL(double, st_pmf, kHStack);
#define RDR_H_PUSH(r, n, p) { RDR_STACK_AT(st_ref, sp) = (r); RDR_STACK_AT(st_num, sp) = (unsigned char)(n); RDR_STACK_AT(st_pmf, sp) = (p); sp++; }
    int sp = 0;
    int selected = -1;
    double edge_w = 0, wsum = 0;
    double imp_cs = es.cs_root != kNoEdgeTree ? 1.0 : 0.0, imp_ncs = es.ncs_root != kNoEdgeTree ? 1.0 : 0.0;
    if (imp_cs <= 0 && imp_ncs <= 0) return -1;
    double prob_cs = imp_cs / (imp_cs + imp_ncs), prob_ncs = 1 - prob_cs;
    double exp_cs = kHSamples * prob_cs, exp_ncs = kHSamples * prob_ncs;
    int n_cs = int(floor(exp_cs)), n_ncs = int(floor(exp_ncs));
    if (n_cs + n_ncs < kHSamples) {
        double prob = exp_cs - n_cs;
        if (sample < prob) { n_cs++; sample /= prob; }
        else { n_ncs++; sample = (sample - prob) / (1 - prob); }
    }
    if (n_cs > 0) RDR_H_PUSH(es.cs_root, n_cs, prob_cs)
    if (n_ncs > 0) RDR_H_PUSH(es.ncs_root, n_ncs, prob_ncs)
    while (sp > 0) {
        --sp;
        HItem it = HItem{RDR_STACK_AT(st_ref, sp), (int)RDR_STACK_AT(st_num, sp), RDR_STACK_AT(st_pmf, sp)};
        if (it.ref < 0) {
            const int leaf_edge = ~it.ref;
            double w = it.num * leaf_importance_h(sc, es, leaf_edge, c) / it.pmf;
            if (w > 0) {
                double prev = wsum;
                wsum += w;
                double nw = w / wsum;
                if (resample <= nw || prev == 0) {
                    selected = leaf_edge;
                    edge_w = w * it.pmf;
                    resample /= nw;
                } else {
                    resample = (resample - nw) / (1 - nw);
                }
            }
        } else {
            const EdgeNodeP &nd = edge_node(es, it.ref);
            const int tree = it.ref & kEdgeTreeBit;
            const bool tree3d = tree == 0;
            int c0 = nd.c_ref[0] < 0 ? nd.c_ref[0] : (nd.c_ref[0] | tree);
            int c1 = nd.c_ref[1] < 0 ? nd.c_ref[1] : (nd.c_ref[1] | tree);
            double i0, i1;
            if (box_contains(v3_of(nd.p_min), v3_of(nd.p_max), c.pos)) { i0 = i1 = 1; }
            else { i0 = node_importance(nd, 0, tree3d, c, q_pos); i1 = node_importance(nd, 1, tree3d, c, q_pos); }
            if (i0 > 0 || i1 > 0) {
                double p0 = i0 / (i0 + i1), p1 = 1 - p0;
                double e0 = it.num * p0, e1 = it.num * p1;
                int s0 = int(floor(e0)), s1 = int(floor(e1));
                if (s0 + s1 < it.num) {
                    double prob = e0 - s0;
                    if (sample < prob) { s0++; sample /= prob; }
                    else { s1++; sample = (sample - prob) / (1 - prob); }
                }
                if (s0 > 0 && sp < kHStack) RDR_H_PUSH(c0, s0, it.pmf * p0)
                if (s1 > 0 && sp < kHStack) RDR_H_PUSH(c1, s1, it.pmf * p1)
            }
        }
    }
    if (edge_w <= 0 || wsum <= 0) return -1;
    double pmf_h = edge_w * kHSamples / wsum;
    weight = 1 / pmf_h;
    return selected;
#undef RDR_H_PUSH
}

// The same pick with the leaf work deferred (SecEdgePickH2).  Interior steps consume only `sample`, leaf steps only
// `resample` and the reservoir, so the two sequences are independent: the descent records the leaves it pops, in pop
// order, and a second loop evaluates their importance and runs the reservoir -- the same operations on the same operands
// in the same order per variable.  On the GPU the lanes of a wave then run the (cheap, uniform) descent together and the
// (expensive: silhouette test + two LTC line integrals) leaf evaluations together instead of serialising the two bodies
// whenever a wave holds both kinds of entries (lane utilisation of the fused loop: 0.44, profiles/r1_pmc_sq.csv).
// The descent stack keeps kHStackLds entries per lane in LDS (13 B each: 4 workgroups per CU instead of 3) and spills the
// rare deeper entries to `spill`; the leaf list lives in HBM (written and read once, coalesced by entry index).
// One interior node = one 128-byte line.  Left to itself the compiler sinks each field's load into the branch that first
// uses it (the tests have early outs), which turns a step into ~9 dependent L1 round trips; this fetches the line with
// eight 16-byte loads issued together and pins them in registers.
RDR_DEV_FN EdgeNodeP load_node_line(const EdgeNodeP *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *q = reinterpret_cast<const u32x4 *>(p);
    u32x4 r0 = q[0], r1 = q[1], r2 = q[2], r3 = q[3], r4 = q[4], r5 = q[5], r6 = q[6], r7 = q[7];
    asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
    u32x4 all[8] = {r0, r1, r2, r3, r4, r5, r6, r7};
    EdgeNodeP out;
    __builtin_memcpy(&out, all, sizeof(out));
    return out;
#else
    return *p;
#endif
}
struct HLeaf { int ref, num; double pmf; };        // 16 B; ref: ~edge id for leaves, node reference for spilled stack entries
constexpr int kHStackLds = 12;
template <bool PRELOAD>
RDR_DEV_FN int pick_edge_hierarchical_deferred(const SceneD &sc, const EdgeSceneD &es, const LtcCtx &c, double sample, double resample,
                                               double &weight, HLeaf *leaves, HLeaf *spill, size_t stride) {
    const SilQuery q_pos = sil_query(es, c.pos);
    RDR_STACK_DECL(int, st_ref, kHStackLds);
    RDR_STACK_DECL(unsigned char, st_num, kHStackLds);
    RDR_STACK_DECL(double, st_pmf, kHStackLds);
    int sp = 0, nleaf = 0;
    auto push = [&](int r, int n, double p) {
        if (sp < kHStackLds) { RDR_STACK_AT(st_ref, sp) = r; RDR_STACK_AT(st_num, sp) = (unsigned char)n; RDR_STACK_AT(st_pmf, sp) = p; }
        else spill[(size_t)(sp - kHStackLds) * stride] = HLeaf{r, n, p};
        sp++;
    };
    double imp_cs = es.cs_root != kNoEdgeTree ? 1.0 : 0.0, imp_ncs = es.ncs_root != kNoEdgeTree ? 1.0 : 0.0;
    if (imp_cs <= 0 && imp_ncs <= 0) return -1;
    double prob_cs = imp_cs / (imp_cs + imp_ncs), prob_ncs = 1 - prob_cs;
    double exp_cs = kHSamples * prob_cs, exp_ncs = kHSamples * prob_ncs;
    int n_cs = int(floor(exp_cs)), n_ncs = int(floor(exp_ncs));
    if (n_cs + n_ncs < kHSamples) {
        double prob = exp_cs - n_cs;
        if (sample < prob) { n_cs++; sample /= prob; }
        else { n_ncs++; sample = (sample - prob) / (1 - prob); }
    }
    if (n_cs > 0) push(es.cs_root, n_cs, prob_cs);
    if (n_ncs > 0) push(es.ncs_root, n_ncs, prob_ncs);
    // every live entry (pending or recorded leaf) carries >= 1 of the kHSamples samples: sp + nleaf <= kHSamples
    while (sp > 0) {
        --sp;
        HItem it;
        if (sp < kHStackLds) it = HItem{RDR_STACK_AT(st_ref, sp), (int)RDR_STACK_AT(st_num, sp), RDR_STACK_AT(st_pmf, sp)};
        else { const HLeaf sl = spill[(size_t)(sp - kHStackLds) * stride]; it = HItem{sl.ref, sl.num, sl.pmf}; }
        if (it.ref < 0) {
            leaves[(size_t)nleaf * stride] = HLeaf{it.ref, it.num, it.pmf};
            nleaf++;
            continue;
        }
        const EdgeNodeP nd_line = PRELOAD ? load_node_line(&edge_node(es, it.ref)) : EdgeNodeP();
        const EdgeNodeP &nd = PRELOAD ? nd_line : edge_node(es, it.ref);
        const int tree = it.ref & kEdgeTreeBit;
        const bool tree3d = tree == 0;
        int c0 = nd.c_ref[0] < 0 ? nd.c_ref[0] : (nd.c_ref[0] | tree);
        int c1 = nd.c_ref[1] < 0 ? nd.c_ref[1] : (nd.c_ref[1] | tree);
        double i0, i1;
        if (box_contains(v3_of(nd.p_min), v3_of(nd.p_max), c.pos)) { i0 = i1 = 1; }
        else { i0 = node_importance(nd, 0, tree3d, c, q_pos); i1 = node_importance(nd, 1, tree3d, c, q_pos); }
        if (i0 > 0 || i1 > 0) {
            double p0 = i0 / (i0 + i1), p1 = 1 - p0;
            double e0 = it.num * p0, e1 = it.num * p1;
            int s0 = int(floor(e0)), s1 = int(floor(e1));
            if (s0 + s1 < it.num) {
                double prob = e0 - s0;
                if (sample < prob) { s0++; sample /= prob; }
                else { s1++; sample = (sample - prob) / (1 - prob); }
            }
            if (s0 > 0 && sp + nleaf < kHSamples) push(c0, s0, it.pmf * p0);
            if (s1 > 0 && sp + nleaf < kHSamples) push(c1, s1, it.pmf * p1);
        }
    }
    int selected = -1;
    double edge_w = 0, wsum = 0;
    for (int j = 0; j < nleaf; ++j) {
        const HLeaf it = leaves[(size_t)j * stride];
        const int leaf_edge = ~it.ref;
        double w = it.num * leaf_importance_h(sc, es, leaf_edge, c) / it.pmf;
        if (w > 0) {
            double prev = wsum;
            wsum += w;
            double nw = w / wsum;
            if (resample <= nw || prev == 0) {
                selected = leaf_edge;
                edge_w = w * it.pmf;
                resample /= nw;
            } else {
                resample = (resample - nw) / (1 - nw);
            }
        }
    }
    if (edge_w <= 0 || wsum <= 0) return -1;
    double pmf_h = edge_w * kHSamples / wsum;
    weight = 1 / pmf_h;
    return selected;
}

// Tail of the NEE-mode pick: pmf of the selected edge, Jacobian of the NEE-ray / billboard intersection, point on the edge.
RDR_FN int finish_edge_nee(const SceneD &sc, const EdgeSceneD &es, const Ray &nee, bool nee_valid, const Surf &nee_pt, int nee_shape,
                           int selected, double edge_w, double wsum, double &weight, V3 &edge_pt, V3 &mwt) {
    double pmf = edge_w / wsum;
    const EdgeD &e = es.edges[selected];
    V3 a = edge_v0(sc.shapes, e), b = edge_v1(sc.shapes, e);
    V3 pn = nee.dir;
    double t = -(dot(nee.org, pn) - dot(a, pn)) / dot(nee.dir, pn);
    if (t < nee.tmin || t > nee.tmax) return -1;
    V3 ip = nee.org + nee.dir * t;
    double jac = 0, pdf_nee = 0;
    if (nee_valid) {
        V3 ln = nee_pt.geom_normal, lpos = nee_pt.position;
        double tau = dot(lpos - nee.org, ln) / dot(ip - nee.org, ln);
        V3 omega = ip - nee.org;
        jac = len(tau * ((b - a) - omega * (dot(b - a, ln) / dot(omega, ln))));
        const ShapeD &lsh = sc.shapes[nee_shape];
        pdf_nee = sc.light_pmf[lsh.light_id] / sc.light_areas[lsh.light_id];
    } else {
        // environment light (:1349-1353)
        jac = 1 / len_sq(ip - nee.org);
        pdf_nee = envmap_pdf(*sc.envmap, nee.dir);
    }
    if (pmf <= 0 || jac <= 0 || pdf_nee <= 0) return -1;
    weight = 1 / (2 * es.edge_bounds_expand * pmf * jac * pdf_nee);
    V3 ap = a - ip;
    V3 ab = normalize(b - a);
    edge_pt = ip + ap - (dot(ap, ab)) * ab - nee.org;
    mwt = b - a;
    return selected;
}


RDR_FN M3 ltc_matrix(const float *tab, const Surf &sp, V3 wi, double roughness) {
    double ct = dot(wi, sp.frame.n);
    double theta = acos(ct);
    int rid = iclamp(int(roughness * (128 - 1)), 0, 128 - 1);
    int tid = iclamp(int((theta / (M_PI / 2.f)) * (128 - 1)), 0, 128 - 1);
    const float *p = tab + 9 * (rid + tid * 128);
    M3 r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = p[3 * i + j];
    return r;
}

// The secondary-edge sampler is cut into four stages so that lanes running the two very different
// edge-selection procedures (stochastic hierarchy descent vs. NEE-billboard gathering) do not share
// wavefronts: SecEdgeSetup decides the mode per slot, the slots are compacted per mode, SecEdgePickH /
// SecEdgePickN traverse the edge hierarchies, SecEdgeFinish places the point on the edge and emits
// the two rays.  Every stage rebuilds the small per-slot context (sec_prepare) from the stored path
// vertex instead of passing ~400 bytes per slot through HBM.
struct SecPick { int eid; double ew; V3 sample_p, mwt; };

struct SecPre {
    bool live;                 // false: slot produces no edge sample
    VertexCtx c;
    LtcCtx lc;
    double pd, ps, m_pmf, roughness, nee_pmf, edge_sel, resample_sel, bsdf_comp, t_sel;
    bool dg, use_nee, nee_valid;
    Ray nee; Surf nee_pt; int nee_shape;
};

RDR_FN SecPre sec_prepare(const SceneD &sc, const EdgeSceneD &es, const SamplerD &rng_main, int dim_main,
                          const SamplerD &rng_edge, int dim_edge, const VSlice &v, int p, int idx) {
    SecPre s;
    s.live = false;
    s.c = load_vertex(sc, v, p);
    if (s.c.mrough > 1e-2f) return s;
    // NEE segment of this vertex (recomputed from the forward sampler's numbers)
    LightDraw ld = draw_light(rng_main, p, dim_main);
    LightPick pk = pick_light(sc, ld.light_sel, ld.tri_sel);
    s.nee_valid = pk.shape_id >= 0;
    s.nee_shape = pk.shape_id;
    s.nee_pt = surf_zero();
    s.nee = make_ray(s.c.sp.position, v3(0));
    if (s.nee_valid) {
        s.nee_pt = sample_tri(sc.shapes[pk.shape_id], pk.tri_id, ld.uv);
        s.nee = shadow_ray_to(s.c.sp.position, s.nee_pt.position);
        s.nee.tmax = len(s.nee_pt.position - s.nee.org);
    } else if (sc.envmap != nullptr) {
        s.nee = make_ray(s.c.sp.position, envmap_sample(*sc.envmap, ld.uv));     // tmax = inf (:1381-1385)
    }
    const SamplerD::Lane eln = rng_edge.lane(idx);
    s.edge_sel = rng_edge.draw(eln, dim_edge); s.resample_sel = rng_edge.draw(eln, dim_edge + 1);
    s.bsdf_comp = rng_edge.draw(eln, dim_edge + 2); s.t_sel = rng_edge.draw(eln, dim_edge + 3);
    const MaterialD &mat = *s.c.mat;
    V3 kd = tex3(mat.diffuse, s.c.sp), ks = tex3(mat.specular, s.c.sp);
    double wd = luminance(kd), ws = luminance(ks), wsum = wd + ws;
    if (wsum <= 0.f) return s;
    s.pd = wd / wsum; s.ps = ws / wsum;
    V3 n = s.c.sp.frame.n;
    V3 wi = s.c.wi;
    if (mat.two_sided && dot(wi, n) < 0.f) n = -n;
    V3 fx = normalize(wi - n * dot(wi, n));
    V3 fy = cross(n, fx);
    if (dot(wi, n) > 1 - 1e-6f) onb(n, fx, fy);
    Frame iso{fx, fy, n};
    s.lc.pos = s.c.sp.position;
    s.roughness = dmax(tex1(mat.roughness, s.c.sp), s.c.mrough);
    if (s.bsdf_comp <= s.pd) {
        s.lc.m_inv = m3_from_frame(iso);
        s.lc.m = m3_inverse(s.lc.m_inv);
        s.m_pmf = s.pd;
    } else {
        s.lc.m_inv = m3_mul(m3_inverse(ltc_matrix(es.ltc, s.c.sp, wi, s.roughness)), m3_from_frame(iso));
        s.lc.m = m3_inverse(s.lc.m_inv);
        s.m_pmf = s.ps;
    }
    s.use_nee = false;
    s.nee_pmf = 1;
    s.dg = s.bsdf_comp <= s.pd || s.roughness > 0.1;
    if (s.dg) {
        s.use_nee = s.edge_sel < 0.5;
        if (s.roughness > 0.1) s.nee_pmf = 0.5f;
        else s.nee_pmf = s.use_nee ? s.pd * 0.5f : 1.f - s.pd * 0.5f;
    }
    if (!s.use_nee && s.dg) s.edge_sel = (s.edge_sel - 0.5) * 2;
    s.live = true;
    return s;
}

struct SecEdgeArgs {          // what every stage of the sampler needs
    SceneD sc; EdgeSceneD es;
    SamplerD rng_main; int dim_main;      // the forward sampler's light draw of this vertex
    SamplerD rng_edge; int dim_edge;      // edge sampler: 4 numbers per compacted slot
    const int *active; VSlice v;        // main-path vertex
};

// mode[slot]: 0 = no sample, 1 = hierarchical pick, 2 / 3 = NEE-billboard pick (3: the vertex lies on a shape with at least
// kDenseShapeTriangles triangles).  Also resets the slot's outputs.
constexpr int kDenseShapeTriangles = 512;
// What the slot set-up hands to the hierarchical pick (SecEdgePickHDescend / SecEdgePickHLeaves below): shading position, LTC
// matrix and the two random numbers of the pick -- sec_prepare's results, computed once here instead of again in the pick.
struct HDescent { double pos[3], m_inv[9], sample, resample; int nleaf, pad; };      // 120 B per slot
struct SecEdgeSetup {
    SecEdgeArgs a; unsigned char *mode; SecondaryEdgeRec *recs; SecPick *picks; VSlice ev; double *edge_tmin;
    HDescent *hd = nullptr;         // null: the one-launch forms of the pick (they call sec_prepare themselves)
    RDR_FN void make_lean() { lean_scene(a.sc); lean_slice(a.v); lean_slice(ev); }
    RDR_FN void make_mid() { mid_scene(a.sc); }
    RDR_FN void operator()(int idx) const {
        int p = a.active[idx];
        int l0 = 2 * idx, l1 = 2 * idx + 1;
        SecPre s = sec_prepare(a.sc, a.es, a.rng_main, a.dim_main, a.rng_edge, a.dim_edge, a.v, p, idx);
        SecondaryEdgeRec rec;
        rec.edge = EdgeD{-1, 0, 0, 0, 0};
        rec.edge_pt = rec.mwt = v3(0); rec.sp_pos = s.c.sp.position;
        rec.use_nee_ray = 0; rec.diffuse_or_glossy = 0;
        recs[idx] = rec;
        picks[idx] = SecPick{-1, 0.0, v3(0), v3(0)};
        for (int l = l0; l <= l1; ++l) {
            st3(ev.thr, ev.n, l, 0, v3(0));
            store_ray(ev, l, v3(0), v3(0));
            ev.shape[l] = -1; ev.tri[l] = -1;
            ev.mrough[l] = s.c.mrough;
            edge_tmin[l] = 1e-3f;
            store_rdiff(ev, l, raydiff_zero());
        }
        // NEE-mode slots are listed in two groups: a segment that starts on a finely tessellated shape begins inside a cloud of
        // billboards and meets 64 ... 250+ hierarchy entries, one that starts on a wall meets ~10 -- kept apart, every wave of
        // the gather holds slots of one kind (its lanes otherwise idle through the longest walk of their wave)
        const bool dense = s.live && s.c.shape->num_triangles >= kDenseShapeTriangles;
        mode[idx] = !s.live ? 0 : (s.use_nee ? (dense ? 3 : 2) : 1);
        if (hd && s.live && !s.use_nee) {
            HDescent d;
            d.pos[0] = s.lc.pos.x; d.pos[1] = s.lc.pos.y; d.pos[2] = s.lc.pos.z;
            for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) d.m_inv[3 * r + k] = s.lc.m_inv.m[r][k];
            d.sample = s.edge_sel; d.resample = s.resample_sel; d.nleaf = 0; d.pad = 0;
            hd[idx] = d;
        }
    }
};
struct KeepMode {
    const unsigned char *mode; unsigned char want;
    RDR_FN bool operator()(int idx) const { return mode[idx] == want; }
};

struct SecEdgePickH {
    SecEdgeArgs a; const int *slots; SecPick *picks;
    RDR_FN void make_lean() { lean_scene(a.sc); lean_slice(a.v); }
    RDR_FN void make_mid() { mid_scene(a.sc); }
    RDR_FN void operator()(int i) const {
        int idx = slots[i];
        SecPre s = sec_prepare(a.sc, a.es, a.rng_main, a.dim_main, a.rng_edge, a.dim_edge, a.v, a.active[idx], idx);
        double ew = 0;
        int eid = pick_edge_hierarchical(a.sc, a.es, s.lc, s.edge_sel, s.resample_sel, ew);
        picks[idx] = SecPick{eid, ew, v3(0), v3(0)};
    }
};
template <bool PRELOAD> struct SecEdgePickH2 {        // the hierarchical pick with deferred leaf evaluation (pick_edge_hierarchical_deferred)
    SecEdgeArgs a; const int *slots; SecPick *picks; HLeaf *leaves, *spill; int n;      // leaves: kHSamples x n, spill: (kHSamples - kHStackLds) x n
    RDR_FN void make_lean() { lean_scene(a.sc); lean_slice(a.v); }
    RDR_FN void make_mid() { mid_scene(a.sc); }
    RDR_FN void operator()(int i) const {
        int idx = slots[i];
        SecPre s = sec_prepare(a.sc, a.es, a.rng_main, a.dim_main, a.rng_edge, a.dim_edge, a.v, a.active[idx], idx);
        double ew = 0;
        int eid = pick_edge_hierarchical_deferred<PRELOAD>(a.sc, a.es, s.lc, s.edge_sel, s.resample_sel, ew, leaves + i, spill + i, (size_t)n);
        picks[idx] = SecPick{eid, ew, v3(0), v3(0)};
    }
};
// ---- the same pick in two launches (round 6): a resumable DESCENT with lane refill, then the leaves ------------------------
// SecEdgePickH2 runs one slot per lane from its first root to its last leaf: a slot takes 20 ... 300 interior steps, the wave
// waits for its longest lane (lane utilisation 0.60 at fp64 issue 0.82, profiles/r5_notes.md).  The descent consumes only
// `sample`, the leaves only `resample` (pick_edge_hierarchical_deferred), so they separate cleanly:
//   SecEdgePickHDescend  a walk (State / begin / step / finish) for exec::launch_chunked: a wave owns 256 consecutive list
//                        positions and its idle lanes take the next ones (ballot + popcount, no atomics); a step pops ONE entry
//                        -- an interior node is split, a leaf is recorded; begin() reads the slot's HDescent record (what
//                        SecEdgeSetup's sec_prepare found: shading position, LTC matrix, the two random numbers of the pick
//                        -- the one-launch form computed sec_prepare a second time), finish() adds the number of leaves;
//   SecEdgePickHLeaves   one lane per slot: importance of the recorded leaves in pop order, reservoir, SecPick.
// Same operations on the same operands in the same order per slot as pick_edge_hierarchical_deferred: identical picks.
template <bool PRELOAD> struct SecEdgePickHDescend {
    EdgeSceneD es; const int *slots; HLeaf *leaves, *spill; HDescent *hd; int n;
    struct State {
        int i, idx, sp, nleaf;
        double sample;
        LtcCtx c;
        SilQuery q_pos;
        RDR_WALK_STACK_MEMBER(int, st_ref, kHStackLds)
        RDR_WALK_STACK_MEMBER(unsigned char, st_num, kHStackLds)
        RDR_WALK_STACK_MEMBER(double, st_pmf, kHStackLds)
    };
    RDR_FN void make_lean() {}
    RDR_FN void make_mid() {}
    RDR_DEV_FN void push(State &st, int r, int nn, double p) const {
        auto st_ref = RDR_WALK_STACK(st, int, st_ref, kHStackLds, 11);
        auto st_num = RDR_WALK_STACK(st, unsigned char, st_num, kHStackLds, 12);
        auto st_pmf = RDR_WALK_STACK(st, double, st_pmf, kHStackLds, 13);
        if (st.sp < kHStackLds) { RDR_WALK_AT(st_ref, st.sp) = r; RDR_WALK_AT(st_num, st.sp) = (unsigned char)nn; RDR_WALK_AT(st_pmf, st.sp) = p; }
        else spill[(size_t)(st.sp - kHStackLds) * (size_t)n + st.i] = HLeaf{r, nn, p};
        st.sp++;
    }
    RDR_DEV_FN bool begin(int i, State &st) const {
        st.i = i; st.idx = slots[i]; st.sp = 0; st.nleaf = 0;
        const HDescent d = hd[st.idx];
        st.sample = d.sample;
        st.c.pos = V3{d.pos[0], d.pos[1], d.pos[2]};
        for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) st.c.m_inv.m[r][k] = d.m_inv[3 * r + k];
        st.c.m = m3_inverse(st.c.m_inv);              // as sec_prepare does (ltc_bound needs both; the leaves only m_inv)
        st.q_pos = sil_query(es, st.c.pos);
        double imp_cs = es.cs_root != kNoEdgeTree ? 1.0 : 0.0, imp_ncs = es.ncs_root != kNoEdgeTree ? 1.0 : 0.0;
        if (imp_cs <= 0 && imp_ncs <= 0) { st.nleaf = -1; return false; }            // no edge tree: the pick returns -1
        double prob_cs = imp_cs / (imp_cs + imp_ncs), prob_ncs = 1 - prob_cs;
        double exp_cs = kHSamples * prob_cs, exp_ncs = kHSamples * prob_ncs;
        int n_cs = int(floor(exp_cs)), n_ncs = int(floor(exp_ncs));
        if (n_cs + n_ncs < kHSamples) {
            double prob = exp_cs - n_cs;
            if (st.sample < prob) { n_cs++; st.sample /= prob; }
            else { n_ncs++; st.sample = (st.sample - prob) / (1 - prob); }
        }
        if (n_cs > 0) push(st, es.cs_root, n_cs, prob_cs);
        if (n_ncs > 0) push(st, es.ncs_root, n_ncs, prob_ncs);
        return st.sp > 0;
    }
    RDR_DEV_FN bool step(State &st) const {
        auto st_ref = RDR_WALK_STACK(st, int, st_ref, kHStackLds, 11);
        auto st_num = RDR_WALK_STACK(st, unsigned char, st_num, kHStackLds, 12);
        auto st_pmf = RDR_WALK_STACK(st, double, st_pmf, kHStackLds, 13);
        --st.sp;
        HItem it;
        if (st.sp < kHStackLds) it = HItem{RDR_WALK_AT(st_ref, st.sp), (int)RDR_WALK_AT(st_num, st.sp), RDR_WALK_AT(st_pmf, st.sp)};
        else { const HLeaf sl = spill[(size_t)(st.sp - kHStackLds) * (size_t)n + st.i]; it = HItem{sl.ref, sl.num, sl.pmf}; }
        if (it.ref < 0) {
            leaves[(size_t)st.nleaf * (size_t)n + st.i] = HLeaf{it.ref, it.num, it.pmf};
            st.nleaf++;
            return st.sp == 0;
        }
        const EdgeNodeP nd_line = PRELOAD ? load_node_line(&edge_node(es, it.ref)) : EdgeNodeP();
        const EdgeNodeP &nd = PRELOAD ? nd_line : edge_node(es, it.ref);
        const int tree = it.ref & kEdgeTreeBit;
        const bool tree3d = tree == 0;
        int c0 = nd.c_ref[0] < 0 ? nd.c_ref[0] : (nd.c_ref[0] | tree);
        int c1 = nd.c_ref[1] < 0 ? nd.c_ref[1] : (nd.c_ref[1] | tree);
        double i0, i1;
        if (box_contains(v3_of(nd.p_min), v3_of(nd.p_max), st.c.pos)) { i0 = i1 = 1; }
        else { i0 = node_importance(nd, 0, tree3d, st.c, st.q_pos); i1 = node_importance(nd, 1, tree3d, st.c, st.q_pos); }
        if (i0 > 0 || i1 > 0) {
            double p0 = i0 / (i0 + i1), p1 = 1 - p0;
            double e0 = it.num * p0, e1 = it.num * p1;
            int s0 = int(floor(e0)), s1 = int(floor(e1));
            if (s0 + s1 < it.num) {
                double prob = e0 - s0;
                if (st.sample < prob) { s0++; st.sample /= prob; }
                else { s1++; st.sample = (st.sample - prob) / (1 - prob); }
            }
            if (s0 > 0 && st.sp + st.nleaf < kHSamples) push(st, c0, s0, it.pmf * p0);
            if (s1 > 0 && st.sp + st.nleaf < kHSamples) push(st, c1, s1, it.pmf * p1);
        }
        return st.sp == 0;
    }
    RDR_DEV_FN void finish(State &st) const { hd[st.idx].nleaf = st.nleaf; }
};
struct SecEdgePickHLeaves {
    SecEdgeArgs a; const int *slots; SecPick *picks; const HLeaf *leaves; const HDescent *in; int n;
    RDR_FN void make_lean() { lean_scene(a.sc); lean_slice(a.v); }
    RDR_FN void make_mid() { mid_scene(a.sc); }
    RDR_FN void operator()(int i) const {
        const int idx = slots[i];
        const HDescent d = in[idx];
        LtcCtx c;
        c.pos = V3{d.pos[0], d.pos[1], d.pos[2]};
        for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) { c.m_inv.m[r][k] = d.m_inv[3 * r + k]; c.m.m[r][k] = 0; }
        double resample = d.resample;
        int selected = -1;
        double edge_w = 0, wsum = 0;
        for (int j = 0; j < d.nleaf; ++j) {
            const HLeaf it = leaves[(size_t)j * (size_t)n + i];
            const int leaf_edge = ~it.ref;
            double w = it.num * leaf_importance_h(a.sc, a.es, leaf_edge, c) / it.pmf;
            if (w > 0) {
                double prev = wsum;
                wsum += w;
                double nw = w / wsum;
                if (resample <= nw || prev == 0) {
                    selected = leaf_edge;
                    edge_w = w * it.pmf;
                    resample /= nw;
                } else {
                    resample = (resample - nw) / (1 - nw);
                }
            }
        }
        int eid = -1;
        double ew = 0;
        if (d.nleaf >= 0 && !(edge_w <= 0 || wsum <= 0)) { ew = 1 / (edge_w * kHSamples / wsum); eid = selected; }
        picks[idx] = SecPick{eid, ew, v3(0), v3(0)};
    }
};
// SecEdgePickHLeaves as a walk (one recorded leaf per step) for exec::launch_chunked: slots record 1 ... 16 leaves.
struct SecEdgePickHLeavesWalk {
    SceneD sc; EdgeSceneD es; const int *slots; SecPick *picks; const HLeaf *leaves; const HDescent *in; int n;
    struct State { int i, idx, j, nleaf, selected; double resample, edge_w, wsum; LtcCtx c; };
    RDR_FN void make_lean() { lean_scene(sc); }
    RDR_FN void make_mid() { mid_scene(sc); }
    RDR_DEV_FN bool begin(int i, State &st) const {
        st.i = i; st.idx = slots[i]; st.j = 0; st.selected = -1; st.edge_w = 0; st.wsum = 0;
        const HDescent d = in[st.idx];
        st.nleaf = d.nleaf; st.resample = d.resample;
        st.c.pos = V3{d.pos[0], d.pos[1], d.pos[2]};
        for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) { st.c.m_inv.m[r][k] = d.m_inv[3 * r + k]; st.c.m.m[r][k] = 0; }
        return st.nleaf > 0;
    }
    RDR_DEV_FN bool step(State &st) const {
        const HLeaf it = leaves[(size_t)st.j * (size_t)n + st.i];
        const int leaf_edge = ~it.ref;
        double w = it.num * leaf_importance_h(sc, es, leaf_edge, st.c) / it.pmf;
        if (w > 0) {
            double prev = st.wsum;
            st.wsum += w;
            double nw = w / st.wsum;
            if (st.resample <= nw || prev == 0) {
                st.selected = leaf_edge;
                st.edge_w = w * it.pmf;
                st.resample /= nw;
            } else {
                st.resample = (st.resample - nw) / (1 - nw);
            }
        }
        return ++st.j >= st.nleaf;
    }
    RDR_DEV_FN void finish(State &st) const {
        int eid = -1;
        double ew = 0;
        if (st.nleaf >= 0 && !(st.edge_w <= 0 || st.wsum <= 0)) { ew = 1 / (st.edge_w * kHSamples / st.wsum); eid = st.selected; }
        picks[st.idx] = SecPick{eid, ew, v3(0), v3(0)};
    }
};
// The NEE-mode pick as a resumable walk for exec::launch_persistent: same tests in the same order as
// the reference's sample_edge_l (src/edge.cpp:1239-1364), one popped reference per step.
constexpr int kPickOverflow = -2;     // SecPick::eid of a slot the gather hands to the reference-order walk
template <int NS> struct SecEdgePickNWalk {
    SecEdgeArgs a; const int *slots; SecPick *picks;
    int only_overflow;          // 1: walk only the slots SecEdgeGatherN marked kPickOverflow, leave the others alone
    const int *walk_needed = nullptr;      // (with only_overflow) GatherBook::walk_needed: 0 = no slot kept the mark
    RDR_DEV_FN bool gate_closed() const { return only_overflow && walk_needed && *walk_needed == 0; }
    struct State {
        int idx, sp, selected;
        double edge_w, wsum, resample;
        LtcCtx c;
        Ray nee; bool nee_valid;
        V3 inv_dir;                 // 1 / nee.dir
        SilQuery q_pos, q_nee;      // silhouette queries of the shading point and of the light point
        RDR_WALK_STACK_MEMBER(int, stack, NS)
    };
    RDR_FN void make_lean() { lean_scene(a.sc); lean_slice(a.v); }
    RDR_FN void make_mid() { mid_scene(a.sc); }
    RDR_DEV_FN bool begin(int i, State &st) const {
        st.idx = slots[i];
        if (only_overflow && picks[st.idx].eid != kPickOverflow) { st.idx = -1; st.sp = 0; st.selected = -1; return false; }
        SecPre s = sec_prepare(a.sc, a.es, a.rng_main, a.dim_main, a.rng_edge, a.dim_edge, a.v, a.active[st.idx], st.idx);
        st.sp = 0; st.selected = -1; st.edge_w = 0; st.wsum = 0; st.resample = s.resample_sel;
        st.c = s.lc; st.nee = s.nee; st.nee_valid = s.nee_valid;
        st.inv_dir = V3{1 / s.nee.dir.x, 1 / s.nee.dir.y, 1 / s.nee.dir.z};
        st.q_pos = sil_query(a.es, s.lc.pos); st.q_nee = sil_query(a.es, s.nee_pt.position);
        auto stack = RDR_WALK_STACK(st, int, stack, NS, 1);
        if (a.es.cs_root != kNoEdgeTree) { RDR_WALK_AT(stack, st.sp) = a.es.cs_root; st.sp++; }
        if (a.es.ncs_root != kNoEdgeTree) { RDR_WALK_AT(stack, st.sp) = a.es.ncs_root; st.sp++; }
        return st.sp > 0;
    }
    RDR_DEV_FN bool step(State &st) const {
        const SceneD &sc = a.sc; const EdgeSceneD &es = a.es;
        auto stack = RDR_WALK_STACK(st, int, stack, NS, 1);
        --st.sp;
        int ref = RDR_WALK_AT(stack, st.sp);
        if (ref < 0) {
            const int leaf_edge = ~ref;
            double w = leaf_importance_l(sc, es, leaf_edge, st.c, st.nee, st.nee_valid);
            if (w > 0) {
                double prev = st.wsum;
                st.wsum += w;
                double nw = w / st.wsum;
                if (st.resample <= nw || prev == 0) { st.selected = leaf_edge; st.edge_w = w; st.resample /= nw; }
                else st.resample = (st.resample - nw) / (1 - nw);
            }
        } else {
            const EdgeNodeP &nd = edge_node(es, ref);
            const int tree = ref & kEdgeTreeBit;
            const bool tree3d = tree == 0;
            for (int k = 0; k < 2; ++k) {
                bool ok = may_hold_silhouette(tree3d, nd.c_dx_min[k], nd.c_dx_max[k], st.q_pos);
                if (ok && st.nee_valid) ok = may_hold_silhouette(tree3d, nd.c_dx_min[k], nd.c_dx_max[k], st.q_nee);
                if (ok) ok = ray_box_expand(v3_of(nd.c_pmin[k]), v3_of(nd.c_pmax[k]), st.nee, st.inv_dir, es.edge_bounds_expand);
                if (ok && st.sp < NS) { RDR_WALK_AT(stack, st.sp) = nd.c_ref[k] < 0 ? nd.c_ref[k] : (nd.c_ref[k] | tree); st.sp++; }
            }
        }
        return st.sp == 0;
    }
    // weight, point on the edge and edge vector of the selected edge (src/edge.cpp:1319-1363); the NEE segment
    // is rebuilt from the slot rather than carried through the walk
    RDR_DEV_FN void finish(State &st) const {
        const SceneD &sc = a.sc; const EdgeSceneD &es = a.es;
        if (st.idx < 0) return;
        SecPick out{-1, 0.0, v3(0), v3(0)};
        if (st.selected != -1) {
            SecPre s = sec_prepare(sc, es, a.rng_main, a.dim_main, a.rng_edge, a.dim_edge, a.v, a.active[st.idx], st.idx);
            double ew = 0;
            V3 sample_p = v3(0), mwt = v3(0);
            int eid = finish_edge_nee(sc, es, s.nee, s.nee_valid, s.nee_pt, s.nee_shape, st.selected, st.edge_w, st.wsum, ew, sample_p, mwt);
            out = SecPick{eid, ew, sample_p, mwt};
        }
        picks[st.idx] = out;
    }
};

// The NEE-mode pick without the reference's traversal order.
//
// sample_edge_l (src/edge.cpp:1239-1364) walks both edge hierarchies depth-first, descends into a child when (i) its Hough
// x-interval may hold a silhouette seen from the shading point and from the light point and (ii) the NEE segment passes its
// spatial box grown by the billboard half-width, evaluates leaf_importance_l at every leaf it reaches and keeps one leaf by
// reservoir sampling.  All three node tests are monotone in box inclusion, also in floating point (a bigger interval /
// box passes whenever a smaller one does: every operation in them is monotone and rounds monotonically), and an inner
// node's bounds are the unions of its children's.  So a leaf is reached exactly when its OWN bounds pass the tests -- the
// hierarchy only prunes -- and the outcome is determined by the set of leaves with positive importance and by the order in
// which the walk meets them, which is a fixed order of the leaves (EdgeSceneD::leaf_rank).
// This stage therefore finds those leaves with any traversal it likes -- here a SAH hierarchy over the billboard boxes
// (EdgeSceneD::gather; the reference's trees are built for importance sampling, a segment query visits 100 ... 600+ of their
// nodes per slot with a long tail, profiles/r1_notes.md) -- applies the reference's own tests to each candidate edge, and
// replays the reservoir over the positive leaves in rank order: identical arithmetic on identical operands, identical picks.
// Slots with more than kGatherCands positive leaves are marked kPickOverflow and walked by SecEdgePickNWalk.
// Stack entries are 32-bit: an inner node is pushed as the index of its first child (its two children are adjacent, one
// 64-byte fetch per step), a leaf as kGatherLeafBit | count << 24 | first slot -- what a step needs is known when it pops.
constexpr int kGatherLeafBit = 1 << 30;
struct alignas(16) GatherQuad { float x, y, z; int w; };     // half an rt::Node: {lo.xyz, a} or {hi.xyz, b}
// ray_box_expand without the per-axis early return: the entry distance only grows and the exit distance only shrinks from
// axis to axis, so "t0 > t1 after some axis" and "t0 > t1 after the last axis" are the same verdict; without the returns the
// three axes' loads are issued together instead of one dependent round trip per axis.
RDR_FN bool ray_box_all_axes(V3 lo, V3 hi, const Ray &r, V3 inv_dir, double expand) {
    double t0 = r.tmin, t1 = r.tmax;
    for (int i = 0; i < 3; ++i) {
        double inv = comp(inv_dir, i);
        double tn = (comp(lo, i) - expand - comp(r.org, i)) * inv;
        double tf = (comp(hi, i) + expand - comp(r.org, i)) * inv;
        if (tn > tf) { double t = tn; tn = tf; tf = t; }
        tf *= (1 + 1e-6f);
        t0 = tn > t0 ? tn : t0;
        t1 = tf < t1 ? tf : t1;
    }
    return !(t0 > t1);
}
RDR_FN int gather_entry(const GatherQuad &lo, const GatherQuad &hi) {      // stack entry for the node whose record is (lo, hi)
    return hi.w > 0 ? (kGatherLeafBit | (hi.w << 24) | lo.w) : lo.w;
}
// Three stages.  A segment that runs along the silhouette of a finely tessellated shape meets thousands of billboards:
// one lane walking all of them would set the duration of the whole launch (the reference-order walk has the same tail:
// 640 steps at the 95th percentile, thousands at the end, profiles/r1_notes.md).  So
//   SecEdgeGatherN      one lane per slot, at most kGatherBudget pops and kGatherCands positive leaves.  A lane that needs
//                       more registers its slot as "heavy" (atomic counter), moves its candidates to the slot's big list and
//                       hands every entry left on its stack -- a subtree each -- to the work list;
//   SecEdgeGatherSub    one lane per work item: walks that subtree to the end, appending to the slot's big list;
//   SecEdgeGatherReplay one lane per heavy slot: reservoir replay over the big list.
// A slot whose lists overflow (kGatherCandsBig candidates, kGatherHeavyCap slots, kGatherWorkCap items) keeps the
// kPickOverflow mark and is walked by SecEdgePickNWalk.
constexpr int kGatherBudget = 256, kGatherCandsBig = 256, kGatherHeavyCap = 8192, kGatherWorkCap = 131072, kGatherPoison = 1 << 20;
constexpr int kGatherHeavyMax = 65536, kGatherWorkMax = 1 << 20;      // (kGatherHeavyCap / kGatherWorkCap: the smallest lists; render.cpp sizes them by the launch set)
struct GatherWork { int heavy, entry; };
struct GatherBook {                 // zeroed before every SecEdgeGatherN launch
    int heavy_count, work_count;
    int walk_needed;                // some slot keeps its kPickOverflow mark: SecEdgePickNWalk has work (else it returns at once)
    int cand_count[kGatherHeavyMax];
};
// heavy_cap / work_cap: how much of the (kGatherHeavyCap / kGatherWorkCap sized) lists may be used -- the full size, or less
// when RDR_GATHER_CAPS=h,w asks for it, which is how the tests reach the overflow paths.
struct GatherShared { GatherBook *book; int *heavy_slot; GatherWork *work; GatherCand *cands_big; int heavy_cap, work_cap; };

struct GatherCtx { LtcCtx lc; Ray nee; bool nee_valid; V3 inv_dir; SilQuery q_pos, q_nee; double resample; };
RDR_FN GatherCtx gather_ctx(const SecEdgeArgs &a, int idx) {
    GatherCtx c;
    SecPre s = sec_prepare(a.sc, a.es, a.rng_main, a.dim_main, a.rng_edge, a.dim_edge, a.v, a.active[idx], idx);
    c.lc = s.lc; c.nee = s.nee; c.nee_valid = s.nee_valid; c.resample = s.resample_sel;
    c.q_pos = sil_query(a.es, s.lc.pos); c.q_nee = sil_query(a.es, s.nee_pt.position);
    c.inv_dir = V3{1 / s.nee.dir.x, 1 / s.nee.dir.y, 1 / s.nee.dir.z};
    return c;
}
RDR_FN void gather_append_big(const GatherShared &sh, int heavy, const GatherCand &cd) {
    if (heavy < 0 || heavy >= sh.heavy_cap) return;
    const int k = atomic_fetch_add(&sh.book->cand_count[heavy], 1);
    if (k >= 0 && k < kGatherCandsBig) sh.cands_big[(size_t)heavy * kGatherCandsBig + k] = cd;
}
// One pop of the gather: an inner entry tests its two children and pushes the ones the segment may reach, a leaf entry
// applies the reference's own tests to its 1..4 edges and reports the positive ones through `emit`.
// The walk of the gather, in two alternating phases so that the lanes of a wave run the same body: (A) pop inner entries,
// test their two children, push inner children on the node stack and LEAF children on a small per-lane list; (B) when the list
// is nearly full or the stack is empty, evaluate the listed leaves -- the reference's own tests on each of their 1..4 edges,
// positive ones reported through `emit`.  (Interleaved, a wave executed both bodies in nearly every iteration with a fraction of
// its lanes in each: lane utilisation 0.22.)  Stops with entries left on the stack when `budget` pops are spent.
constexpr int kGatherLeafBatch = 16;
template <int NS, class Emit>
RDR_DEV_FN void gather_walk(const SceneD &sc, const EdgeSceneD &es, const GatherCtx &c, int *stk, int &sp, int *lst, int &nl,
                            long budget, const Emit &emit, long &pops, long &h_edges) {
    const GatherQuad *quads = reinterpret_cast<const GatherQuad *>(es.gather.nodes);
    while (sp > 0 || nl > 0) {
        while (sp > 0 && nl <= kGatherLeafBatch - 2 && pops < budget) {
            --sp; ++pops;
            const int e = RDR_STACK_AT(stk, sp);
            const GatherQuad l_lo = quads[2 * e], l_hi = quads[2 * e + 1], r_lo = quads[2 * e + 2], r_hi = quads[2 * e + 3];
            const int el = gather_entry(l_lo, l_hi), er = gather_entry(r_lo, r_hi);      // (before the tests: whole-record loads)
            const bool hl = ray_box_all_axes(V3{(double)l_lo.x, (double)l_lo.y, (double)l_lo.z}, V3{(double)l_hi.x, (double)l_hi.y, (double)l_hi.z}, c.nee, c.inv_dir, 0.0);
            const bool hr = ray_box_all_axes(V3{(double)r_lo.x, (double)r_lo.y, (double)r_lo.z}, V3{(double)r_hi.x, (double)r_hi.y, (double)r_hi.z}, c.nee, c.inv_dir, 0.0);
            if (hl) {
                if (el & kGatherLeafBit) { RDR_STACK_AT(lst, nl) = el; nl++; }
                else if (sp < NS) { RDR_STACK_AT(stk, sp) = el; sp++; }
            }
            if (hr) {
                if (er & kGatherLeafBit) { RDR_STACK_AT(lst, nl) = er; nl++; }
                else if (sp < NS) { RDR_STACK_AT(stk, sp) = er; sp++; }
            }
        }
        for (int j = 0; j < nl; ++j) {
            const int e = RDR_STACK_AT(lst, j);
            const int first = e & 0xffffff, count = (e >> 24) & 63;
            ++pops;
            for (int k = 0; k < count; ++k) {
                const GatherLeaf gl = es.gleaf[first + k];
                h_edges++;
                // the edge's own leaf, tested like the reference tests it from its parent
                bool ok = sphere_box_x(c.q_pos, gl.dx_lo, gl.dx_hi);
                if (c.nee_valid) ok = ok && sphere_box_x(c.q_nee, gl.dx_lo, gl.dx_hi);
                const V3 p0 = v3_of(gl.v0), p1 = v3_of(gl.v1);
                const V3 blo = V3{dmin(p0.x, p1.x), dmin(p0.y, p1.y), dmin(p0.z, p1.z)}, bhi = V3{dmax(p0.x, p1.x), dmax(p0.y, p1.y), dmax(p0.z, p1.z)};
                ok = ok && ray_box_all_axes(blo, bhi, c.nee, c.inv_dir, es.edge_bounds_expand);
                if (!ok) continue;
                h_edges += 1000000;          // (harness statistics: survivors of the cheap tests in the upper digits)
                const double w = leaf_importance_gathered(gl, es.edge_bounds_expand, c.lc, c.nee, c.nee_valid);
                if (w > 0) emit(GatherCand{gl.rank, gl.eid, w});
            }
        }
        nl = 0;
        if (pops >= budget) break;
    }
}
// Reservoir replay in the reference's leaf order (src/edge.cpp:1300-1316) over `n` candidates, then the pick's tail.
RDR_FN SecPick gather_replay(const SecEdgeArgs &a, int idx, const GatherCand *list, int n, double resample) {
    int selected = -1, last_rank = -1;
    double edge_w = 0, wsum = 0;
    for (int t = 0; t < n; ++t) {
        int best = -1, best_rank = 0x7fffffff;
        for (int j = 0; j < n; ++j) {
            const int r = list[j].rank;
            if (r > last_rank && r < best_rank) { best_rank = r; best = j; }
        }
        const GatherCand cd = list[best];
        last_rank = best_rank;
        const double prev = wsum;
        wsum += cd.w;
        const double nw = cd.w / wsum;
        if (resample <= nw || prev == 0) { selected = cd.eid; edge_w = cd.w; resample /= nw; }
        else resample = (resample - nw) / (1 - nw);
    }
    SecPick out{-1, 0.0, v3(0), v3(0)};
    if (selected != -1) {
        SecPre s = sec_prepare(a.sc, a.es, a.rng_main, a.dim_main, a.rng_edge, a.dim_edge, a.v, a.active[idx], idx);
        double ew = 0;
        V3 sample_p = v3(0), mwt = v3(0);
        int eid = finish_edge_nee(a.sc, a.es, s.nee, s.nee_valid, s.nee_pt, s.nee_shape, selected, edge_w, wsum, ew, sample_p, mwt);
        out = SecPick{eid, ew, sample_p, mwt};
    }
    return out;
}

template <int NS> struct SecEdgeGatherN {
    SecEdgeArgs a; const int *slots; SecPick *picks; GatherCand *cands;     // cands: kGatherCands per list position
    GatherShared sh; int budget;                                            // budget: kGatherBudget (RDR_GATHER_BUDGET overrides)
    RDR_FN void make_lean() { lean_scene(a.sc); lean_slice(a.v); }
    RDR_FN void make_mid() { mid_scene(a.sc); }
    RDR_FN void operator()(int i) const {
        const SceneD &sc = a.sc; const EdgeSceneD &es = a.es;
        const int idx = slots[i];
        const GatherCtx c = gather_ctx(a, idx);
        GatherCand *mine = cands + (size_t)kGatherCands * i;
        int ncand = 0, heavy = -1;
        RDR_STACK_DECL(int, stk, NS);
        RDR_STACK_DECL(int, lst, kGatherLeafBatch);
        int sp = 0, nl = 0;
        if (es.gather.num_nodes > 0) {
            const GatherQuad *quads = reinterpret_cast<const GatherQuad *>(es.gather.nodes);
            const GatherQuad lo = quads[0], hi = quads[1];
            if (ray_box_all_axes(V3{(double)lo.x, (double)lo.y, (double)lo.z}, V3{(double)hi.x, (double)hi.y, (double)hi.z}, c.nee, c.inv_dir, 0.0)) {
                const int e = gather_entry(lo, hi);          // a one-leaf hierarchy: the root goes straight to the leaf list
                if (e & kGatherLeafBit) { RDR_STACK_AT(lst, nl) = e; nl++; }
                else { RDR_STACK_AT(stk, sp) = e; sp++; }
            }
        }
        auto promote = [&]() {              // this slot continues in the big lists
            heavy = atomic_fetch_add(&sh.book->heavy_count, 1);
            if (heavy < sh.heavy_cap) sh.heavy_slot[heavy] = i;
            else sh.book->walk_needed = 1;                 // no room in the big lists: the mark stays, the walk takes the slot
            for (int k = 0; k < ncand; ++k) gather_append_big(sh, heavy, mine[k]);
        };
        auto emit = [&](const GatherCand &cd) {
            if (heavy < 0 && ncand < kGatherCands) { mine[ncand] = cd; ncand++; return; }
            if (heavy < 0) promote();
            gather_append_big(sh, heavy, cd);
        };
        long h_nodes = 0, h_edges = 0;
        gather_walk<NS>(sc, es, c, stk, sp, lst, nl, budget, emit, h_nodes, h_edges);
        if (sp > 0) {                        // budget spent: every entry left is a subtree for SecEdgeGatherSub
            if (heavy < 0) promote();
            bool lost = heavy >= sh.heavy_cap;
            for (int k = 0; k < sp && !lost; ++k) {
                const int w = atomic_fetch_add(&sh.book->work_count, 1);
                if (w < sh.work_cap) sh.work[w] = GatherWork{heavy, RDR_STACK_AT(stk, k)};
                else lost = true;
            }
            if (lost && heavy < sh.heavy_cap) atomic_fetch_add(&sh.book->cand_count[heavy], kGatherPoison);
            if (lost) sh.book->walk_needed = 1;
        }
        exec::gather_stats_add(h_nodes, h_edges % 1000000, ncand, (int)(h_edges / 1000000));      // (harness statistics; nothing on the GPU)
        if (heavy >= 0) { picks[idx] = SecPick{kPickOverflow, 0.0, v3(0), v3(0)}; return; }
        picks[idx] = gather_replay(a, idx, mine, ncand, c.resample);
    }
};
template <int NS> struct SecEdgeGatherSub {
    SecEdgeArgs a; const int *slots; GatherShared sh;
    RDR_FN void make_lean() { lean_scene(a.sc); lean_slice(a.v); }
    RDR_FN void make_mid() { mid_scene(a.sc); }
    RDR_FN void operator()(int j) const {
        const int n_work = sh.book->work_count < sh.work_cap ? sh.book->work_count : sh.work_cap;
        if (j >= n_work) return;
        const GatherWork wk = sh.work[j];
        const int idx = slots[sh.heavy_slot[wk.heavy]];
        const GatherCtx c = gather_ctx(a, idx);
        RDR_STACK_DECL(int, stk, NS);
        RDR_STACK_DECL(int, lst, kGatherLeafBatch);
        int sp = 0, nl = 0;
        RDR_STACK_AT(stk, sp) = wk.entry; sp++;             // (only inner entries are ever left on a stack)
        auto emit = [&](const GatherCand &cd) { gather_append_big(sh, wk.heavy, cd); };
        long pops = 0, h_edges = 0;
        gather_walk<NS>(a.sc, a.es, c, stk, sp, lst, nl, (long)1 << 60, emit, pops, h_edges);
    }
};
struct SecEdgeGatherReplay {
    SecEdgeArgs a; const int *slots; SecPick *picks; GatherShared sh;
    RDR_FN void make_lean() { lean_scene(a.sc); lean_slice(a.v); }
    RDR_FN void make_mid() { mid_scene(a.sc); }
    RDR_FN void operator()(int h) const {
        const int n_heavy = sh.book->heavy_count < sh.heavy_cap ? sh.book->heavy_count : sh.heavy_cap;
        if (h >= n_heavy) return;
        const int n = sh.book->cand_count[h];
        if (n < 0 || n > kGatherCandsBig) { sh.book->walk_needed = 1; return; }          // stays kPickOverflow: SecEdgePickNWalk takes it
        const int idx = slots[sh.heavy_slot[h]];
        SecPre s = sec_prepare(a.sc, a.es, a.rng_main, a.dim_main, a.rng_edge, a.dim_edge, a.v, a.active[idx], idx);
        picks[idx] = gather_replay(a, idx, sh.cands_big + (size_t)h * kGatherCandsBig, n, s.resample_sel);
    }
};

struct SecEdgeFinish {
    SecEdgeArgs a; const unsigned char *mode; const SecPick *picks;
    const float *d_image; int nd, radiance_dim;
    SecondaryEdgeRec *recs; VSlice ev; double *edge_tmin;
    RDR_FN void make_lean() { lean_scene(a.sc); lean_slice(a.v); lean_slice(ev); nd = 3; radiance_dim = 0; }
    RDR_FN void make_mid() { mid_scene(a.sc); nd = 3; radiance_dim = 0; }
    RDR_FN void operator()(int idx) const {
        if (mode[idx] == 0) return;
        SecPick pkd = picks[idx];
        int eid = pkd.eid;
        double ew = pkd.ew;
        if (eid == -1 || ew <= 0) return;
        int p = a.active[idx];
        int l0 = 2 * idx, l1 = 2 * idx + 1;
        const SceneD &sc = a.sc;
        const EdgeSceneD &es = a.es;
        SecPre s = sec_prepare(sc, es, a.rng_main, a.dim_main, a.rng_edge, a.dim_edge, a.v, p, idx);
        const VertexCtx &c = s.c;
        const LtcCtx &lc = s.lc;
        V3 sample_p = pkd.sample_p, mwt = pkd.mwt;
        if (!s.use_nee) {
            const EdgeD &e = es.edges[eid];
            if (!edge_is_silhouette(sc.shapes, lc.pos, e)) return;
            V3 ea = edge_v0(sc.shapes, e), eb = edge_v1(sc.shapes, e);
            V3 ao = m3_apply(lc.m_inv, ea - lc.pos), bo = m3_apply(lc.m_inv, eb - lc.pos);
            if (ao.z <= 0.f && bo.z <= 0.f) return;
            LineSetup ls = line_setup(ao, bo);
            double Il0 = line_I(ls, ls.l0), Il1 = line_I(ls, ls.l1);
            double norm = Il1 - Il0;
            double lb = ls.l0, ub = ls.l1;
            if (lb > ub) { double tt = lb; lb = ub; ub = tt; }
            double l = 0.5f * (lb + ub);
            for (int it = 0; it < 20; ++it) {
                if (!(l >= lb && l <= ub)) l = 0.5f * (lb + ub);
                double value = (line_I(ls, l) - Il0) / norm - s.t_sel;
                if (fabs(value) < 1e-5f || it == 19) break;
                if (value > 0.f) ub = l; else lb = l;
                double dsq = ls.d * ls.d + l * l;
                double deriv = 2.f * ls.d * (ls.vo + l * ls.wt).z / (norm * dsq * dsq);
                l -= value / deriv;
            }
            double dsq = ls.d * ls.d + l * l;
            double lpdf = 2.f * ls.d * (ls.vo + l * ls.wt).z / (norm * dsq * dsq);
            if (lpdf <= 0.f) return;
            sample_p = m3_apply(lc.m, ls.vo + l * ls.wt);
            ew /= (s.m_pmf * lpdf);
            mwt = m3_apply(lc.m, ls.wt);
        }
        const EdgeD &e = es.edges[eid];
        V3 ea = edge_v0(sc.shapes, e), eb = edge_v1(sc.shapes, e);
        V3 hn = normalize(cross(ea - lc.pos, eb - lc.pos));
        double off = 1e-5f / len(sample_p);
        V3 sdir = normalize(sample_p);
        V3 up = normalize(sdir + off * hn), lo = normalize(sdir - off * hn);
        V3 wi = c.wi;
        V3 f = bsdf_eval(*c.mat, c.sp, wi, sdir, c.mrough);
        if (sum(f) < 1e-6f) return;
        V3 dc = image_grad(d_image, nd, radiance_dim, p);
        SecondaryEdgeRec rec = recs[idx];
        rec.edge = e; rec.edge_pt = sample_p; rec.mwt = mwt;
        rec.use_nee_ray = s.use_nee ? 1 : 0; rec.diffuse_or_glossy = s.dg ? 1 : 0;
        recs[idx] = rec;
        store_ray(ev, l0, lc.pos, up);
        store_ray(ev, l1, lc.pos, lo);
        edge_tmin[l0] = edge_tmin[l1] = 1e-3f * len(sample_p);
        RayDiff brd;
        brd.org_dx = c.rd_surf.org_dx; brd.org_dy = c.rd_surf.org_dy;
        if (s.bsdf_comp <= s.pd) {
            brd.dir_dx = V3{0.03f, 0.03f, 0.03f};
            brd.dir_dy = V3{0.03f, 0.03f, 0.03f};
        } else {
            V3 h = normalize(wi + sdir);
            double hz = dot(h, c.sp.frame.n);
            V3 dmdx = c.sp.dn_dx * hz, dmdy = c.sp.dn_dy * hz;
            V3 ddx = c.rd_surf.dir_dx, ddy = c.rd_surf.dir_dy;
            V3 ddn_dx = ddx * h - wi * dmdx, ddn_dy = ddy * h - wi * dmdy;   // per-component, as in the reference
            brd.dir_dx = ddx - 2 * (-dot(wi, h) * c.sp.dn_dx + ddn_dx * h);
            brd.dir_dy = ddy - 2 * (-dot(wi, h) * c.sp.dn_dy + ddn_dy * h);
        }
        store_rdiff(ev, l0, brd); store_rdiff(ev, l1, brd);
        if (ev.erd) { st_erd(ev, l0, brd); st_erd(ev, l1, brd); }
        V3 nt = ld3(a.v.thr, a.v.n, p, 0) * f * dc * ew / s.nee_pmf;
        st3(ev.thr, ev.n, l0, 0, nt);
        st3(ev.thr, ev.n, l1, 0, -nt);
    }
};

// Jacobian of the ray/plane intersection w.r.t. the line parameter (src/edge.cpp:1828-1853).
RDR_FN V3 isect_jacobian(V3 org, V3 dir, V3 p, V3 n, V3 l) {
    double dn = dot(dir, n);
    if (fabs(dn) < 1e-10f) return v3(0);
    double d = -dot(p, n);
    double t = -(dot(org, n) + d) / dn;
    if (t <= 0) return v3(0);
    return t * (l - dir * (dot(l, n) / dot(dir, n)));
}

// Where the edge rays of a secondary pass hit: written on hits only, and read by SecondaryEdgeDerivatives for every lane that
// carries radiance -- for a lane that reached the ENVIRONMENT light that is whatever an earlier pass left at its index, like the
// reference's never-cleared edge_surface_points (src/pathtracer.cpp:584-600, 702)  [quirk].  A sample batch keeps one copy
// per sample (lane 2 (rank) + side of the batch's list -> entry 2 (rank within the sample) + side of sample s, via `seg`), and
// marks what the batch has written: a read of an entry the SAME sample has not written during this batch needs what EARLIER
// SAMPLES left there, which in a batch is only known once every sample's passes have run.  The term that depends on it is
// LINEAR in the position (SecondaryEdgeDerivatives), and so is everything the adjoint sweep does with it afterwards: the read
// is recorded as a HitEvent and its whole contribution is replayed after the sweep (render.cpp: "replay"; InjectHitEvents
// below).  Only when the event list overflows is the read counted as a violation, and render() renders the call again one
// sample at a time.
struct HitEvent { int p, entry, shape_id, v0, v1, depth; double contrib; V3 sp_pos; };
struct HitPosView {
    double *pos; int n;                          // 3 x n, stride n
    const int *seg; int S, P0;                   // null / 0: one sample, lanes are entries
    unsigned char *written; int *violations;     // null outside batches
    HitEvent *events; int *event_count; int event_cap, depth;
    RDR_FN int index(int lane) const {
        if (!seg) return lane;
        const int idx = lane >> 1;
        int s = 0;
        while (s + 1 < S && seg[s + 1] <= idx) ++s;
        return lane + 2 * (s * P0 - seg[s]);
    }
};
struct SecondaryEdgeWeights {
    SceneD sc; const SecondaryEdgeRec *recs; VSlice ev; HitPosView hp;
    RDR_FN void make_lean() { lean_scene(sc); lean_slice(ev); }
    RDR_FN void make_mid() { mid_scene(sc); }
    RDR_FN void scale_lane(const SecondaryEdgeRec &rec, int l) const {
        if (ev.shape[l] < 0) {
            if (sc.envmap != nullptr) {
                // the edge ray reaches the environment light (:1900-1912)
                V3 a = edge_v0(sc.shapes, rec.edge), b = edge_v1(sc.shapes, rec.edge);
                double dirac_j = len(cross(a - rec.sp_pos, b - rec.sp_pos));
                double line_j = 1 / len_sq(rec.edge_pt - rec.sp_pos);
                st3(ev.thr, ev.n, l, 0, ld3(ev.thr, ev.n, l, 0) * (line_j / dirac_j));
            }
            return;
        }
        RayDiff tmp;
        Surf hp = surf_at(sc.shapes[ev.shape[l]], ev.tri[l], load_ray(ev, l), load_rdiff(ev, l), tmp, !sc.no_diffs);
        {
            const int hi = this->hp.index(l);
            st3(this->hp.pos, this->hp.n, hi, 0, hp.position);
            if (this->hp.written) this->hp.written[hi] = 1;
        }
        V3 dir = hp.position - rec.sp_pos;
        double d2 = len_sq(dir);
        if (d2 < 1e-8f) { st3(ev.thr, ev.n, l, 0, v3(0)); return; }
        V3 nd = dir / sqrt(d2);
        double geo = fabs(dot(hp.geom_normal, nd)) / d2;
        V3 jac = isect_jacobian(rec.sp_pos, rec.edge_pt, hp.position, hp.geom_normal, rec.mwt);
        V3 a = edge_v0(sc.shapes, rec.edge), b = edge_v1(sc.shapes, rec.edge);
        V3 hn = normalize(cross(a - rec.sp_pos, b - rec.sp_pos));
        double line_j = len(jac) / len(cross(hp.geom_normal, hn));
        double dirac_j = len(cross(a - rec.sp_pos, b - rec.sp_pos));
        double w = line_j / dirac_j;
        st3(ev.thr, ev.n, l, 0, ld3(ev.thr, ev.n, l, 0) * (geo * w));
    }
    RDR_FN void operator()(int idx) const {
        const SecondaryEdgeRec &rec = recs[idx];
        if (rec.edge.shape_id < 0) return;
        int l0 = 2 * idx, l1 = 2 * idx + 1;
        int light0 = ev.shape[l0] >= 0 ? sc.shapes[ev.shape[l0]].light_id : -1;
        int light1 = ev.shape[l1] >= 0 ? sc.shapes[ev.shape[l1]].light_id : -1;
        bool hit_light = light0 != -1 || light1 != -1;
        if (!hit_light && sc.envmap != nullptr) hit_light = envmap_pdf(*sc.envmap, normalize(rec.edge_pt - rec.sp_pos)) > 0;
        if (rec.use_nee_ray) {
            if (hit_light) {
                st3(ev.thr, ev.n, l0, 0, ld3(ev.thr, ev.n, l0, 0) * 0.5f);
                st3(ev.thr, ev.n, l1, 0, ld3(ev.thr, ev.n, l1, 0) * 0.5f);
            } else {
                st3(ev.thr, ev.n, l0, 0, v3(0));
                st3(ev.thr, ev.n, l1, 0, v3(0));
            }
        } else if (hit_light && rec.diffuse_or_glossy) {
            st3(ev.thr, ev.n, l0, 0, ld3(ev.thr, ev.n, l0, 0) * 0.5f);
            st3(ev.thr, ev.n, l1, 0, ld3(ev.thr, ev.n, l1, 0) * 0.5f);
        }
        scale_lane(rec, l0);
        scale_lane(rec, l1);
    }
};

struct SecondaryEdgeDerivatives {
    SceneD sc; GScene g; const int *active; const SecondaryEdgeRec *recs;
    HitPosView hp; const double *edge_contrib; AdjState adj;
    RDR_FN void operator()(int idx) const {
        const SecondaryEdgeRec &rec = recs[idx];
        if (rec.edge.shape_id < 0) return;
        if (edge_contrib[2 * idx] == 0 && edge_contrib[2 * idx + 1] == 0) return;      // nothing but zeros to add
        int p = active[idx];
        V3 a = edge_v0(sc.shapes, rec.edge), b = edge_v1(sc.shapes, rec.edge);
        V3 dp = v3(0), da = v3(0), db = v3(0);
        for (int k = 0; k < 2; ++k) {
            double contrib = edge_contrib[2 * idx + k];
            if (contrib == 0) continue;
            const int hi = hp.index(2 * idx + k);
            if (hp.written && !hp.written[hi]) {                                        // (see HitPosView)
                const int at = hp.events ? atomic_fetch_add(hp.event_count, 1) : hp.event_cap;
                if (at < hp.event_cap) hp.events[at] = HitEvent{p, hi, rec.edge.shape_id, rec.edge.v0, rec.edge.v1, hp.depth, contrib, rec.sp_pos};
                else atomic_fetch_add(hp.violations, 1);
                continue;
            }
            V3 x = ld3(hp.pos, hp.n, hi, 0);
            V3 pos = rec.sp_pos;
            V3 d0 = a - pos, d1 = b - pos;
            dp += (cross(d1, d0) + cross(x - pos, d1) + cross(d0, x - pos)) * contrib;   // Eq. 16 (errata)
            da += cross(d1, x - pos) * contrib;
            db += cross(x - pos, d0) * contrib;
        }
        // position adjoint of the shading point (slots 0..2 of the adjoint point record)
        adj.point[(size_t)0 * adj.n + p] += dp.x;
        adj.point[(size_t)1 * adj.n + p] += dp.y;
        adj.point[(size_t)2 * adj.n + p] += dp.z;
        if (adj.carries) adj.carries[p] = 1;
        double *gv = g.shapes[rec.edge.shape_id].vertices;
        accum3(gv + 3 * rec.edge.v0, da);
        accum3(gv + 3 * rec.edge.v1, db);
    }
};

// Replay of the recorded stale reads (HitPosView): the position an event's lane would have found in the reference's scratch is
// the latest write of an EARLIER sample of the batch to the same entry -- every sample's final state is known now -- else what
// the batch found there (`carry`).  The event's terms go where SecondaryEdgeDerivatives would have put them: the edge's two
// vertices, and the shading point's adjoint position, from where the replayed adjoint sweep carries it on; `live` marks the
// lanes that sweep has to run.
struct InjectHitEvents {
    SceneD sc; GScene g; const HitEvent *events; int depth;
    const double *pos; int n; const unsigned char *written; const double *carry; int entries_per_sample;      // 2 P0
    AdjState adj; unsigned char *live;
    RDR_FN void operator()(int i) const {
        const HitEvent e = events[i];
        if (e.depth != depth) return;
        const int s = e.entry / entries_per_sample, l = e.entry - s * entries_per_sample;
        V3 x = ld3(carry, entries_per_sample, l, 0);
        for (int t = s - 1; t >= 0; --t)
            if (written[t * entries_per_sample + l]) { x = ld3(pos, n, t * entries_per_sample + l, 0); break; }
        const EdgeD edge{e.shape_id, e.v0, e.v1, 0, 0};
        V3 a = edge_v0(sc.shapes, edge), b = edge_v1(sc.shapes, edge);
        V3 d0 = a - e.sp_pos, d1 = b - e.sp_pos;
        V3 dp = (cross(d1, d0) + cross(x - e.sp_pos, d1) + cross(d0, x - e.sp_pos)) * e.contrib;
        V3 da = cross(d1, x - e.sp_pos) * e.contrib, db = cross(x - e.sp_pos, d0) * e.contrib;
        atomic_add_f64(adj.point + (size_t)0 * adj.n + e.p, dp.x);       // (both lanes of a slot may be events: atomic adds)
        atomic_add_f64(adj.point + (size_t)1 * adj.n + e.p, dp.y);
        atomic_add_f64(adj.point + (size_t)2 * adj.n + e.p, dp.z);
        live[e.p] = 1;
        double *gv = g.shapes[e.shape_id].vertices;
        accum3(gv + 3 * e.v0, da);
        accum3(gv + 3 * e.v1, db);
    }
};
// after a batch: what the next batch finds in the scratch: per entry the write of the batch's last sample that wrote it
struct HitPosCarryAdvance {
    double *carry; int entries_per_sample, S; const double *pos; int n; const unsigned char *written;
    RDR_FN void operator()(int l) const {
        for (int t = S - 1; t >= 0; --t)
            if (written[t * entries_per_sample + l]) { st3(carry, entries_per_sample, l, 0, ld3(pos, n, t * entries_per_sample + l, 0)); return; }
    }
};

} // namespace rdr
