// stages_fwd.h -- forward wavefront stages (one lane = one live path).
//
// What the reference does in seven launches per bounce over a 280-byte-per-vertex AoS PathBuffer
// (src/pathtracer.cpp:240-390: sample_primary_rays, intersect+intersect_shape, accumulate_primary_
// contribs, sample_point_on_light, bsdf_sample, accumulate_path_contribs, ...) is regrouped here
// around the two traversal launches of a bounce:
//     GenPrimary -> [closest] -> ShadePrimary -> compact
//     per bounce:  BounceSample -> [any-hit, closest] -> BounceContrib -> compact
// Per path vertex only {ray, incoming ray differential, hit ids, throughput, min-roughness, one
// occlusion bit} stay in HBM (SoA, 185 B/vertex instead of ~1.7 kB): shading points, light points
// and every random number are *recomputed* from those where needed (the Sobol' stream is
// stateless), trading fp64 flops, which MI355X has, for HBM bytes, which bound this path.
//
// Lanes are addressed like the reference so that sample-exact parity holds: state lives at the
// lane's own id (pixel id for camera paths, edge-ray id for edge sub-paths), the live-lane list is
// order-preserving, and lane `p` draws from Sobol' slot `p >> rng_shift` (shift 1 = the two rays of
// an edge sample share their numbers, src/pathtracer.cpp:630-645 copy_interleave).
#pragma once
#include "bsdf.h"
#include "bvh.h"
#include "camera.h"
#include "lights.h"
#include "envmap.h"
#include "sobol.h"

namespace rdr {

// ---- per-vertex state, struct-of-arrays with stride n -------------------------------------------
struct VSlice {
    double *ray;      // 6 x n : org.xyz, dir.xyz of the ray arriving at this vertex
    double *rdiff;    // 12 x n: differential carried by that ray (before transfer onto the surface)
    int *shape, *tri; // hit ids of this vertex (shape < 0: no hit)
    double *thr;      // 3 x n : path throughput arriving here
    double *mrough;   // n     : running minimum roughness
    unsigned char *occl;   // n: the NEE shadow ray cast from this vertex was blocked
    int n;
    // Edge sub-path slices only, and only when some texture has mip levels (otherwise null): the
    // reference's single in-place `edge_ray_differentials` buffer (src/pathtracer.cpp:60), 12 x n.
    // It is written per slot by the primary edge sampler but read per lane (src/edge.cpp:608 vs
    // src/scene.cpp:585), so a primary edge ray starts from whatever differential an earlier stage
    // left at its lane index.  Stages mirror every write the reference makes so that those stale
    // reads see the same values.  [quirk]
    double *erd;
    // Sample batches of such scenes ("chain mode", render.cpp): the buffer belongs to one sample at a time in the reference, so
    // the batch keeps one copy per sample -- entry l of sample s at s * 2 P0 + l, which is where the primary-edge lanes of a
    // batch live anyway.  A secondary-edge pass of a batch numbers its lanes 2 x (rank in the batch's concatenated live-lane
    // list) + side: `erd_seg` (where each sample's part of that list starts) maps them to the sample's own entries.
    // `erd_touched` marks the entries written during the current batch: a primary-edge lane whose entry is not marked would
    // read what EARLIER SAMPLES left there and is handed to the sequential part of the pass (render.cpp).
    unsigned char *erd_touched;
    const int *erd_seg; int erd_S, erd_P0;
};
RDR_FN void st_rdiff(double *b, int n, int i, const RayDiff &r);
RDR_FN RayDiff ld_rdiff(const double *b, int n, int i);
RDR_FN int erd_index(const VSlice &v, int p) {
    if (!v.erd_seg) return p;
    const int idx = p >> 1;
    int s = 0;
    while (s + 1 < v.erd_S && v.erd_seg[s + 1] <= idx) ++s;          // <= 16 samples per batch
    return p + 2 * (s * v.erd_P0 - v.erd_seg[s]);
}
RDR_FN RayDiff ld_erd(const VSlice &v, int p) { return ld_rdiff(v.erd, v.n, erd_index(v, p)); }
RDR_FN void st_erd(const VSlice &v, int p, const RayDiff &r) {
    const int i = erd_index(v, p);
    st_rdiff(v.erd, v.n, i, r);
    if (v.erd_touched) v.erd_touched[i] = 1;
}

RDR_FN V3 ld3(const double *b, int n, int i, int k) { return V3{b[(size_t)(k) * n + i], b[(size_t)(k + 1) * n + i], b[(size_t)(k + 2) * n + i]}; }
RDR_FN void st3(double *b, int n, int i, int k, V3 v) { b[(size_t)(k) * n + i] = v.x; b[(size_t)(k + 1) * n + i] = v.y; b[(size_t)(k + 2) * n + i] = v.z; }

RDR_FN Ray load_ray(const VSlice &v, int i) { return make_ray(ld3(v.ray, v.n, i, 0), ld3(v.ray, v.n, i, 3)); }
RDR_FN void store_ray(const VSlice &v, int i, V3 o, V3 d) { st3(v.ray, v.n, i, 0, o); st3(v.ray, v.n, i, 3, d); }
RDR_FN RayDiff load_rdiff(const VSlice &v, int i) {
    if (!v.rdiff) return raydiff_zero();        // lean stages: differentials are neither stored nor used
    return RayDiff{ld3(v.rdiff, v.n, i, 0), ld3(v.rdiff, v.n, i, 3), ld3(v.rdiff, v.n, i, 6), ld3(v.rdiff, v.n, i, 9)};
}
RDR_FN void st_rdiff(double *b, int n, int i, const RayDiff &r) {
    st3(b, n, i, 0, r.org_dx); st3(b, n, i, 3, r.org_dy);
    st3(b, n, i, 6, r.dir_dx); st3(b, n, i, 9, r.dir_dy);
}
RDR_FN RayDiff ld_rdiff(const double *b, int n, int i) {
    return RayDiff{ld3(b, n, i, 0), ld3(b, n, i, 3), ld3(b, n, i, 6), ld3(b, n, i, 9)};
}
RDR_FN void store_rdiff(const VSlice &v, int i, const RayDiff &r) { if (v.rdiff) st_rdiff(v.rdiff, v.n, i, r); }

RDR_FN rt::RayRec ray_rec(const Ray &r, bool dead) {
    rt::RayRec rec;
    rec.ox = (float)r.org.x; rec.oy = (float)r.org.y; rec.oz = (float)r.org.z; rec.tmin = (float)r.tmin;
    rec.dx = (float)r.dir.x; rec.dy = (float)r.dir.y; rec.dz = (float)r.dir.z;
    rec.tmax = dead ? -1.f : (float)r.tmax;
    return rec;
}
RDR_FN void put_ray(rt::RayRec *q, int slot, const Ray &r, bool dead) { q[slot] = ray_rec(r, dead); }

// Where a stage deposits radiance: the image (camera paths) and/or a per-lane scalar (edge paths).
constexpr int kMaxChannels = 16;
constexpr int kMaxGeneric = 16;      // widest generic texture the G-buffer channels accept

// Output channel list (src/channels.h:6-37): ids in request order, nd = total floats per pixel,
// radiance_dim = offset of the radiance triple (-1 if not requested).
// radiance_dim is the reference's `radiance_dimension`: the INDEX of the radiance channel in the channel list,
// which path contributions and the edge estimators use as a dimension offset (src/channels.cpp:27,
// src/path_contribution.cpp:126, src/edge.cpp:451) -- equal to the true offset only while every channel
// before radiance is one float wide.  The first-hit emission uses the true offset, radiance_off.  [quirk]
// The ids live in device memory, not in the functor: a run-time index into a functor member would force the whole
// (kilobyte-sized) functor out of the kernel-argument segment into per-lane scratch.
struct ChannelsD { int n; const int *id; int radiance_only; int nd, radiance_dim, radiance_off, max_generic; };

// Where a stage deposits its result: the image (camera paths) and/or a per-lane scalar (edge paths).
struct Sink {
    float *image;          // [num_pixels * nd], may be null
    double *edge_contrib;  // [lanes], may be null
    int nd, radiance_dim;  // channel layout (src/channels.cpp)
    double weight;         // 1 / spp
    ChannelsD ch;
    const double *multipliers;   // per lane x nd weights of the primary-edge estimator, or null
};

// ---- lean specialisation ---------------------------------------------------------------------------
// Plain scenes -- pinhole camera without lens distortion, no environment light, constant reflectances (no image
// textures, no normal maps), radiance only -- are what optimisation loops over geometry mostly render.  Without
// mip levels and environment lookups ray differentials have no effect at all.
// For them the host launches LeanStage<Stage>: the stage's scene copy gets those facts written in as constants,
// so after inlining the compiler drops the other camera models, the environment-light estimators and the
// G-buffer channels -- and with them the registers (and scratch) those paths would pin.
RDR_FN void lean_scene(SceneD &sc) { sc.envmap = nullptr; sc.cam.kind = kCamPerspective; sc.cam.distortion.defined = 0; sc.no_diffs = 1; sc.plain_materials = 1; }
RDR_FN void lean_slice(VSlice &v) { v.rdiff = nullptr; v.erd = nullptr; v.erd_touched = nullptr; v.erd_seg = nullptr; }
// "Mid" specialisation: pinhole camera without lens distortion, no environment light, radiance-only output -- but image
// textures, mip levels (hence ray differentials), normal maps and vertex colours are all live.  What a textured scene lit by
// area lights renders (BASELINE config 5's stand-in); the general stages carry three more camera models, the environment
// estimators and the G-buffer code and need every register plus AGPR spills for it.
RDR_FN void mid_scene(SceneD &sc) { sc.envmap = nullptr; sc.cam.kind = kCamPerspective; sc.cam.distortion.defined = 0; }
RDR_FN void lean_channels(ChannelsD &ch) { ch.n = 1; ch.radiance_only = 1; ch.radiance_dim = 0; ch.radiance_off = 0; ch.nd = 3; }
template <class Stage, class = void> struct LeanBlocks { static constexpr int value = 1; };
template <class Stage> struct LeanBlocks<Stage, decltype((void)Stage::kLeanBlocksPerCU)> { static constexpr int value = Stage::kLeanBlocksPerCU; };
template <class Stage> struct LeanStage {
    static constexpr int kMinBlocksPerCU = LeanBlocks<Stage>::value;
    Stage f;
    RDR_FN void operator()(int i) const { Stage g = f; g.make_lean(); RDR_INLINE_CALL g(i); }
};
template <class Stage, class = void> struct MidBlocks { static constexpr int value = 1; };
template <class Stage> struct MidBlocks<Stage, decltype((void)Stage::kMidBlocksPerCU)> { static constexpr int value = Stage::kMidBlocksPerCU; };
template <class Stage> struct MidStage {
    static constexpr int kMinBlocksPerCU = MidBlocks<Stage>::value;
    Stage f;
    RDR_FN void operator()(int i) const { Stage g = f; g.make_mid(); RDR_INLINE_CALL g(i); }
};
// The same for resumable walks (exec::launch_persistent).
template <class Walk> struct LeanWalk {
    Walk w;
    using State = typename Walk::State;
    RDR_DEV_FN bool begin(int i, State &st) const { Walk g = w; g.make_lean(); return g.begin(i, st); }
    RDR_DEV_FN bool step(State &st) const { Walk g = w; g.make_lean(); return g.step(st); }
    RDR_DEV_FN void finish(State &st) const { Walk g = w; g.make_lean(); g.finish(st); }
    RDR_DEV_FN bool gate_closed() const { return w.gate_closed(); }
};
template <class Walk> struct MidWalk {
    Walk w;
    using State = typename Walk::State;
    RDR_DEV_FN bool begin(int i, State &st) const { Walk g = w; g.make_mid(); return g.begin(i, st); }
    RDR_DEV_FN bool step(State &st) const { Walk g = w; g.make_mid(); return g.step(st); }
    RDR_DEV_FN void finish(State &st) const { Walk g = w; g.make_mid(); g.finish(st); }
    RDR_DEV_FN bool gate_closed() const { return w.gate_closed(); }
};

struct LightDraw { double light_sel, tri_sel; V2 uv; };
RDR_FN LightDraw draw_light(const SamplerD &rng, const SamplerD::Lane &ln, int dim) {
    return LightDraw{rng.draw(ln, dim), rng.draw(ln, dim + 1), v2(rng.draw(ln, dim + 2), rng.draw(ln, dim + 3))};
}
RDR_FN LightDraw draw_light(const SamplerD &rng, int slot, int dim) { return draw_light(rng, rng.lane(slot), dim); }

// ---- stage: camera rays -------------------------------------------------------------------------
struct GenPrimary {
    SceneD sc; SamplerD rng; int sample_center;
    VSlice v0; rt::RayRec *q;
    RDR_FN void make_lean() { lean_scene(sc); lean_slice(v0); }
    RDR_FN void make_mid() { mid_scene(sc); }
    RDR_FN void operator()(int p) const {
        V2 s = v2(0.5, 0.5);
        if (!sample_center) { const SamplerD::Lane ln = rng.lane(p); s = v2(rng.draw(ln, 0), rng.draw(ln, 1)); }
        RayDiff rd = raydiff_zero();
        Ray r = sc.no_diffs ? primary_ray(sc.cam, pixel_to_screen(sc.cam, p, s))
                            : primary_ray_with_diff(sc.cam, pixel_to_screen(sc.cam, p, s), rd);
        store_ray(v0, p, r.org, r.dir);
        store_rdiff(v0, p, rd);
        st3(v0.thr, v0.n, p, 0, v3(1));
        v0.mrough[p] = 0;
        put_ray(q, p, r, false);
    }
};

// Emission seen directly along `ray` at a hit (src/primary_contribution.cpp:13-31).
RDR_FN V3 direct_emission(const SceneD &sc, int shape, int tri, const Ray &ray, const RayDiff &rd) {
    V3 e = v3(0);
    if (shape < 0) {
        // a ray without a direction (a fisheye pixel outside the image circle) is not an active pixel in the reference
        // (src/active_pixels.cpp:8-28: init_active_pixels drops it before anything is accumulated): it sees nothing
        if (sc.envmap != nullptr && sc.envmap->directly_visible && len_sq(ray.dir) > 0) e = envmap_eval(*sc.envmap, ray.dir, rd);
        return e;
    }
    const ShapeD &sh = sc.shapes[shape];
    if (sh.light_id >= 0) {
        const LightD &l = sc.lights[sh.light_id];
        if (l.directly_visible) {
            bool facing = true;
            if (!l.two_sided) {
                RayDiff tmp;
                Surf sp = surf_at(sh, tri, ray, rd, tmp, !sc.no_diffs);
                facing = dot(-ray.dir, sp.frame.n) > 0;
            }
            if (facing) e += v3f(l.intensity);
        }
    }
    return e;
}

// ---- stage: first-hit contribution -------------------------------------------------------------
// Lanes: `active[idx]` (null = identity).  Records the hit ids of vertex `v` from queue slot idx.
// Value of one non-radiance G-buffer channel at a first hit (src/primary_contribution.cpp:45-253).
// Returns the number of components written to val (0: channel produces nothing here).
RDR_FN int channel_value(const SceneD &sc, int channel, const ShapeD &sh, const Surf &sp, const Ray &ray,
                         int shape_id, int tri_id, int max_generic, double *val) {
    const MaterialD &m = sc.materials[sh.material_id];
    switch (channel) {
        case 1: val[0] = 1; return 1;                                                   // alpha
        case 2: val[0] = len(sp.position - ray.org); return 1;                          // depth
        case 3: val[0] = sp.position.x; val[1] = sp.position.y; val[2] = sp.position.z; return 3;
        case 4: val[0] = sp.geom_normal.x; val[1] = sp.geom_normal.y; val[2] = sp.geom_normal.z; return 3;
        case 5: {
            V3 n = sp.frame.n;
            if (has_normal_map(m)) n = perturbed_frame(m, sp).n;
            val[0] = n.x; val[1] = n.y; val[2] = n.z; return 3;
        }
        case 6: val[0] = sp.uv.x; val[1] = sp.uv.y; return 2;
        case 7: val[0] = sp.bary.x; val[1] = sp.bary.y; return 2;
        case 8: { V3 r = m.use_vertex_color ? sp.color : tex3(m.diffuse, sp); val[0] = r.x; val[1] = r.y; val[2] = r.z; return 3; }
        case 9: { V3 r = tex3(m.specular, sp); val[0] = r.x; val[1] = r.y; val[2] = r.z; return 3; }
        case 10: val[0] = tex1(m.roughness, sp); return 1;
        case 11: {
            if (m.generic.num_levels <= 0) return 0;
            tex_fetch(m.generic, sp.uv, sp.du_dxy, sp.dv_dxy, val);
            return m.generic.channels;
        }
        case 12: val[0] = sp.color.x; val[1] = sp.color.y; val[2] = sp.color.z; return 3;
        case 13: val[0] = shape_id; return 1;
        case 14: val[0] = tri_id; return 1;
        case 15: val[0] = sh.material_id; return 1;
        default: return 0;
    }
}
RDR_FN int channel_width(int channel, int max_generic) {
    switch (channel) {
        case 0: case 3: case 4: case 5: case 8: case 9: case 12: return 3;
        case 6: case 7: return 2;
        case 11: return max_generic;
        default: return 1;
    }
}

// First-hit contribution of lane p whose hit ids are (shape, tri): radiance (emission) plus every
// requested G-buffer channel.  accumulate_primary_contribs, src/primary_contribution.cpp:6-436.
RDR_FN void shade_first_hit(const SceneD &sc, const Sink &sink, const VSlice &v, int p, int shape, int tri) {
    Ray ray = load_ray(v, p);
    RayDiff rd = load_rdiff(v, p);
    V3 e = direct_emission(sc, shape, tri, ray, rd);
    V3 c = sink.weight * ld3(v.thr, v.n, p, 0) * e;
    bool only_radiance = sink.ch.radiance_only != 0;
    if (only_radiance) {
        if (sink.image) {
            float *px = sink.image + (size_t)sink.nd * p + sink.radiance_dim;
            px[0] += float(c.x); px[1] += float(c.y); px[2] += float(c.z);
        }
        if (sink.edge_contrib) sink.edge_contrib[p] += sum(c);
        return;
    }
    Surf sp = surf_zero();
    if (shape >= 0) { RayDiff tmp; sp = surf_at(sc.shapes[shape], tri, ray, rd, tmp, !sc.no_diffs); }
    int d = 0;
    for (int k = 0; k < sink.ch.n; ++k) {
        int id = sink.ch.id[k];
        int width = channel_width(id, sink.ch.max_generic);
        if (id == 0) {
            if (sink.image) {
                float *px = sink.image + (size_t)sink.nd * p + d;
                px[0] += float(c.x); px[1] += float(c.y); px[2] += float(c.z);
            }
            if (sink.edge_contrib) sink.edge_contrib[p] += sum(c);
        } else if (shape >= 0) {
            double val[kMaxGeneric];
            int nv = channel_value(sc, id, sc.shapes[shape], sp, ray, shape, tri, sink.ch.max_generic, val);
            bool is_id = id >= 13;
            if (sink.image) {
                float *px = sink.image + (size_t)sink.nd * p + d;
                for (int j = 0; j < nv; ++j) {
                    if (is_id) { px[j] = float(val[j]); continue; }      // ids: last sample wins
                    double x = val[j] * sink.weight;
                    if (sink.multipliers) x *= sink.multipliers[(size_t)sink.nd * p + d + j];
                    px[j] += float(x);
                }
            }
            if (sink.edge_contrib && sink.multipliers && !is_id) {
                // [quirk] the edge estimator ignores the normal map for this channel (primary_contribution.cpp:306-316)
                if (id == 5) { val[0] = sp.frame.n.x; val[1] = sp.frame.n.y; val[2] = sp.frame.n.z; }
                double acc = 0;
                for (int j = 0; j < nv; ++j) acc += (val[j] * sink.weight) * sink.multipliers[(size_t)sink.nd * p + d + j];
                sink.edge_contrib[p] += acc;
            }
        }
        d += width;
    }
}

struct ShadePrimary {
    SceneD sc; const int *active; VSlice v; const rt::HitRec *hits; Sink sink;
    RDR_FN void make_lean() { lean_scene(sc); lean_slice(v); lean_channels(sink.ch); sink.multipliers = nullptr; }
    RDR_FN void make_mid() { mid_scene(sc); lean_channels(sink.ch); sink.multipliers = nullptr; }
    RDR_FN void operator()(int idx) const {
        int p = active ? active[idx] : idx;
        rt::HitRec h = hits[idx];
        v.shape[p] = h.shape; v.tri[p] = h.shape >= 0 ? h.prim : -1;
        shade_first_hit(sc, sink, v, p, h.shape, h.shape >= 0 ? h.prim : -1);
    }
};

// Everything a bounce needs at its shading vertex, rebuilt from the stored ids.
struct VertexCtx {
    Ray ray; RayDiff rd_in, rd_surf; Surf sp; V3 wi; double mrough;
    const ShapeD *shape; const MaterialD *mat;
};
RDR_FN VertexCtx load_vertex(const SceneD &sc, const VSlice &v, int p) {
    VertexCtx c;
    c.ray = load_ray(v, p);
    c.rd_in = load_rdiff(v, p);
    c.shape = &sc.shapes[v.shape[p]];
    c.mat = &sc.materials[c.shape->material_id];
    c.sp = surf_at(*c.shape, v.tri[p], c.ray, c.rd_in, c.rd_surf, !sc.no_diffs);
    c.sp.plain = sc.plain_materials;
    c.wi = -c.ray.dir;
    c.mrough = v.mrough[p];
    return c;
}

// Does the ray, by the traversal kernels' own triangle rule (raytri.h; same fp32 ray record, same fp32 corners), meet ANY triangle
// of an area light within (tmin, tmax)?  A superset of "its closest hit lies on an emitter": the closest-hit query reports an
// emitter triangle only if this test passes for that triangle.  For scenes with a handful of emitter triangles (render.cpp decides).
RDR_FN bool ray_meets_emitter_triangle(const SceneD &sc, const rt::RayRec &r) {
    const float o[3] = {r.ox, r.oy, r.oz}, d[3] = {r.dx, r.dy, r.dz};
    for (int l = 0; l < sc.num_area_lights; ++l) {
        const ShapeD &sh = sc.shapes[sc.lights[l].shape_id];
        for (int t = 0; t < sh.num_triangles; ++t) {
            float a[3], b[3], c[3];
            if (sh.geom) {
                const TriGeomD g = load_dev(sh.geom + t);
                for (int k = 0; k < 3; ++k) { a[k] = g.p[k]; b[k] = g.p[3 + k]; c[k] = g.p[6 + k]; }
            } else {
                const int i0 = sh.indices[3 * t], i1 = sh.indices[3 * t + 1], i2 = sh.indices[3 * t + 2];
                for (int k = 0; k < 3; ++k) { a[k] = sh.vertices[3 * i0 + k]; b[k] = sh.vertices[3 * i1 + k]; c[k] = sh.vertices[3 * i2 + k]; }
            }
            float tt;
            if (rt::ray_triangle(o, d, r.tmin, r.tmax, a, b, c, &tt)) return true;
        }
    }
    return false;
}

// ---- stage: draw the NEE point and the BSDF direction, emit both rays ---------------------------
struct BounceSample {
    static constexpr int kMidBlocksPerCU = 3;
    static constexpr int kMinBlocksPerCU = 3;          // general form
    SceneD sc; SamplerD rng; int dim, rng_shift;
    RDR_FN void make_lean() { lean_scene(sc); lean_slice(v); lean_slice(vn); }
    RDR_FN void make_mid() { mid_scene(sc); }
    const int *active; VSlice v, vn;
    rt::RayRec *q_nee, *q_bsdf;
    // The path's LAST bounce (set by the host, plain scenes with a handful of emitter triangles): the vertex the continuation ray
    // reaches is never shaded -- it matters only as an emitter seen through the BSDF (eval_bounce's light branch and its adjoint).
    // A ray that meets no emitter triangle at all is queued dead: the query answers "no hit" without a traversal, which is what
    // every consumer makes of a hit on a non-emitter there.  The others are traced as ever (an occluder may still come first).
    int last_bounce_emitters = 0;
    RDR_FN void operator()(int idx) const {
        int p = active[idx];
        VertexCtx c = load_vertex(sc, v, p);
        RDR_INLINE_CALL sample_from(c, p, idx);
    }
    // the bounce of lane p from its vertex `c`; the two rays go to queue slot idx
    RDR_FN void sample_from(const VertexCtx &c, int p, int idx) const {
        int slot = p >> rng_shift;
        const SamplerD::Lane ln = rng.lane(slot);            // the lane's seven numbers share this
        // next-event estimation ray
        LightDraw ld = draw_light(rng, ln, dim);
        LightPick pk = pick_light(sc, ld.light_sel, ld.tri_sel);
        if (pk.shape_id >= 0) {
            const ShapeD &lsh = sc.shapes[pk.shape_id];
            Surf lp = sample_tri(lsh, pk.tri_id, ld.uv);
            // A shadow ray whose answer cannot matter is not traced: when the light faces away, or the direction
            // fails one of bsdf_eval's geometric early-outs, the estimator and its adjoint are exactly zero whether
            // the segment is blocked or not (eval_bounce / AdjBounceNee / adj_bsdf_eval test the same conditions on
            // the same values).  About a third of the shadow rays of the Cornell-box benchmark are of this kind.
            bool moot = false;
            {
                V3 dir = lp.position - c.sp.position;
                double d2 = len_sq(dir);
                if (d2 > 1e-20f && lsh.light_id >= 0) {
                    V3 wo = dir / sqrt(d2);
                    const LightD &l = sc.lights[lsh.light_id];
                    if (!(l.two_sided || dot(-wo, lp.frame.n) > 0)) moot = true;
                    ShadeCtx sx = shade_ctx(*c.mat, c.sp);
                    double gwi = dot(sx.gn, c.wi), gwo = dot(sx.gn, wo);
                    double swi = fabs(dot(sx.fr.n, c.wi)), swo = fabs(dot(sx.fr.n, wo));
                    if (gwi * gwo < 0) moot = true;
                    if (!c.mat->two_sided && gwi < 0 && gwo < 0) moot = true;
                    if (swi == 0 || swo <= 1e-3f || fabs(gwo) <= 1e-3f) moot = true;
                }
            }
            put_ray(q_nee, idx, shadow_ray_to(c.sp.position, lp.position), moot);
        } else if (sc.envmap != nullptr) {
            Ray er = make_ray(c.sp.position, envmap_sample(*sc.envmap, ld.uv));      // tmin 1e-3f, tmax inf (:709-711)
            put_ray(q_nee, idx, er, false);
        } else {
            Ray dead = make_ray(c.sp.position, v3(0));
            put_ray(q_nee, idx, dead, true);
        }
        // BSDF ray
        V2 buv = v2(rng.draw(ln, dim + 4), rng.draw(ln, dim + 5));
        double bw = rng.draw(ln, dim + 6);
        // a sampler that bails out (one-sided surface seen from behind) leaves the differential untouched
        RayDiff wo_rd = vn.erd ? ld_erd(vn, p) : raydiff_zero();
        double next_mr;
        V3 dir = bsdf_sample_dir(*c.mat, c.sp, c.wi, buv, bw, c.mrough, c.rd_surf, wo_rd, next_mr, !sc.no_diffs);
        store_ray(vn, p, c.sp.position, dir);
        store_rdiff(vn, p, wo_rd);
        if (vn.erd) st_erd(vn, p, wo_rd);
        vn.mrough[p] = next_mr;
        Ray nr = make_ray(c.sp.position, dir);
        bool dead = len_sq(dir) <= 1e-3f;
        if (last_bounce_emitters && !dead) dead = !ray_meets_emitter_triangle(sc, ray_rec(nr, false));
        put_ray(q_bsdf, idx, nr, dead);
    }
};

// Result of evaluating one bounce (shared by the forward and the adjoint stage).
struct BounceEval {
    V3 nee, scatter;        // contributions before multiplying the throughput
    V3 next_thr; bool next_thr_valid;
};

// NEE + BSDF-hit emission with power-2 MIS (src/path_contribution.cpp:24-118).
RDR_FN BounceEval eval_bounce(const SceneD &sc, const VertexCtx &c, V3 thr,
                              bool nee_visible, const LightPick &pk, const Surf &lp, V2 light_uv,
                              int bshape, const Surf &bp, V3 bsdf_dir) {
    BounceEval r;
    r.nee = r.scatter = r.next_thr = v3(0);
    r.next_thr_valid = false;
    V3 pos = c.sp.position;
    if (nee_visible && pk.shape_id >= 0) {
        const ShapeD &lsh = sc.shapes[pk.shape_id];
        V3 dir = lp.position - pos;
        double d2 = len_sq(dir);
        V3 wo = dir / sqrt(d2);
        if (d2 > 1e-20f && lsh.light_id >= 0) {
            const LightD &l = sc.lights[lsh.light_id];
            if (l.two_sided || dot(-wo, lp.frame.n) > 0) {
                V3 f = bsdf_eval(*c.mat, c.sp, c.wi, wo, c.mrough);
                double g = fabs(dot(wo, lp.geom_normal)) / d2;
                double pdf_nee = sc.light_pmf[lsh.light_id] / sc.light_areas[lsh.light_id];
                double pdf_b = bsdf_pdf(*c.mat, c.sp, c.wi, wo, c.mrough) * g;
                double mis = 1 / (1 + sq(pdf_b / pdf_nee));
                r.nee = (mis * g / pdf_nee) * f * v3f(l.intensity);
            }
        }
    } else if (nee_visible && sc.envmap != nullptr) {
        // environment light: wo is the sampled direction itself, no geometry term (:51-70)
        V3 wo = envmap_sample(*sc.envmap, light_uv);
        double pdf_nee = envmap_pdf(*sc.envmap, wo) * sc.light_pmf[sc.num_lights - 1];
        if (pdf_nee > 0) {
            V3 f = bsdf_eval(*c.mat, c.sp, c.wi, wo, c.mrough);
            V3 Le = envmap_eval(*sc.envmap, wo, raydiff_zero());
            double pdf_b = bsdf_pdf(*c.mat, c.sp, c.wi, wo, c.mrough);
            double mis = 1 / (1 + sq(pdf_b / pdf_nee));
            r.nee = (mis / pdf_nee) * f * Le;
        }
    }
    if (bshape >= 0) {
        const ShapeD &bsh = sc.shapes[bshape];
        V3 dir = bp.position - pos;
        double d2 = len_sq(dir);
        V3 wo = dir / sqrt(d2);
        double pdf_b = bsdf_pdf(*c.mat, c.sp, c.wi, wo, c.mrough);
        r.next_thr_valid = true;
        if (d2 > 1e-20f && pdf_b > 1e-20f) {
            V3 f = bsdf_eval(*c.mat, c.sp, c.wi, wo, c.mrough);
            if (bsh.light_id >= 0) {
                const LightD &l = sc.lights[bsh.light_id];
                if (l.two_sided || dot(-wo, bp.frame.n) > 0) {
                    double inv_area = 1 / sc.light_areas[bsh.light_id];
                    double g = fabs(dot(wo, bp.geom_normal)) / d2;
                    double pdf_nee = (sc.light_pmf[bsh.light_id] * inv_area) / g;
                    double mis = 1 / (1 + sq(pdf_nee / pdf_b));
                    r.scatter = (mis / pdf_b) * f * v3f(l.intensity);
                }
            }
            r.next_thr = thr * (f / pdf_b);
        } else {
            r.next_thr = v3(0);
        }
    } else if (sc.envmap != nullptr) {
        // the BSDF-sampled ray left the scene: it sees the environment light (:99-118)
        V3 wo = bsdf_dir;
        double pdf_b = bsdf_pdf(*c.mat, c.sp, c.wi, wo, c.mrough);
        if (len_sq(wo) > 0 && pdf_b > 1e-20f) {
            V3 f = bsdf_eval(*c.mat, c.sp, c.wi, wo, c.mrough);
            V3 Le = envmap_eval(*sc.envmap, wo, raydiff_zero());
            double pdf_nee = envmap_pdf(*sc.envmap, wo) * sc.light_pmf[sc.num_lights - 1];
            double mis = 1 / (1 + sq(pdf_nee / pdf_b));
            r.scatter = (mis / pdf_b) * f * Le;
        }
    }
    return r;
}

// Next-event estimation towards an area light: is this vertex / light-sample pair one of the cases in which the estimator AND
// every adjoint of it are exactly zero for geometric reasons -- the light faces away, the sample lies below the horizon of the
// shading or the geometric normal (bsdf_eval's early-outs)?  ONE definition, used by BounceContrib (it folds the answer into
// the slice's occlusion byte) and by AdjBounceNee (src/path_contribution.cpp:206-294 evaluates these as it goes): the adjoint
// stage then runs over the compacted list of lanes whose byte is clear (render.cpp) -- full waves instead of 0.6 of a wave.
RDR_FN bool nee_is_geometrically_zero(const SceneD &sc, const VertexCtx &c, const LightPick &pk, const Surf &lp) {
    if (pk.shape_id < 0) return sc.envmap == nullptr;          // environment light: never decided here
    const ShapeD &lsh = sc.shapes[pk.shape_id];
    if (lsh.light_id < 0) return true;
    const V3 dir = lp.position - c.sp.position;
    const double d2 = len_sq(dir);
    const V3 wo = dir / sqrt(d2);
    const LightD &l = sc.lights[lsh.light_id];
    if (!(l.two_sided || dot(-wo, lp.frame.n) > 0)) return true;
    const ShadeCtx sx = shade_ctx(*c.mat, c.sp);
    const double gwi = dot(sx.gn, c.wi), gwo = dot(sx.gn, wo);
    const double swi = fabs(dot(sx.fr.n, c.wi)), swo = fabs(dot(sx.fr.n, wo));
    if (gwi * gwo < 0) return true;
    if (!c.mat->two_sided && gwi < 0 && gwo < 0) return true;
    if (swi == 0 || swo <= 1e-3f || fabs(gwo) <= 1e-3f) return true;
    return false;
}

// ---- stage: gather both query results, accumulate the bounce, advance the throughput ------------
struct BounceContrib {
    static constexpr int kMidBlocksPerCU = 3;
    static constexpr int kMinBlocksPerCU = 3;          // general form
    SceneD sc; SamplerD rng; int dim, rng_shift;
    const int *active; VSlice v, vn;
    const rt::HitRec *h_nee, *h_bsdf;
    Sink sink;
    RDR_FN void make_lean() { lean_scene(sc); lean_slice(v); lean_slice(vn); lean_channels(sink.ch); }
    RDR_FN void make_mid() { mid_scene(sc); lean_channels(sink.ch); }
    RDR_FN void operator()(int idx) const {
        VertexCtx next;
        RDR_INLINE_CALL contrib(active[idx], idx, next);
    }
    // the bounce of lane p whose query results sit in queue slot qi; `next`: the vertex the continuation ray reached (valid
    // when the function returns true) -- what load_vertex(sc, vn, p) would rebuild from the slice: BounceContribSample goes on
    // from it without the reload
    RDR_FN bool contrib(int p, int idx, VertexCtx &next) const {
        int slot = p >> rng_shift;
        // both query results are fetched with the lane's first loads (further down they would wait behind the branches of the
        // light pick: the compiler does not move a load across a branch)
        const rt::HitRec hn = h_nee[idx], hb = h_bsdf[idx];
        const V3 thr = ld3(v.thr, v.n, p, 0);
        VertexCtx c = load_vertex(sc, v, p);
        LightDraw ld = draw_light(rng, slot, dim);
        LightPick pk = pick_light(sc, ld.light_sel, ld.tri_sel);
        Surf lp = surf_zero();
        if (pk.shape_id >= 0) lp = sample_tri(sc.shapes[pk.shape_id], pk.tri_id, ld.uv);
        bool blocked = hn.shape >= 0;
        // the byte AdjBounceNee skips a lane by: shadow ray blocked, or nothing to differentiate for geometric reasons
        if (v.occl) v.occl[p] = (blocked || nee_is_geometrically_zero(sc, c, pk, lp)) ? 1 : 0;
        vn.shape[p] = hb.shape; vn.tri[p] = hb.shape >= 0 ? hb.prim : -1;
        Surf bp = surf_zero();
        if (hb.shape >= 0) {
            RayDiff tmp;
            next.ray = load_ray(vn, p);
            next.rd_in = load_rdiff(vn, p);
            bp = surf_at(sc.shapes[hb.shape], hb.prim, next.ray, next.rd_in, tmp, !sc.no_diffs);
            if (vn.erd) st_erd(vn, p, tmp);
            next.rd_surf = tmp;
            next.shape = &sc.shapes[hb.shape];
            next.mat = &sc.materials[next.shape->material_id];
            next.sp = bp; next.sp.plain = sc.plain_materials;
            next.wi = -next.ray.dir;
            next.mrough = vn.mrough[p];
        }
        // (the camera paths of a gradient render feed no image and no edge estimate: their next-event term is consumed by nobody
        //  -- the adjoint stages re-evaluate it -- and is not evaluated; throughput and occlusion byte are what they leave behind)
        const bool consumed = sink.image != nullptr || sink.edge_contrib != nullptr;
        BounceEval e = eval_bounce(sc, c, thr, !blocked && consumed, pk, lp, ld.uv, hb.shape, bp, load_ray(vn, p).dir);
        if (e.next_thr_valid) st3(vn.thr, vn.n, p, 0, e.next_thr);
        V3 pc = thr * (e.nee + e.scatter);
        if (sink.image) {
            float *px = sink.image + (size_t)sink.nd * p + sink.radiance_dim;
            px[0] += float(sink.weight * pc.x); px[1] += float(sink.weight * pc.y); px[2] += float(sink.weight * pc.z);
        }
        if (sink.edge_contrib) sink.edge_contrib[p] += sum(sink.weight * pc);
        return hb.shape >= 0;
    }
};

// ---- stage: BounceContrib of bounce d and BounceSample of bounce d + 1 in one launch (round 4) ------------------------------
// The vertex a lane's continuation ray reached is rebuilt by BounceContrib (hit triangle -> surface point) and would be
// rebuilt again, from the slice, by the next bounce's BounceSample: here the lane goes straight on and draws its next two rays
// from the vertex it has in registers -- one launch (and one launch tail) per bounce less, no reload of ray / ids / triangle
// record.  The lanes are those of bounce d's list; a lane whose ray missed leaves two dead queue slots.  The rays of bounce
// d + 1 therefore sit at the positions of bounce d's list: `qpos` (from the compaction that made this list) says where a
// lane's query results are.  Same arithmetic on the same operands as the two stages: bit-identical results.  Sobol' only: the
// PCG sampler's states advance between the bounces (render.cpp).
struct BounceContribSample {
    static constexpr int kMidBlocksPerCU = 3;
    static constexpr int kMinBlocksPerCU = 3;
    BounceContrib c; BounceSample s;          // s: v = c.vn, vn = the slice after it, dim = c.dim + 7 (its `active` is not used)
    const int *qpos;                          // queue slot of list position idx (null: idx itself)
    int contrib_only = 0;                     // the chain's last bounce: nothing follows
    RDR_FN void make_lean() { c.make_lean(); s.make_lean(); }
    RDR_FN void make_mid() { c.make_mid(); s.make_mid(); }
    RDR_FN void operator()(int idx) const {
        const int p = c.active[idx];
        VertexCtx next;
        bool hit;
        RDR_INLINE_CALL hit = c.contrib(p, qpos ? qpos[idx] : idx, next);
        if (contrib_only) return;
        if (hit) { RDR_INLINE_CALL s.sample_from(next, p, idx); }
        else {
            Ray dead = make_ray(v3(0), v3(0));
            put_ray(s.q_nee, idx, dead, true);
            put_ray(s.q_bsdf, idx, dead, true);
        }
    }
};

// Predicates for the order-preserving compaction of the live-lane list.
struct KeepHit {       // lane survives iff its vertex recorded a hit (update_active_pixels)
    const int *shape;
    RDR_FN bool operator()(int p) const { return shape[p] >= 0; }
};
struct KeepNonZeroDir {   // init_active_pixels: drop rays with an all-zero direction
    const double *ray; int n;
    RDR_FN bool operator()(int p) const { return !all_zero(ld3(ray, n, p, 3)); }
};

} // namespace rdr
