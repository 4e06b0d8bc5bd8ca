// render.cpp -- host driver of the wavefront differentiable path tracer (rdr_render()).
//
// Counterpart of render() in src/pathtracer.cpp:177-958: per Sobol' sample it runs the forward
// wavefront, and -- when an image gradient is given -- re-walks the stored path backwards,
// adding the two edge-sampling estimators.  The structure of the loops (which lanes are alive,
// which Sobol' dimensions each draw consumes, in which order contributions reach the image) is
// the reference's, because sample-exact parity depends on it; how a bounce is cut into kernels
// and what is kept in HBM is ours (stages_fwd.h / stages_bwd.h / stages_edge.h).
#include "render.h"
#include "tuning.h"
#include <atomic>
#include <exception>
#include <cstdio>
#include <cstdlib>
#include "stages_fwd.h"
#include "stages_bwd.h"
#include "stages_edge.h"
#include <algorithm>
#include <memory>
#include <stdexcept>
#include <type_traits>
#include <vector>

namespace rdr {

namespace {

// Independent stages on side streams (DESIGN.md section 3 "Streams")?  Part of the call's tuning (RDR_TUNE_NO_OVERLAP): bench.py
// times the traversal kernel on its own in an extra, untimed pass.
inline bool overlap_on() { return !tuning().has(RDR_TUNE_NO_OVERLAP); }

// Device arena: typed arrays from the caching allocator (exec::pool_alloc); they go back to its free lists when the
// call ends, so the next render() of the same shape performs no hipMalloc / hipFree at all.
struct Arena {
    std::vector<void *> blocks;
    template <class T> T *get(size_t count) {
        T *p = (T *)exec::pool_alloc(sizeof(T) * (count ? count : 1));
        blocks.push_back(p);
        return p;
    }
    // Blocks go back to the pool for ANY stream to reuse: on the normal path render() has drained its streams by then; when the
    // call ends by an exception, kernels may still be running on them
    ~Arena() {
        if (std::uncaught_exceptions() > 0) exec::device_sync();
        for (void *p : blocks) exec::pool_free(p);
    }
};

// `with_rdiff` false: the lean stages neither store nor read ray differentials (stages_fwd.h: lean_slice nulls the pointer in
// every kernel) -- 96 of a slice's 185 bytes per lane, 30 % of everything a gradient render of a plain scene allocates
VSlice make_slice(Arena &a, int n, bool with_occl, bool with_rdiff = true) {
    VSlice v;
    v.n = n;
    v.ray = a.get<double>((size_t)6 * n);
    v.rdiff = with_rdiff ? a.get<double>((size_t)12 * n) : nullptr;
    v.shape = a.get<int>(n); v.tri = a.get<int>(n);
    v.thr = a.get<double>((size_t)3 * n);
    v.mrough = a.get<double>(n);
    v.occl = with_occl ? a.get<unsigned char>(n) : nullptr;
    v.erd = nullptr; v.erd_touched = nullptr; v.erd_seg = nullptr; v.erd_S = 0; v.erd_P0 = 0;
    return v;
}

struct Queues {
    rt::RayRec *nee, *bsdf;
    rt::HitRec *h_nee, *h_bsdf;
    int *pos[2];                   // fused bounces: queue slot of every entry of the current live-lane list (ping-pong)
};

// Bounces of one chain of paths (the camera paths of a sample batch, or the sub-paths of one edge pass).  Fused form (Sobol'
// sampler; RDR_TUNE_NO_FUSED_BOUNCE switches it off): the stage that evaluates bounce d also draws the rays of bounce d + 1
// (stages_fwd.h: BounceContribSample), so from the second bounce on the rays sit at the positions of the PREVIOUS live-lane
// list: `queue_n` entries, a lane's slot in `qpos`.
struct BounceChain {
    bool fused = false;
    bool have_rays = false;        // the rays of the bounce about to run were drawn by the previous bounce's stage
    const int *qpos = nullptr;
    exec::Count queue_n{0};
    int flip = 0;
};

struct ChannelLayout { int nd, radiance_dim; ChannelsD ch; };
ChannelLayout layout_of(const rdr_render_options &o, int max_generic) {
    ChannelLayout l;
    l.nd = 0; l.radiance_dim = -1;
    if (o.num_channels > kMaxChannels) throw std::runtime_error("render: too many channels");
    if (max_generic > kMaxGeneric) throw std::runtime_error("render: generic textures wider than 16 channels are not supported");
    l.ch.n = o.num_channels; l.ch.max_generic = max_generic; l.ch.radiance_off = -1;
    l.ch.id = nullptr;                 // device copy made by render()
    l.ch.radiance_only = o.num_channels == 1 && o.channels[0] == RDR_CH_RADIANCE;
    int d = 0;
    for (int i = 0; i < o.num_channels; ++i) {
        if (o.channels[i] == RDR_CH_RADIANCE) {
            if (l.radiance_dim != -1) throw std::runtime_error("Duplicated radiance channel");   // src/channels.cpp:24-26
            l.radiance_dim = i;             // [quirk] channel index, see ChannelsD
            l.ch.radiance_off = d;
        }
        int one = o.channels[i];
        int w = compute_num_channels(&one, 1, max_generic);
        if (w < 0) throw std::runtime_error("render: unknown channel id");
        d += w;
    }
    l.nd = d;
    l.ch.nd = d; l.ch.radiance_dim = l.radiance_dim;
    return l;
}

// Debugging aid (RDR_DEBUG_DUMP=<dir>): stage buffers are written to <dir>/<tag>.bin so a GPU run can be
// diffed against the CPU harness stage by stage.  Off unless the variable is set.
void debug_dump(const char *tag, int sample, int depth, const void *dev_ptr, size_t bytes) {
    static const char *dir = std::getenv("RDR_DEBUG_DUMP");
    if (!dir || !dev_ptr) return;
    std::vector<char> host(bytes);
    exec::download(host.data(), dev_ptr, bytes);
    char path[1024];
    std::snprintf(path, sizeof(path), "%s/%s_s%d_d%d.bin", dir, tag, sample, depth);
    if (FILE *f = std::fopen(path, "wb")) { std::fwrite(host.data(), 1, bytes, f); std::fclose(f); }
}

// PCG streams are stateful, so a shard that starts at sample k > 0 cannot reproduce the single-process stream
// (how far it has advanced depends on the earlier samples); shards draw from their own seeds instead.
uint64_t pcg_stream_seed(const rdr_render_options &o) {
    return (uint64_t)o.seed + (uint64_t)o.sample_offset * 0x9E3779B97F4A7C15ULL;
}

bool scene_is_lean(const Scene &scene, const ChannelsD &ch) {
    const CameraD &c = scene.d.cam;
    return scene.d.envmap == nullptr && c.kind == kCamPerspective && !c.distortion.defined && ch.radiance_only &&
           !scene.has_mipmaps && !scene.has_textures && !scene.has_vertex_colors;
}

// Which specialisation of the stages a scene can use (stages_fwd.h): kLean / kMid / kGeneral.
constexpr int kGeneral = 0, kLean = 1, kMid = 2;
int scene_kind(const Scene &scene, const ChannelsD &ch) {
    if (tuning().has(RDR_TUNE_FORCE_GENERAL)) return kGeneral;       // A/B: what the specialisations buy
    if (scene_is_lean(scene, ch)) return kLean;
    const CameraD &c = scene.d.cam;
    if (scene.d.envmap == nullptr && c.kind == kCamPerspective && !c.distortion.defined && ch.radiance_only) return kMid;
    return kGeneral;
}
// Launch `f`, or the specialisation of it that the scene allows (see LeanStage / MidStage in stages_fwd.h).
template <class F> void launch_v(int kind, exec::Count n, const F &f) {
    if (kind == kLean) exec::launch(n, LeanStage<F>{f});
    else if (kind == kMid) exec::launch(n, MidStage<F>{f});
    else exec::launch(n, f);
}

// Side streams of a large frame run at low priority (exec::side_stream: 2, 3), those of a small one at the default (0, 1).
inline int side_index(int k, int lanes) { return lanes >= (1 << 19) ? k + 2 : k; }

// One NEE + BSDF bounce over the live lanes of `v`; fills `vn` and the next live-lane list.  The lane count stays on the
// device (exec::Count); `dyn` / `dyn_inc`: the dimension counter that advances if this bounce had lanes to run.
// When to fuse (measured, bunny_box, `profiles/r4_notes.md`): a 256 x 256 x 4 spp optimisation-loop iteration -- chains of
// launches that each last as long as their longest lane -- 12.4 -> 11.5 ms; the 1024 x 1024 benchmark 62.7 -> 58.5 Msamples/s:
// the fused stage holds 188 registers (two waves per SIMD where BounceContrib runs three and BounceSample five) and the next
// bounce's queues keep the dead slots of the lanes that missed.  So: chains of up to 2^19 lanes; plain scenes only (the textured
// forms of the fused stage need 430-520 B of scratch per lane at their register cap); the stateless sampler only.
inline bool fuse_bounces(bool pcg, int kind, int chain_lanes) {
    return !pcg && kind == kLean && chain_lanes <= (1 << 19) && !tuning().has(RDR_TUNE_NO_FUSED_BOUNCE);
}
constexpr int kEmitterTestMax = 8;          // most emitter triangles tested per last-bounce ray in BounceSample (beyond: every ray is traced)
// `chain` / `vnn` (fused form): the chain's state, and the slice that receives the rays of the NEXT bounce (null: this is the
// chain's last bounce)
exec::Count run_bounce(const Scene &scene, const SceneD &sd, const SamplerD &rng, int dim, int rng_shift,
                       const int *active, exec::Count num_active, const VSlice &v, const VSlice &vn,
                       const Queues &q, const Sink &sink, int *next_active, int *dyn = nullptr, int dyn_inc = 0,
                       bool shadow_rays_coherent = false, BounceChain *chain = nullptr, const VSlice *vnn = nullptr, bool last_bounce = false,
                       bool next_is_last = false) {
    const int lean = scene_kind(scene, sink.ch);
    const bool fused = chain && chain->fused;
    const bool drawn = fused && chain->have_rays;                      // this bounce's rays exist already
    const exec::Count queue_n = drawn ? chain->queue_n : num_active;
    const int *qpos = drawn ? chain->qpos : nullptr;
    // the last bounce of a plain scene lit by a few triangles: continuation rays that meet no emitter triangle are not traced
    // (stages_fwd.h: BounceSample::last_bounce_emitters)
    const bool emitter_test = lean == kLean && scene.emitter_triangles > 0 && scene.emitter_triangles <= kEmitterTestMax &&
                              !tuning().has(RDR_TUNE_TRACE_EVERY_CONTINUATION);
    if (!drawn) {
        BounceSample bs{sd, rng, dim, rng_shift, active, v, vn, q.nee, q.bsdf};
        bs.last_bounce_emitters = last_bounce && emitter_test;
        launch_v(lean, num_active, bs);
    }
    // the shadow-ray and the continuation-ray queue are traced side by side: both kernels wait on dependent loads
    // with a fraction of their lanes active (profiles/r1_notes.md), so they fill each other's gaps
    const bool side = overlap_on();
    if (side) {
        static thread_local exec::Fence *fences[16][2] = {};                   // per host thread and device
        const int dev = exec::current_device();
        exec::Fence *&queued = fences[dev & 15][0], *&shadow_done = fences[dev & 15][1];
        if (!queued) { queued = new exec::Fence(); shadow_done = new exec::Fence(); }
        hipStream_t main_stream = exec::ctx().stream;
        queued->after(main_stream);
        {
            exec::StreamScope on(exec::side_stream(side_index(1, num_active.upper)));
            queued->gate(exec::ctx().stream);
            exec::trace(scene.bvh, q.nee, q.h_nee, queue_n, true, shadow_rays_coherent);
            shadow_done->after(exec::ctx().stream);
        }
        exec::trace(scene.bvh, q.bsdf, q.h_bsdf, queue_n, false);
        shadow_done->gate(main_stream);
    } else {
        exec::trace(scene.bvh, q.nee, q.h_nee, queue_n, true, shadow_rays_coherent);
        exec::trace(scene.bvh, q.bsdf, q.h_bsdf, queue_n, false);
    }
    const BounceContrib contrib{sd, rng, dim, rng_shift, active, v, vn, q.h_nee, q.h_bsdf, sink};
    if (!fused) {
        launch_v(lean, num_active, contrib);
        return exec::compact_dev(active, num_active, next_active, KeepHit{vn.shape}, nullptr, dyn, dyn_inc);
    }
    if (vnn) {
        // ... and the rays of the next bounce: the next list's lanes are drawn HERE, at this list's positions
        BounceSample next{sd, rng, dim + 7, rng_shift, active, vn, *vnn, q.nee, q.bsdf};
        next.last_bounce_emitters = next_is_last && emitter_test;
        launch_v(lean, num_active, BounceContribSample{contrib, next, qpos});
    } else if (qpos) {
        launch_v(lean, num_active, BounceContribSample{contrib, BounceSample{sd, rng, dim + 7, rng_shift, active, vn, vn, nullptr, nullptr}, qpos, 1});
    } else {
        launch_v(lean, num_active, contrib);
    }
    int *pos_out = q.pos[chain->flip];
    chain->flip ^= 1;
    const exec::Count next = exec::compact_dev(active, num_active, next_active, KeepHit{vn.shape}, nullptr, dyn, dyn_inc, pos_out);
    chain->have_rays = vnn != nullptr;
    chain->qpos = pos_out;
    chain->queue_n = num_active;
    return next;
}

// ---- gradient accumulators ------------------------------------------------------------------------
// fp64 mirrors of every tensor in the caller's DScene; folded into the fp32 tensors by flush().
struct GradStore {
    Arena arena;
    GScene g;
    std::vector<GShape> h_shapes;
    std::vector<GMaterial> h_materials;

    // The accumulators live in ONE allocation in two tiers (exec.h: ReplicaLayout), each a block of `stride` doubles that
    // is replicated `replicas` times (replica r at base + r * stride); rdr::accum() picks the replica from the wave id,
    // which spreads the atomics on hot addresses (camera, lights, constant albedos, wall corners) over many cache lines /
    // memory channels.  Tensors of at most kSmallTensor elements go to the small tier (256 replicas) while it has room;
    // the rest -- image textures, big meshes -- to the large tier, which gets the replicas that fit exec::replica_budget():
    // 256 MiB, up to 1 GiB for a job whose length pays for zeroing and summing that much.
    // (One tier for everything gave the camera of a scene with 60 MB of texture gradients 4 replicas: every stage that adds
    // to it or to the walls ran 2-3.5 x longer than with 256, profiles/r3_notes.md.)  flush() sums the replicas in fixed order.
    static constexpr size_t kSmallTensor = 16384, kSmallTierMax = 65536;      // doubles (one replica of the small tier: <= 512 KiB)
    struct Tier { double *base = nullptr; size_t stride = 0, cursor = 0; int replicas = 1; };
    Tier tier[2];                  // 0 = small tensors, 1 = large
    struct Pair { double *acc; float *out; size_t count; int tier; };
    std::vector<Pair> pairs;
    bool counting = true;

    double *place(size_t count, int &t) {
        const size_t padded = (count + 3) & ~(size_t)3;
        t = (count <= kSmallTensor && tier[0].cursor + padded <= kSmallTierMax) ? 0 : 1;
        const size_t at = tier[t].cursor;
        tier[t].cursor += padded;
        return counting ? reinterpret_cast<double *>(8) : tier[t].base + at;       // placeholder in pass 1
    }
    double *mirror(float *out, size_t count) {
        if (!out || count == 0) return nullptr;
        int t;
        double *acc = place(count, t);
        if (!counting) pairs.push_back(Pair{acc, out, count, t});
        return acc;
    }
    GTex mirror_tex(const TexD &t, const rdr_dtexture_desc &d) {
        GTex g;
        for (int i = 0; i < kMaxMip; ++i) g.texels[i] = nullptr;
        g.uv_scale = nullptr;
        if (t.num_levels == 0 || d.num_levels == 0) return g;
        bool constant = t.width[0] <= 0 && t.height[0] <= 0;
        for (int i = 0; i < t.num_levels && i < d.num_levels; ++i) {
            size_t count = constant ? (size_t)t.channels : (size_t)t.width[i] * t.height[i] * t.channels;
            g.texels[i] = mirror(d.texels[i], count);
        }
        g.uv_scale = mirror(d.uv_scale, 2);
        return g;
    }

    GradStore(const Scene &scene, const rdr_dscene_desc &ds, size_t job_samples) {
        counting = true;
        layout(scene, ds);                                   // pass 1: size of one replica of each tier
        for (Tier &t : tier) t.stride = (t.cursor + 31) & ~(size_t)31;
        tier[0].replicas = tier[0].stride ? exec::choose_replicas(tier[0].stride * sizeof(double), 256 * kSmallTierMax * sizeof(double)) : 1;
        tier[1].replicas = tier[1].stride ? exec::choose_replicas(tier[1].stride * sizeof(double), exec::replica_budget(job_samples)) : 1;
        const size_t small_total = tier[0].stride * tier[0].replicas, total = small_total + tier[1].stride * tier[1].replicas;
        double *block = arena.get<double>(total);
        exec::zero(block, sizeof(double) * total);
        tier[0].base = block; tier[1].base = block + small_total;
        exec::set_replicas(ReplicaLayout{block + tier[0].stride, tier[0].stride, tier[1].stride,
                                         (unsigned)(tier[0].replicas - 1), (unsigned)(tier[1].replicas - 1)});
        counting = false; pairs.clear();
        for (Tier &t : tier) t.cursor = 0;
        layout(scene, ds);                                   // pass 2: real pointers
        // accum_triple / accum_block pick the replica from the FIRST address of a group (exec.h: replica_of) and add the
        // others at the same offset: a tensor must lie in one tier as a whole (place() puts it there; checked, not assumed)
        for (const Pair &p : pairs) {
            const Tier &t = tier[p.tier];
            if (p.acc < t.base || p.acc + p.count > t.base + t.stride)
                throw std::runtime_error("render: gradient accumulator straddles a replica tier (GradStore::layout)");
        }
        accumulators_laid_out(tier[0].base, tier[0].stride);           // (a hook of the accumulator backend: nothing on the device)
    }
    void layout(const Scene &scene, const rdr_dscene_desc &ds) {
        if (ds.num_shapes != (int)scene.shapes.size() || ds.num_materials != (int)scene.materials.size() ||
            ds.num_area_lights != (int)scene.lights.size())
            throw std::runtime_error("render: DScene does not match the Scene (shape/material/light counts)");
        h_shapes.resize(scene.shapes.size());
        for (size_t i = 0; i < scene.shapes.size(); ++i) {
            const ShapeD &sh = scene.shapes[i];
            const rdr_dshape_desc &d = ds.shapes[i];
            h_shapes[i].vertices = mirror(d.vertices, (size_t)3 * sh.num_vertices);
            if (!h_shapes[i].vertices) throw std::runtime_error("render: DShape.vertices is required");
            h_shapes[i].uvs = sh.uvs ? mirror(d.uvs, (size_t)2 * (sh.num_uv_vertices > 0 ? sh.num_uv_vertices : sh.num_vertices)) : nullptr;
            h_shapes[i].normals = sh.normals ? mirror(d.normals, (size_t)3 * (sh.num_normal_vertices > 0 ? sh.num_normal_vertices : sh.num_vertices)) : nullptr;
            h_shapes[i].colors = sh.colors ? mirror(d.colors, (size_t)3 * sh.num_vertices) : nullptr;
        }
        h_materials.resize(scene.materials.size());
        for (size_t i = 0; i < scene.materials.size(); ++i) {
            const MaterialD &m = scene.materials[i];
            const rdr_dmaterial_desc &d = ds.materials[i];
            h_materials[i].diffuse = mirror_tex(m.diffuse, d.diffuse_reflectance);
            h_materials[i].specular = mirror_tex(m.specular, d.specular_reflectance);
            h_materials[i].roughness = mirror_tex(m.roughness, d.roughness);
            h_materials[i].generic = mirror_tex(m.generic, d.generic_texture);
            h_materials[i].normal_map = mirror_tex(m.normal_map, d.normal_map);
        }
        if (!counting) {
            g.shapes = arena.get<GShape>(h_shapes.size());
            exec::upload(g.shapes, h_shapes.data(), sizeof(GShape) * h_shapes.size());
            g.materials = arena.get<GMaterial>(h_materials.size());
            exec::upload(g.materials, h_materials.data(), sizeof(GMaterial) * h_materials.size());
        }
        // light intensities: one contiguous fp64 block, scattered back per light
        g.light_intensity = nullptr;
        if (!scene.lights.empty()) {
            int light_tier = 0;
            g.light_intensity = place(3 * scene.lights.size(), light_tier);
            if (!counting)
                for (size_t l = 0; l < scene.lights.size(); ++l)
                    if (ds.area_lights[l].intensity) pairs.push_back(Pair{g.light_intensity + 3 * l, ds.area_lights[l].intensity, 3, light_tier});
        }
        const rdr_dcamera_desc &dc = ds.camera;
        g.cam.position = mirror(dc.position, 3); g.cam.look = mirror(dc.look, 3); g.cam.up = mirror(dc.up, 3);
        g.cam.cam_to_world = mirror(dc.cam_to_world, 16); g.cam.world_to_cam = mirror(dc.world_to_cam, 16);
        g.cam.intrinsic_mat_inv = mirror(dc.intrinsic_mat_inv, 9); g.cam.intrinsic_mat = mirror(dc.intrinsic_mat, 9);
        g.cam.distortion = mirror(dc.distortion_params, 8);
        g.envmap = nullptr;
        if (scene.d.envmap && ds.envmap) {
            GEnvmap h_envmap;
            h_envmap.values = mirror_tex(scene.h_envmap.values, ds.envmap->values);
            h_envmap.world_to_env = mirror(ds.envmap->world_to_env, 16);
            if (!counting) {
                g.envmap = arena.get<GEnvmap>(1);
                exec::upload(g.envmap, &h_envmap, sizeof(GEnvmap));
            }
        }
    }
    void flush() {
        for (const Pair &p : pairs)
            if (p.tier == 0) accumulator_before_fold(tier[0].base, p.acc, p.count);
        accumulators_folded();
        // one launch per tier for all its tensors, unless two mirrors feed overlapping output ranges (a tensor shared by two
        // DScene entries): those must add one after the other
        std::vector<Pair> by_out(pairs);
        std::sort(by_out.begin(), by_out.end(), [](const Pair &a, const Pair &b) { return a.out < b.out; });
        bool aliased = false;
        for (size_t i = 1; i < by_out.size(); ++i) aliased = aliased || by_out[i - 1].out + by_out[i - 1].count > by_out[i].out;
        for (int t = 0; t < 2; ++t) {
            const Tier &tr = tier[t];
            std::vector<FlushSegment> seg;
            for (const Pair &p : pairs) if (p.tier == t) seg.push_back(FlushSegment{(size_t)(p.acc - tr.base), p.count, p.out});
            if (seg.empty()) continue;
            std::sort(seg.begin(), seg.end(), [](const FlushSegment &a, const FlushSegment &b) { return a.begin < b.begin; });
            if (tr.stride > (size_t)0x7fffffff) throw std::runtime_error("render: gradient block too large");
            FlushSegment *d_seg = arena.get<FlushSegment>(seg.size());
            exec::upload(d_seg, seg.data(), sizeof(FlushSegment) * seg.size());
            if (!aliased) exec::launch((int)tr.stride, FlushGrad{tr.base, tr.stride, tr.replicas, d_seg, (int)seg.size()});
            else for (size_t i = 0; i < seg.size(); ++i)
                exec::launch((int)(seg[i].begin + seg[i].count), FlushGrad{tr.base, tr.stride, tr.replicas, d_seg + i, 1});
        }
        exec::set_replicas(ReplicaLayout{nullptr, 0, 0, 0, 0});
    }
};

// ---- backward sweep of one sample (src/pathtracer.cpp:392-944) ------------------------------------
// Sample batches (small frames, gradient renders of plain scenes with the Sobol' sampler): `S` consecutive samples are
// rendered as ONE set of lanes -- lane v = pixel v % P0 of the batch's sample v / P0 -- so that a launch covers S x P0 lanes
// instead of P0.  At 256 x 256 a sample is a chain of ~280 dependent launches that each last as long as their longest
// lane (one wave per SIMD); four samples per launch take barely longer.  Nothing in the stage bodies changes: per-lane state
// is indexed by the lane id as before; the samplers map a slot to (sample, slot of that sample) (sobol.h), the camera maps a
// lane to its pixel (CameraD::batch_rows), the upstream image gradient is replicated per sample, and what the reference
// decides per sample -- compacted ranks of the secondary-edge sampler's slots, dimension counters that advance only for
// samples whose lists are not empty -- is kept per sample (`seg` tables, edge_dyn[s]).
struct BatchView { int S = 1, P0 = 0, rows = 0; bool on = false; };

struct Backward {
    const Scene &scene; const rdr_render_options &opt;
    int P, B; const float *d_image; float *screen_grad; double weight; int nd, radiance_dim;
    BatchView batch;               // P = batch.S * batch.P0 lanes when batch.on
    SceneD sd;                     // scene.d (+ the batch's camera mapping)
    GradStore &grads;              // shared by the sample workers: every add is an atomic
    Arena arena;
    AdjState adj;
    int adj_point_doubles = kAdjPointDoubles;

    ChannelsD ch;
    int lean = kGeneral;               // which stage specialisation the scene qualifies for (kLean / kMid / kGeneral)
    uint64_t *pcg_edge = nullptr;      // PCG edge sampler: one state per slot (src/pathtracer.cpp:221-222)
    int *edge_dyn = nullptr;           // device-side part of the edge sampler's dimension counter (see run_sample)
    int *hoist_dyn = nullptr;          // the same counter as the hoisted first-vertex picks will find it (they run beside the sweep)
    double *multipliers = nullptr;     // [2P x nd], primary-edge channel weights (non-radiance channels only)

    Backward(const Scene &scene_, const rdr_render_options &opt_, GradStore &grads_, int P_, int B_,
             const float *d_image_, float *screen_grad_, double weight_, int nd_, int radiance_dim_, const ChannelsD &ch_,
             const BatchView &batch_)
        : scene(scene_), opt(opt_), P(P_), B(B_), d_image(d_image_), screen_grad(screen_grad_), weight(weight_),
          nd(nd_), radiance_dim(radiance_dim_), batch(batch_), sd(scene_.d), grads(grads_), ch(ch_) {
        if (batch.on) sd.cam.batch_rows = batch.rows;
        lean = scene_kind(scene, ch);
        adj.n = P; adj.plain = 0;
        adj.thr = arena.get<double>((size_t)3 * P);
        adj.ray_dir = arena.get<double>((size_t)3 * P);
        // (the lean stages keep position and shading normal: 6 of the 24 components, stages_bwd.h: AdjState)
        adj_point_doubles = lean == kLean ? kAdjPointDoublesLean : kAdjPointDoubles;
        adj.point = arena.get<double>((size_t)adj_point_doubles * P);
        adj.carries = scene.d.envmap == nullptr ? arena.get<unsigned char>((size_t)P) : nullptr;      // (stages_bwd.h: AdjState)
        nee_act = arena.get<int>((size_t)P);
        const bool edges_on = scene.edges && scene.edges->d.num_edges > 0 &&
                              (scene.use_primary_edges || scene.use_secondary_edges);
        if (edges_on) {
            edge_dyn = arena.get<int>(kMaxBatch);
            hoist_dyn = arena.get<int>(kMaxBatch);
            const int L = 2 * P;                      // edge lanes: two rays per sample slot
            ea = make_slice(arena, L, false, lean != kLean);
            eb = make_slice(arena, L, false, lean != kLean);
            if (scene.has_mipmaps) {
                // the reference's buffer is a fresh allocation per render call; fresh pages read as zero
                ea.erd = eb.erd = arena.get<double>((size_t)12 * L);
                exec::zero(ea.erd, sizeof(double) * 12 * L);
                if (batch.on) {
                    // chain mode (see run_sample, "primary edges"): one copy of that buffer per sample of the batch, which
                    // entries the batch has written, the state that travels from sample to sample, and the lanes that need it
                    chain_n = 2 * batch.P0;
                    ea.erd_touched = eb.erd_touched = arena.get<unsigned char>(L);
                    erd_chain = arena.get<double>((size_t)12 * chain_n);
                    exec::zero(erd_chain, sizeof(double) * 12 * chain_n);
                    deferred = arena.get<int>(L);
                    deferred_seg = arena.get<int>(kMaxBatch + 1);
                    deferred_count = arena.get<int>(kMaxBatch);
                    prim_dyn = arena.get<int>(kMaxBatch);
                }
            }
            for (int k = 0; k < 3; ++k) elist[k] = arena.get<int>(L);
            nee_slots = arena.get<int>(P);
            gather_cands = arena.get<GatherCand>((size_t)kGatherCands * P);
            gshared.book = arena.get<GatherBook>(1);
            // The lists of the next-event-mode pick's heavy slots (more than kGatherCands candidates or kGatherBudget pops: ~0.3 % of
            // the slots of the Cornell-box benchmark) grow with the launch set: 8 192 heavy slots / 131 072 work items were sized for
            // one-sample launches of a 1024 x 1024 frame -- an 8-sample set has ~11 k heavy slots, and every slot past the cap fell
            // back to the reference-order walk (a few thousand sequential walks: 4.5 ms per launch set, profiles/r6_notes.md 8b).
            gshared.heavy_cap = (int)std::min<long long>(std::max<long long>((long long)P / 256, kGatherHeavyCap), kGatherHeavyMax);
            gshared.work_cap = (int)std::min<long long>(std::max<long long>((long long)P / 32, kGatherWorkCap), kGatherWorkMax);
            gshared.heavy_slot = arena.get<int>(gshared.heavy_cap);
            gshared.work = arena.get<GatherWork>(gshared.work_cap);
            gshared.cands_big = arena.get<GatherCand>((size_t)kGatherCandsBig * gshared.heavy_cap);
            // tests: small lists, so that the overflow paths run (rdr_tuning::gather_*_cap_plus1)
            if (tuning().gather_heavy_cap >= 0) gshared.heavy_cap = std::min(tuning().gather_heavy_cap, gshared.heavy_cap);
            if (tuning().gather_work_cap >= 0) gshared.work_cap = std::min(tuning().gather_work_cap, gshared.work_cap);
            h_leaves = arena.get<HLeaf>((size_t)kHSamples * P);
            h_spill = arena.get<HLeaf>((size_t)(kHSamples - kHStackLds) * P);
            h_descent = arena.get<HDescent>((size_t)P);
            edge_contrib = arena.get<double>(L);
            edge_tmin = arena.get<double>(L);
            hit_pos = arena.get<double>((size_t)3 * L);
            // only written on hits; lanes that reach the environment light read what an earlier hit left there,
            // like the reference's never-cleared edge_surface_points (src/pathtracer.cpp:594-600); fresh pages are 0
            exec::zero(hit_pos, sizeof(double) * 3 * L);
            if (batch.on && scene.d.envmap != nullptr) {          // see HitPosView (stages_edge.h)
                hp_written = arena.get<unsigned char>(L);
                hp_violations = arena.get<int>(1);
                exec::zero(hp_violations, sizeof(int));
                // stale reads across the samples of a batch are recorded and replayed after the sweep (HitPosView)
                hp_event_cap = L;
                hp_events = arena.get<HitEvent>((size_t)hp_event_cap);
                hp_event_count = arena.get<int>(1);
                exec::zero(hp_event_count, sizeof(int));
                hp_carry = arena.get<double>((size_t)3 * 2 * batch.P0);
                exec::zero(hp_carry, sizeof(double) * 3 * 2 * batch.P0);       // (the reference's fresh pages)
                replay_live = arena.get<unsigned char>(P);
            }
            prim_recs = arena.get<PrimaryEdgeRec>(P);
            sec_recs = arena.get<SecondaryEdgeRec>(P);
            sec_picks = arena.get<SecPick>(P);
            sec_mode = arena.get<unsigned char>(P);
            if (!ch.radiance_only) multipliers = arena.get<double>((size_t)L * nd);
            if (opt.sampler_type == RDR_SAMPLER_INDEPENDENT) {
                pcg_edge = arena.get<uint64_t>(P);
                exec::launch(P, PcgInit{pcg_edge, pcg_stream_seed(opt) + 131071U});
            }
        }
    }

    VSlice ea, eb;                 // ping-pong vertex slices of the edge sub-paths (2P lanes each)
    // chain mode: sample batches of a scene with mip levels (run_sample, "primary edges")
    int chain_n = 0;                   // entries of the reference's differential buffer: 2 x pixels
    double *erd_chain = nullptr;       // [12 x chain_n] what the last finished sample left in it
    int *deferred = nullptr, *deferred_seg = nullptr, *deferred_count = nullptr, *prim_dyn = nullptr;
    bool chain_mode() const { return erd_chain != nullptr; }
    // the secondary-edge pass of depth d numbers its lanes by rank in the batch's list: map them to the samples' own entries
    void erd_view(const int *seg) { ea.erd_seg = eb.erd_seg = seg; ea.erd_S = eb.erd_S = cur_S; ea.erd_P0 = eb.erd_P0 = batch.P0; }
    int *elist[3] = {nullptr, nullptr, nullptr};
    HLeaf *h_leaves = nullptr, *h_spill = nullptr;   // hierarchical pick: recorded leaves / spilled stack entries per list position
    int *nee_act = nullptr;                          // lanes of a depth whose next-event estimate has something to differentiate
    HDescent *h_descent = nullptr;                   // ... and what the descent hands to the leaf launch (stages_edge.h)
    GatherShared gshared{nullptr, nullptr, nullptr, nullptr, 0, 0};     // heavy slots of the gather: big candidate lists, subtree work items
    GatherCand *gather_cands = nullptr;        // positive leaves found by the NEE-mode gather, kGatherCands per list position
    int *nee_slots = nullptr;                  // slots of the NEE-mode edge pick (its walk runs beside the hierarchical pick)
    exec::Fence depth_begin, adjoint_done, setup_done, walk_done, picks_begin, pickh_done;
    const bool overlap = overlap_on();
    double *edge_contrib = nullptr, *edge_tmin = nullptr, *hit_pos = nullptr;
    unsigned char *hp_written = nullptr; int *hp_violations = nullptr;
    HitEvent *hp_events = nullptr; int *hp_event_count = nullptr; int hp_event_cap = 0;
    double *hp_carry = nullptr;        // [3 x 2 P0] what the batch found in the scratch (state after the previous batch)
    unsigned char *replay_live = nullptr;
    HitPosView hit_view(const int *seg, int depth = 0) const {
        return HitPosView{hit_pos, ea.n, hp_written ? seg : nullptr, cur_S, batch.P0, hp_written, hp_violations,
                          hp_events, hp_event_count, hp_event_cap, depth};
    }
    PrimaryEdgeRec *prim_recs = nullptr;
    SecondaryEdgeRec *sec_recs = nullptr;
    SecPick *sec_picks = nullptr;
    unsigned char *sec_mode = nullptr;

    // PCG edge sampler: the states of slots [0, n) move on by `count` numbers (one next_*_samples call group); `gate`: only
    // if that device-side lane count is positive (a bounce without lanes draws nothing)
    void edge_rng_consumed(exec::Count n, int count, const int *gate = nullptr) {
        if (pcg_edge) exec::launch(n, PcgAdvance{pcg_edge, count, nullptr, gate});
    }
    void edge_rng_consumed_n(exec::Count n, int count) { edge_rng_consumed(n, count, nullptr); }
    // The edge sampler at dimension `edim` + what the device-side counter holds (see SamplerD::dyn)
    // `seg`: the slots are compacted ranks of the batch's samples (secondary-edge passes); otherwise, in a batch, every sample
    // owns P0 consecutive slots (primary-edge pass)
    SamplerD edge_rng_at(const SamplerD &rng_edge, int edim, const int *dyn = nullptr, const int *seg = nullptr) const {
        SamplerD r = rng_edge; r.pcg_base = edim; r.dyn = dyn ? dyn : edge_dyn;
        if (batch.on) { r.batch = cur_S; r.seg = seg; r.batch_lanes = seg ? 0 : batch.P0; }
        return r;
    }
    // per path depth of the current batch: where each sample's part of the live-lane list starts (device, kMaxBatch + 1 ints)
    const int *seg_tables = nullptr;
    int cur_S = 1;                 // samples in the batch that is being rendered
    const int *seg_of(int d) const { return batch.on ? seg_tables + (size_t)d * (kMaxBatch + 1) : nullptr; }
    // A secondary-edge pass at a depth that had lanes has drawn its four numbers per slot (`n`: that depth's live-lane count;
    // in a batch: per sample whose part of the list is not empty)
    void secondary_pass_consumed(exec::Count n, int d) {
        if (batch.on) exec::launch(cur_S, BumpDynSeg{edge_dyn, seg_of(d), 4});
        else exec::launch(1, BumpDyn{edge_dyn, n.dev, 4});
        edge_rng_consumed_n(n, 4);
    }

    template <int LEAN> void launch_pick_n(int need, exec::Count nN, const SecEdgeArgs &sa) {
        // order-free gather over the billboard hierarchy (SecEdgeGatherN), then the reference-order walk for the slots it
        // marked kPickOverflow (none once the big lists hold every heavy slot); RDR_PICKN_WALK=1 walks every slot (A/B measurements)
        const bool walk_all = tuning().has(RDR_TUNE_PICKN_WALK);
        const bool gather = !walk_all && sa.es.gather.num_nodes > 0;
        if (gather) {
            const int gneed = sa.es.gather.stack_need;
            exec::zero(gshared.book, sizeof(GatherBook));
            auto passes = [&](auto tag) {
                constexpr int NS = decltype(tag)::value;
                const int budget = tuning().gather_budget;
                launch_v(LEAN, nN, SecEdgeGatherN<NS>{sa, nee_slots, sec_picks, gather_cands, gshared, budget});
                launch_v(LEAN, gshared.work_cap, SecEdgeGatherSub<NS>{sa, nee_slots, gshared});
                launch_v(LEAN, gshared.heavy_cap, SecEdgeGatherReplay{sa, nee_slots, sec_picks, gshared});
            };
            if (gneed <= 24) passes(std::integral_constant<int, 24>{});
            else if (gneed <= 40) passes(std::integral_constant<int, 40>{});
            else passes(std::integral_constant<int, 64>{});
        }
        const int only_overflow = gather ? 1 : 0;
        const int *walk_needed = gather ? &gshared.book->walk_needed : nullptr;
        auto go = [&](auto walk) {
            walk.walk_needed = walk_needed;
            if (LEAN == kLean) exec::launch_persistent(nN, LeanWalk<decltype(walk)>{walk});
            else if (LEAN == kMid) exec::launch_persistent(nN, MidWalk<decltype(walk)>{walk});
            else exec::launch_persistent(nN, walk);
        };
        if (need <= 24) go(SecEdgePickNWalk<24>{sa, nee_slots, sec_picks, only_overflow});
        else if (need <= 32) go(SecEdgePickNWalk<32>{sa, nee_slots, sec_picks, only_overflow});
        else if (need <= 48) go(SecEdgePickNWalk<48>{sa, nee_slots, sec_picks, only_overflow});
        else go(SecEdgePickNWalk<64>{sa, nee_slots, sec_picks, only_overflow});
    }

    // Path-trace the live edge lanes to the end, starting with `n_act` lanes listed in elist[1] whose current vertex is in
    // `ea`.  The reference runs a bounce while lanes are left and advances the edge sampler by 7 dimensions per bounce it ran
    // (src/pathtracer.cpp:590-706); here every bounce up to max_bounces is queued -- one without lanes does nothing -- and the
    // device-side counter advances in the compaction that closes a bounce which had lanes (`edim` is the host-known part).
    // `seg`: the edge lanes are 2 x compacted rank (+ side) of the samples' live-lane lists (secondary-edge pass of that depth);
    // null: 2 x slot (+ side) with P0 slots per sample (primary-edge pass).
    // `dyn`: the per-sample dimension counters to read and advance (default: edge_dyn; the sequential part of a chain-mode
    // primary pass runs on its own copy)
    void trace_edge_paths(const SamplerD &rng_edge, int edim, exec::Count n_act, exec::Count n_slots, int first_depth, const Queues &q,
                          const Sink &sink, bool need_lights, const int *seg = nullptr, int *dyn = nullptr) {
        if (!dyn) dyn = edge_dyn;
        const bool has_lights = sd.num_lights > 0;
        if (need_lights && !has_lights) return;
        int cur = 1;
        BounceChain chain;
        chain.fused = fuse_bounces(pcg_edge != nullptr, lean, n_act.upper);
        for (int depth = first_depth, k = 0; depth < B && n_act.upper > 0; ++depth, ++k) {
            const VSlice &m = (k % 2 == 0) ? ea : eb;
            const VSlice &nx = (k % 2 == 0) ? eb : ea;
            int nxt = (cur == 1) ? 2 : 1;
            exec::Count next = run_bounce(scene, sd, edge_rng_at(rng_edge, edim, dyn, seg), edim, 1, elist[cur], n_act, m, nx, q, sink, elist[nxt],
                                          batch.on ? nullptr : dyn, 7, false, &chain, depth + 1 < B ? &m : nullptr, depth == B - 1, depth + 2 == B);
            // a batch: the counter of every sample that had lanes in this bounce (the list is ascending in the lane id)
            if (batch.on) exec::launch(cur_S, BumpDynList{dyn, elist[cur], n_act.dev, n_act.upper, seg, 2 * batch.P0, 7});
            edge_rng_consumed(n_slots, 7, n_act.dev);
            n_act = next;
            cur = nxt;
        }
    }

    // One sample -- or one batch of `S_now` samples on `lanes` = S_now x P0 lanes (`stride`: lanes the buffers were made for;
    // `seg`: the batch's per-depth segment tables)
    void run_sample(int sample_id, const SamplerD &main_rng, std::vector<VSlice> &vs, int *active, std::vector<exec::Count> &num_active, const Queues &q,
                    int n_lanes, int stride, int S_now, const int *seg) {
        const int P = n_lanes;                        // (shadows the member: the buffers' capacity)
        cur_S = S_now; seg_tables = seg;
        SamplerD rng = main_rng;
        SamplerD rng_edge{scene.sobol_table, opt.seed + 131071U, sample_id, pcg_edge, 0};   // src/pathtracer.cpp:221-227
        const bool has_lights = sd.num_lights > 0;
        const bool edges_on = prim_recs != nullptr;
        Sink esink{nullptr, edge_contrib, nd, radiance_dim, weight, ch, nullptr};
        Sink psink{nullptr, edge_contrib, nd, radiance_dim, weight, ch, multipliers};
        // Edge-sampler dimension = edim (what the host can count: 2 for the primary pass) + *edge_dyn (what depends on live-lane
        // counts, which stay on the device: 4 per secondary pass of a depth THAT HAD LANES -- the reference skips a dead depth
        // before its sampler draws, src/pathtracer.cpp:432-436 -- and 7 per bounce of an edge sub-path that had lanes to run,
        // see trace_edge_paths)
        int edim = 0;
        if (edge_dyn) exec::zero(edge_dyn, sizeof(int) * kMaxBatch);
        if (chain_mode()) exec::zero(ea.erd_touched, (size_t)ea.n);          // nothing of this batch's samples is in their copies yet
        if (hp_written) exec::zero(hp_written, (size_t)ea.n);
        // (the whole buffers: they are component-major with the buffers' lane count as the stride, so the first 3 x n_lanes
        //  doubles of a batch that is smaller than the buffers -- the last one of a call -- are not its lanes' records)
        exec::zero(adj.thr, sizeof(double) * 3 * stride);
        exec::zero(adj.ray_dir, sizeof(double) * 3 * stride);
        exec::zero(adj.point, sizeof(double) * adj_point_doubles * stride);
        if (adj.carries) exec::zero(adj.carries, (size_t)stride);
        const int dim0 = opt.sample_pixel_center ? 0 : 2;
        const bool pickh_fused = tuning().has(RDR_TUNE_PICKH_FUSED);     // A/B: the one-loop form
        const bool pickh_lazy = tuning().has(RDR_TUNE_PICKH_LAZY);       // A/B: per-field node loads
        // Small frames are chains of short launches (a 256 x 256 x 4 spp iteration: ~330 of them in 11 ms): the two extra launches
        // of the split pick and the three of every list compaction cost more there than fuller waves give back (+0.3-0.4 ms,
        // profiles/r6_notes.md).  The large-frame forms from 2^19 lanes per launch set on, or everywhere on request.
        const bool large_forms = tuning().has(RDR_TUNE_LARGE_FORMS) || P >= (1 << 19);
        const bool pickh_one_launch = tuning().has(RDR_TUNE_PICKH_ONE_LAUNCH) || !large_forms;       // one slot per lane from root to last leaf
        const int pickh_k = tuning().pickh_k, pickh_idle = tuning().pickh_idle, pickh_steps = tuning().pickh_steps;      // rdr_tuning::pickh_*
        // The two edge picks of a secondary pass: slot setup, the per-mode slot lists, the NEE-mode gather and the hierarchical
        // pick.  `early`: everything off the calling stream (setup + lists + gather on side stream 1, hierarchical pick on side
        // stream 0), so that the caller's stream is free for the bounce adjoints; otherwise setup, lists and the hierarchical pick
        // run on the calling stream and the gather beside them.  Joined by join_picks().
        struct PickPhase { bool running = false, early = false, side = false; };
        PickPhase picks_phase;
        auto start_picks = [&](int d, int edim_d, const int *dyn, bool early, bool side) -> SecEdgeArgs {
            const exec::Count nA = num_active[d];
            const int *act = active + (size_t)d * stride;
            const EdgeSceneD &es = scene.edges->d;
            SecEdgeArgs sa{sd, es, rng, dim0 + 7 * d, edge_rng_at(rng_edge, edim_d, dyn, seg_of(d)), edim_d, act, vs[d]};
            hipStream_t main_stream = exec::ctx().stream;
            const int need = es.max_stack;
            exec::Count nH(0), nN(0);
            auto lists = [&] {
                launch_v(lean, nA, SecEdgeSetup{sa, sec_mode, sec_recs, sec_picks, ea, edge_tmin, (pickh_fused || pickh_one_launch) ? nullptr : h_descent});
                nH = exec::compact_dev((const int *)nullptr, nA, elist[0], KeepMode{sec_mode, 1});
                const exec::Count nN2 = exec::compact_dev((const int *)nullptr, nA, nee_slots, KeepMode{sec_mode, 2});
                nN = exec::compact_dev((const int *)nullptr, nA, nee_slots, KeepMode{sec_mode, 3}, &nN2);   // dense-shape slots after the others
            };
            auto gather = [&] {
                if (lean == kLean) launch_pick_n<kLean>(need, nN, sa);
                else if (lean == kMid) launch_pick_n<kMid>(need, nN, sa);
                else launch_pick_n<kGeneral>(need, nN, sa);
            };
            auto hierarchical = [&] {
                if (pickh_fused) launch_v(lean, nH, SecEdgePickH{sa, elist[0], sec_picks});
                else if (pickh_one_launch && pickh_lazy) launch_v(lean, nH, SecEdgePickH2<false>{sa, elist[0], sec_picks, h_leaves, h_spill, nH.upper});
                else if (pickh_one_launch) launch_v(lean, nH, SecEdgePickH2<true>{sa, elist[0], sec_picks, h_leaves, h_spill, nH.upper});
                else {
                    // the descent as a walk with wave-local lane refill, then the recorded leaves (stages_edge.h, round 6)
                    auto descend = [&](auto walk) {
                        if (lean == kLean) exec::launch_chunked(nH, LeanWalk<decltype(walk)>{walk}, pickh_k, pickh_idle, pickh_steps);
                        else if (lean == kMid) exec::launch_chunked(nH, MidWalk<decltype(walk)>{walk}, pickh_k, pickh_idle, pickh_steps);
                        else exec::launch_chunked(nH, walk, pickh_k, pickh_idle, pickh_steps);
                    };
                    if (pickh_lazy) descend(SecEdgePickHDescend<false>{sa.es, elist[0], h_leaves, h_spill, h_descent, nH.upper});
                    else descend(SecEdgePickHDescend<true>{sa.es, elist[0], h_leaves, h_spill, h_descent, nH.upper});
                    static const bool leaves_walk = std::getenv("RDR_PICKH_LEAVES_WALK") != nullptr;       // (experiments)
                    if (leaves_walk) {
                        const SecEdgePickHLeavesWalk lw{sa.sc, sa.es, elist[0], sec_picks, h_leaves, h_descent, nH.upper};
                        if (lean == kLean) exec::launch_chunked(nH, LeanWalk<SecEdgePickHLeavesWalk>{lw}, pickh_k, pickh_idle, 2);
                        else if (lean == kMid) exec::launch_chunked(nH, MidWalk<SecEdgePickHLeavesWalk>{lw}, pickh_k, pickh_idle, 2);
                        else exec::launch_chunked(nH, lw, pickh_k, pickh_idle, 2);
                    } else
                    launch_v(lean, nH, SecEdgePickHLeaves{sa, elist[0], sec_picks, h_leaves, h_descent, nH.upper});
                }
            };
            if (early) {
                picks_begin.after(main_stream);
                {
                    exec::StreamScope on(exec::side_stream(side_index(1, P)));
                    picks_begin.gate(exec::ctx().stream);
                    lists();
                    setup_done.after(exec::ctx().stream);
                    gather();
                    walk_done.after(exec::ctx().stream);
                }
                {
                    exec::StreamScope on(exec::side_stream(side_index(0, P)));
                    setup_done.gate(exec::ctx().stream);
                    hierarchical();
                    pickh_done.after(exec::ctx().stream);
                }
            } else {
                lists();
                {
                    if (side) setup_done.after(main_stream);
                    exec::StreamScope on(side ? exec::side_stream(side_index(1, P)) : main_stream);
                    if (side) setup_done.gate(exec::ctx().stream);
                    gather();
                    if (side) walk_done.after(exec::ctx().stream);
                }
                hierarchical();
            }
            picks_phase = PickPhase{true, early, side};
            return sa;
        };
        auto join_picks = [&] {
            hipStream_t main_stream = exec::ctx().stream;
            if (picks_phase.early) { walk_done.gate(main_stream); pickh_done.gate(main_stream); }
            else if (picks_phase.side) walk_done.gate(main_stream);
            picks_phase.running = false;
        };
        // In a purely diffuse scene (and with the stateless sampler) only the first vertex samples secondary edges, and its
        // picks need nothing from the adjoint sweep: they start now, on the side streams, and run beside the bounce adjoints of
        // ALL depths instead of the first vertex's alone (its static sampler dimension is 4 per deeper depth that runs).
        const bool secondary_on = edges_on && scene.use_secondary_edges;
        SecEdgeArgs early_sa{};
        bool hoisted = false;
        const bool hoist_allowed = !tuning().has(RDR_TUNE_NO_HOIST);          // A/B
        if (hoist_allowed && secondary_on && overlap && scene.diffuse_only && pcg_edge == nullptr && has_lights && B >= 2 && num_active[0].upper > 0) {
            // the dimension the sweep will have reached at the first vertex: 4 per deeper depth that has lanes (device counts)
            if (batch.on) {
                int k = 0;
                bool first = true;
                CountLiveDepthsSeg cl{hoist_dyn, {}, 0, 4, 0};
                auto flush_tables = [&] { cl.n = k; cl.add = first ? 0 : 1; exec::launch(cur_S, cl); first = false; k = 0; };
                for (int d = B - 1; d >= 1; --d) {
                    if (num_active[d].upper <= 0) continue;
                    cl.seg[k++] = seg_of(d);
                    if (k == kDepthGates) flush_tables();
                }
                if (k > 0 || first) flush_tables();
            } else {
            int host_part = 0, k = 0;
            bool first = true;
            CountLiveDepths cl{hoist_dyn, {}, 0, 4, 0};
            auto flush_gates = [&] {
                cl.n = k; cl.add = first ? 0 : 1;
                exec::launch(1, cl);
                first = false; k = 0;
            };
            for (int d = B - 1; d >= 1; --d) {
                if (num_active[d].upper <= 0) continue;
                if (!num_active[d].dev) { host_part += 4; continue; }
                cl.gate[k++] = num_active[d].dev;
                if (k == kDepthGates) flush_gates();
            }
            if (k > 0 || first) flush_gates();
            if (host_part) exec::launch(1, BumpDyn{hoist_dyn, nullptr, host_part});
            }
            early_sa = start_picks(0, 0, hoist_dyn, true, true);
            hoisted = true;
        }
        // The next-event half of a bounce adjoint has nothing to do for a lane whose shadow ray was blocked or whose light sample
        // lies below a horizon (the forward pass left both in the slice's occlusion byte): it runs over the compacted list of the
        // others -- full waves (lane utilisation 0.61 over the whole list).  Order-preserving: the adds keep their order.
        // (its own scratch: the pick phase's compactions may be in flight on another stream of this thread)
        const bool nee_compact = !tuning().has(RDR_TUNE_NO_NEE_COMPACT) && large_forms;
        auto adj_nee = [&](const AdjBounceArgs &ba, exec::Count nA, int d) {
            if (!nee_compact || !vs[d].occl) { launch_v(lean, nA, AdjBounceNee{ba}); return; }
            AdjBounceArgs lit = ba;
            const exec::Count nLit = exec::compact_dev(ba.active, nA, nee_act, KeepNeeLive{vs[d].occl}, nullptr, nullptr, 0, nullptr, 1);
            lit.active = nee_act;
            launch_v(lean, nLit, AdjBounceNee{lit});
        };
        // The continuation half of a bounce adjoint overwrites the lane's record with what flows back through the BSDF-sampled
        // ray.  Two kinds of lanes have only zeros to write there, onto records that ARE zero:
        //  * the continuation ray left the scene and there is no environment light (the record was cleared above and no deeper
        //    vertex wrote it: the lane had none): the stage takes its lanes from the NEXT depth's live-lane list;
        //  * the successor's record is all zeros (AdjState::carries: nothing has been written there since the clear -- every lane
        //    of the deepest vertex, a third of the next one up, a tenth further up) and the ray did not reach an emitter: all the
        //    stage adds is a product with those zeros.  One compaction per depth takes them out.
        // (With an environment light a ray that leaves the scene carries radiance: the full list, and at the deepest vertex the
        // lanes that reached an emitter or the environment.)  Same sums: what is skipped multiplied by zeros and added zeros.
        auto adj_scatter = [&](const AdjBounceArgs &ba, exec::Count nA, int d) {
            AdjBounceArgs part = ba;
            exec::Count n = nA;
            if (!nee_compact) {
                // small frames / RDR_TUNE_NO_NEE_COMPACT (no extra launches): the next depth's list as it is
                if (sd.envmap == nullptr) { part.active = active + (size_t)(d + 1) * stride; n = num_active[d + 1]; }
            } else if (adj.carries) {
                n = exec::compact_dev(active + (size_t)(d + 1) * stride, num_active[d + 1], nee_act,
                                      KeepCarryingContinuation{adj.carries, vs[d + 1].shape, sd.shapes}, nullptr, nullptr, 0, nullptr, 1);
                part.active = nee_act;          // (read by this launch before adj_nee, on the same stream, compacts into it again)
            } else if (d == B - 1) {
                n = exec::compact_dev(ba.active, nA, nee_act, KeepLitContinuation{vs[d + 1].shape, sd.shapes, sd.envmap != nullptr},
                                      nullptr, nullptr, 0, nullptr, 1);
                part.active = nee_act;
            }
            launch_v(lean, n, AdjBounceScatter{part});
        };
        for (int d = B - 1; d >= 0 && has_lights; --d) {
            const exec::Count nA = num_active[d];
            if (nA.upper <= 0) continue;
            const int *act = active + (size_t)d * stride;
            AdjBounceArgs ba{sd, grads.g, rng, dim0 + 7 * d, act, vs[d], vs[d + 1], d_image, nd, radiance_dim, weight, adj};
            bool with_edges = secondary_on;
            if (with_edges && d > 0 && scene.diffuse_only) {
                // Every material is purely diffuse: a path that has left its first vertex carries min_roughness 1 (src/material.h:750-752)
                // and the sampler returns at once for every slot (src/edge.cpp:1396-1401).  Nothing of the pass remains but its
                // sampler bookkeeping: four numbers drawn per slot.
                with_edges = false;
                secondary_pass_consumed(nA, d);
            }
            // The bounce adjoint of this depth, the hierarchical edge pick and the NEE-mode gather do not depend on each other
            // (the edge pass touches the adjoint records only in SecondaryEdgeDerivatives); each of them keeps a fraction of
            // the lanes busy, so they run on three streams and are joined before the records are needed.
            hipStream_t main_stream = exec::ctx().stream;
            const bool early_here = hoisted && d == 0;
            const bool side = overlap && with_edges && !early_here;
            if (side) {
                depth_begin.after(main_stream);
                exec::StreamScope on(exec::side_stream(side_index(0, P)));
                depth_begin.gate(exec::ctx().stream);
                adj_scatter(ba, nA, d);
                adj_nee(ba, nA, d);
                adjoint_done.after(exec::ctx().stream);
            } else {
                adj_scatter(ba, nA, d);
                adj_nee(ba, nA, d);
            }
            if (with_edges) {
                // ---- secondary (shadow / inter-reflection) edges at this vertex, :500-706 ----
                const exec::Count lanes = exec::scaled_count(nA, 2);
                if (chain_mode()) erd_view(seg_of(d));
                const SecEdgeArgs sa = early_here ? early_sa : start_picks(d, edim, nullptr, false, side);
                join_picks();
                debug_dump("sec_mode", sample_id, d, sec_mode, (size_t)nA.upper);
                debug_dump("sec_picks", sample_id, d, sec_picks, sizeof(SecPick) * (size_t)nA.upper);
                launch_v(lean, nA, SecEdgeFinish{sa, sec_mode, sec_picks, d_image, nd, radiance_dim, sec_recs, ea, edge_tmin});
                debug_dump("sec_recs", sample_id, d, sec_recs, sizeof(SecondaryEdgeRec) * (size_t)nA.upper);
                secondary_pass_consumed(nA, d);
                const exec::Count n0 = exec::compact_dev((const int *)nullptr, lanes, elist[0], KeepNonZeroDir{ea.ray, ea.n});
                exec::launch(n0, QueueRays{elist[0], ea, edge_tmin, q.bsdf});
                exec::trace(scene.bvh, q.bsdf, q.h_bsdf, n0, false);
                exec::launch(n0, RecordHits{elist[0], ea, q.h_bsdf});
                if (ea.erd) exec::launch(n0, MirrorSurfDiff{sd, elist[0], ea});
                launch_v(lean, nA, SecondaryEdgeWeights{sd, sec_recs, ea, hit_view(seg_of(d))});
                exec::zero(edge_contrib, sizeof(double) * lanes.upper);
                launch_v(lean, n0, ShadeRecorded{sd, elist[0], ea, esink});
                const exec::Count n1 = exec::compact_dev(elist[0], n0, elist[1], KeepHit{ea.shape});
                trace_edge_paths(rng_edge, edim, n1, nA, d + 1, q, esink, false, seg_of(d));
                if (side) adjoint_done.gate(main_stream);          // the only stage of the edge pass that touches the adjoint records
                exec::launch(nA, SecondaryEdgeDerivatives{sd, grads.g, act, sec_recs, hit_view(seg_of(d), d), edge_contrib, adj});
            }
        }
        auto launch_adj_primary = [&] {
            launch_v(lean, P, AdjPrimary{sd, grads.g, rng, opt.sample_pixel_center, vs[0], d_image, nd, radiance_dim, weight, adj, screen_grad, ch});
        };
        // the camera-vertex adjoint runs beside the primary-edge pass unless both would add to the screen-gradient image
        const bool adj_primary_aside = overlap && screen_grad == nullptr && edges_on && scene.use_primary_edges;
        if (adj_primary_aside) {
            hipStream_t main_stream = exec::ctx().stream;
            depth_begin.after(main_stream);
            exec::StreamScope on(exec::side_stream(side_index(0, P)));
            depth_begin.gate(exec::ctx().stream);
            launch_adj_primary();
            adjoint_done.after(exec::ctx().stream);
        } else {
            launch_adj_primary();
        }
        if (edges_on && scene.use_primary_edges) {
            // ---- primary (camera-visible silhouette) edges, :766-942 ----
            const EdgeSceneD &es = scene.edges->d;
            const int lanes = 2 * P;
            exec::zero(edge_contrib, sizeof(double) * lanes);
            erd_view(nullptr);                  // primary-edge lanes 2 (s P0 + slot) + side ARE entry 2 slot + side of sample s's copy
            SamplePrimaryEdges spe{sd, es, edge_rng_at(rng_edge, edim), edim, d_image, nd, radiance_dim, prim_recs, ea, multipliers};
            if (chain_mode()) spe.batch_P0 = batch.P0;
            launch_v(lean, P, spe);
            edim += 2;
            edge_rng_consumed_n(P, 2);
            const exec::Count n0 = exec::compact_dev((const int *)nullptr, lanes, elist[0], KeepNonZeroDir{ea.ray, ea.n});
            // the sub-paths of the lanes listed in `list` (its length: `n`), from the first intersection to the last bounce;
            // `chain`: their differentials come from the state the previous sample left (else: from their own entries)
            auto sub_paths = [&](int *list, exec::Count n, const double *chain, int *dyn) {
                if (ea.erd) exec::launch(n, LoadLaneDiff{list, ea, chain, chain_n, 2 * batch.P0});
                exec::launch(n, QueueRays{list, ea, nullptr, q.bsdf});
                exec::trace(scene.bvh, q.bsdf, q.h_bsdf, n, false);
                launch_v(lean, n, ShadePrimary{sd, list, ea, q.h_bsdf, psink});
                if (ea.erd) exec::launch(n, MirrorSurfDiff{sd, list, ea});
                const exec::Count n1 = exec::compact_dev(list, n, elist[1], KeepHit{ea.shape});
                trace_edge_paths(rng_edge, edim, n1, P, 0, q, esink, true, nullptr, dyn);
            };
            if (!chain_mode()) {
                sub_paths(elist[0], n0, nullptr, nullptr);
            } else {
                // Chain mode.  The reference's differential buffer is written per slot but read per lane, and most of what a
                // lane reads there is left over from earlier stages -- of this sample, or of EARLIER samples (DESIGN.md section
                // 1, "stale scratch"): for those lanes the samples of a batch depend on each other, one after the other.  So:
                // (A) the lanes whose entry this sample wrote itself (its secondary passes, the slot-indexed write above) run
                //     as one batch, every sample on its own copy of the buffer;
                // (B) the others sample by sample: sample s reads the state sample s - 1 left (`erd_chain`: its copy where it
                //     wrote, else what it found), runs its lanes, and leaves its state for sample s + 1.
                // What a lane does depends on no other lane, so the split changes nothing but the order of launches; the
                // dimension counters of (B) start from where the pass started (`prim_dyn`).
                exec::copy_dev(prim_dyn, edge_dyn, sizeof(int) * kMaxBatch);
                const exec::Count nA = exec::compact_dev(elist[0], n0, elist[2], KeepTouched{ea.erd_touched, 1});
                const exec::Count nD = exec::compact_dev(elist[0], n0, deferred, KeepTouched{ea.erd_touched, 0});
                exec::launch(cur_S + 1, SegOffsets{deferred, nD.dev, nD.upper, 2 * batch.P0, cur_S, deferred_seg});
                static const bool tell_chain = std::getenv("RDR_DEBUG_BATCH") != nullptr;
                if (tell_chain) std::fprintf(stderr, "[render] chain mode: %d of %d primary-edge lanes need the previous sample's scratch (sequential part)\n",
                                             exec::read_count(nD), exec::read_count(n0));
                sub_paths(elist[2], nA, nullptr, nullptr);
                for (int k = 0; k < cur_S; ++k) {
                    exec::launch(2 * batch.P0, ExtractSegment{deferred, deferred_seg, k, elist[0], deferred_count + k});
                    sub_paths(elist[0], exec::Count(deferred_count + k, 2 * batch.P0), erd_chain, prim_dyn);
                    exec::launch(chain_n, ChainAdvance{erd_chain, chain_n, ea.erd, ea.n, ea.erd_touched, k});
                }
            }
            launch_v(lean, P, PrimaryEdgeDerivatives{sd, grads.g, prim_recs, edge_contrib, screen_grad});
        }
        if (adj_primary_aside) adjoint_done.gate(exec::ctx().stream);      // the next sample clears the adjoint records
        if (hp_events) replay_stale_hits(main_rng, vs, active, num_active, stride);
    }

    // Replay (sample batches under an environment light).  An edge ray that reaches the environment reads the hit position an
    // earlier pass left at its entry of the reference's scratch (HitPosView); where that pass belongs to an EARLIER SAMPLE of the
    // batch the value was not known when the sweep came by, and the term was recorded instead of added.  It is linear in that
    // position, and the adjoint stages are linear in (adjoint record, upstream gradient): so now that every sample's passes have
    // run, the recorded terms are put into EMPTY adjoint records -- depth by depth, deepest first, as the sweep met them -- and
    // the same stages, with weight 0 and restricted to the lanes that carry something, take them down the paths to the camera
    // and into the gradient buffers.  The sum is the sequential result up to the order of fp64 additions.  Then the state the
    // next batch finds.  (No event, no work: every launch trims itself to the marked lanes / the event count.)
    void replay_stale_hits(const SamplerD &rng, std::vector<VSlice> &vs, int *active, std::vector<exec::Count> &num_active, int stride) {
        const int lanes = cur_S * batch.P0;
        exec::zero(adj.thr, sizeof(double) * 3 * stride);
        exec::zero(adj.ray_dir, sizeof(double) * 3 * stride);
        exec::zero(adj.point, sizeof(double) * adj_point_doubles * stride);
        exec::zero(replay_live, (size_t)stride);
        const exec::Count n_events(hp_event_count, hp_event_cap);
        const int dim0 = opt.sample_pixel_center ? 0 : 2;
        for (int d = B - 1; d >= 0; --d) {
            if (num_active[d].upper <= 0) continue;
            const int *act = active + (size_t)d * stride;
            AdjBounceArgs ba{sd, grads.g, rng, dim0 + 7 * d, act, vs[d], vs[d + 1], d_image, nd, radiance_dim, 0.0, adj};
            launch_v(lean, num_active[d], AdjBounceScatterLive{AdjBounceScatter{ba}, replay_live});
            exec::launch(n_events, InjectHitEvents{sd, grads.g, hp_events, d, hit_pos, ea.n, hp_written, hp_carry, 2 * batch.P0, adj, replay_live});
        }
        launch_v(lean, lanes, AdjPrimaryLive{AdjPrimary{sd, grads.g, rng, opt.sample_pixel_center, vs[0], d_image, nd, radiance_dim, 0.0,
                                                        adj, screen_grad, ch}, replay_live});
        exec::launch(2 * batch.P0, HitPosCarryAdvance{hp_carry, 2 * batch.P0, cur_S, hit_pos, ea.n, hp_written});
        exec::zero(hp_event_count, sizeof(int));
    }
};

} // namespace

namespace {
// A batched gradient render of an environment-lit scene records the reads of the reference's hit-position scratch that reach
// across the samples of a batch and replays them after the sweep (Backward::replay_stale_hits).  Should the event list
// overflow (two entries per lane: not seen), nothing has been written to the caller's tensors yet (the accumulators are folded
// into them at the very end): the call starts over, one sample per launch.
struct RestartUnbatched {};
// ... and scenes of that shape start that way the next time (an optimisation loop builds a Scene per iteration)
std::atomic<uint64_t> g_unbatchable{0};
uint64_t scene_shape_key(const Scene &scene) {
    uint64_t h = 1469598103934665603ULL;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ULL; };
    mix(scene.shapes.size());
    for (const ShapeD &sh : scene.shapes) { mix((uint64_t)sh.num_vertices); mix((uint64_t)sh.num_triangles); }
    mix(scene.materials.size()); mix(scene.lights.size());
    mix((uint64_t)scene.h_envmap.values.width[0]); mix((uint64_t)scene.h_envmap.values.height[0]);
    return h | 1;
}
void render_once(const Scene &scene, const rdr_render_options &opt, float *image, const float *d_image,
                 const rdr_dscene_desc *d_scene, float *screen_gradient_image, const Tuning &tune, bool envmap_batches);
}

void render(const Scene &scene, const rdr_render_options &opt, float *image, const float *d_image,
            const rdr_dscene_desc *d_scene, float *screen_gradient_image, float * /*debug_image*/) {
    const Tuning tune = resolve_tuning(opt.tuning);
    TuningScope tuning_scope(tune);
    const bool envmap_edges = scene.d.envmap != nullptr && d_image != nullptr && (scene.use_primary_edges || scene.use_secondary_edges);
    const uint64_t key = envmap_edges ? scene_shape_key(scene) : 0;
    bool optimistic = envmap_edges && g_unbatchable.load(std::memory_order_relaxed) != key;
    try {
        render_once(scene, opt, image, d_image, d_scene, screen_gradient_image, tune, optimistic);
    } catch (const RestartUnbatched &) {
        exec::device_sync();
        g_unbatchable.store(key, std::memory_order_relaxed);
        render_once(scene, opt, image, d_image, d_scene, screen_gradient_image, tune, false);
    }
}

namespace {
void render_once(const Scene &scene, const rdr_render_options &opt, float *image, const float *d_image,
                 const rdr_dscene_desc *d_scene, float *screen_gradient_image, const Tuning &tune, bool envmap_batches) {
    if (opt.sampler_type != RDR_SAMPLER_SOBOL && opt.sampler_type != RDR_SAMPLER_INDEPENDENT)
        throw std::runtime_error("render: unknown sampler type");
    if (d_image && !d_scene) throw std::runtime_error("render: d_rendered_image given without d_scene");
    // (the edge build create_scene() started is joined where a worker first needs the structures -- after it has queued the
    //  camera-to-light stages of its first sample, which do not: see run_samples; a forward render never waits for it)
    const CameraD &cam = scene.d.cam;
    const int P = (cam.vp_x1 - cam.vp_x0) * (cam.vp_y1 - cam.vp_y0);
    if (P <= 0) return;
    const int B = opt.max_bounces;
    if (B < 0) throw std::runtime_error("render: max_bounces must be >= 0");
    ChannelLayout lay = layout_of(opt, scene.max_generic_texture_dimension);
    if (lay.radiance_dim < 0 && B > 0)
        throw std::runtime_error("render: max_bounces > 0 needs the radiance channel (the reference writes path "
                                 "contributions at the radiance offset, src/path_contribution.cpp:127-129)");
    const int total_spp = opt.total_samples > 0 ? opt.total_samples : opt.num_samples;
    const double weight = 1.0 / total_spp;
    const bool has_lights = scene.d.num_lights > 0;
    if (2 + 7 * B > kSamplerDims) throw std::runtime_error("render: max_bounces exceeds the Sobol' table");

    PhaseTimer timer(d_image ? "render (backward)" : "render (forward)");
    Arena arena;
    {
        int *ids = arena.get<int>(kMaxChannels);
        exec::upload(ids, opt.channels, sizeof(int) * opt.num_channels);
        lay.ch.id = ids;
    }
    std::unique_ptr<GradStore> grads;
    if (d_image) grads.reset(new GradStore(scene, *d_scene, (size_t)P * (size_t)opt.num_samples));

    uint64_t *pcg_main = nullptr;
    if (opt.sampler_type == RDR_SAMPLER_INDEPENDENT) {
        pcg_main = arena.get<uint64_t>(P);
        exec::launch(P, PcgInit{pcg_main, pcg_stream_seed(opt)});
    }
    const int lean = scene_kind(scene, lay.ch);

    // Several samples in flight (backward pass, lean scenes, Sobol' sampler): the samples of a gradient render are
    // independent -- no image is written, the sampler is stateless, gradient adds are atomics -- and at the sizes
    // optimisation loops run at (256x256, a few spp) one sample is a chain of short, latency-bound launches.  So (i) up to
    // kMaxBatch consecutive samples are rendered as one set of lanes (a "sample batch", see BatchView) while that set stays
    // below 2^19 lanes, and (ii) helper host threads drive batches k, k + workers, ... on their own streams with their own
    // buffers.  The forward image needs its fp32 adds in sample order and stays one sample at a time on one stream.
    // (the camera-vertex adjoint adds to the screen-gradient image with plain read-modify-writes: one worker, no batches, then)
    // Not batched: scenes with mip-mapped textures or an environment light (the reference's stale-scratch reads reach from one
    // sample into the next there, DESIGN.md section 1 -- samples of a batch run side by side), the PCG sampler (stateful), a
    // screen-gradient image (plain read-modify-writes per pixel).
    const bool sobol_plain = screen_gradient_image == nullptr && opt.sampler_type == RDR_SAMPLER_SOBOL && !timer.on;
    // Without either edge estimator (a loop that moves materials, textures or lights only: pyredner switches them off when
    // neither the camera nor a vertex requires a gradient) there is no such scratch and every scene is batched.
    const bool no_edge_passes = !scene.use_primary_edges && !scene.use_secondary_edges;
    // With mip levels (and no environment light) the samples of a batch depend on each other through the reference's
    // differential scratch, lane by lane: "chain mode" (Backward::run_sample, "primary edges") batches what does not and runs the
    // rest sample by sample.  One worker then: the chain runs through the batches in order.
    const bool chain = (scene.has_mipmaps || scene.d.envmap != nullptr) && !no_edge_passes;      // (under an environment light: the hit-position scratch)
    // Under an environment light there is a second scratch of that kind -- the hit positions of the edge rays, read stale by the
    // rays that reach the environment (HitPosView, stages_edge.h): reads that reach across the samples of a batch are recorded
    // and replayed after the sweep (Backward::replay_stale_hits); `envmap_batches` is false only for the second attempt of a
    // call whose event list overflowed (render()).
    const bool batchable = sobol_plain && (lean == kLean || no_edge_passes || scene.d.envmap == nullptr || envmap_batches);
    const bool samples_independent = batchable && d_image != nullptr && image == nullptr;
    // A forward render is batched too -- of any scene: the stale scratch belongs to the edge passes -- : its launches deposit
    // per lane into staging planes and ResolveBatchImage adds them to the image in the reference's order (one stream, batches
    // in sample order).
    const bool forward_batches = sobol_plain && d_image == nullptr && image != nullptr;
    BatchView batch;
    batch.P0 = P; batch.rows = cam.vp_y1 - cam.vp_y0;
    if (samples_independent || forward_batches) {
        // Measured (tools/gpu_batch_grid.sh, tools/gpu_ab_env.sh, bunny_box, profiles/r3_notes.md): a batch costs about the same
        // up to ~131 k lanes and little more up to 524 k; two batches in flight on two host threads beat one of twice the size
        // from 262 k lanes on; and a launch keeps getting cheaper per lane up to several million lanes (every kernel of this
        // path ends with a tail of long lanes: the closest-hit launch takes ~80 us + 0.215 ns per ray) -- the 1024 x 1024
        // benchmark ran 53.2 / 56.8 / 59.7 / 60.0 Msamples/s with 1 / 2 / 4 / 8 samples per launch before the refilling traversal
        // kernel, and runs 61.3 / 62.8 / 64.0 with 4 / 8 / 16 since (closest-hit launch at 0.62 / 0.69 / 0.73 of the roofline).
        // So: everything in one batch while that is <= 2^17 lanes, otherwise batches of up to 2^24 lanes (RDR_BATCH_LANES; the
        // buffers of such a batch are ~4 KB per lane: 64 GB of the 288 GB, which the buffer cache keeps between calls), at
        // least two of them, driven by two workers while a batch is below 2^20 lanes.
        const int batch_cap = std::min(kMaxBatch, tune.batch_samples);
        const long long total = (long long)opt.num_samples * P;
        int want = opt.num_samples;
        if (total > (1 << 17) && samples_independent) want = (opt.num_samples + 1) / 2;
        // Lanes per batch (rdr_tuning::batch_lanes): 2^24 in a process that has the device to itself; 2^22 when another allocator
        // (torch holding a network next to the renderer) has taken more than a tenth of the device's memory -- the buffers of a
        // 2^24-lane batch are ~48 GB, of a 2^22-lane batch 12 GB, for 4 % of throughput at 1024 x 1024 (profiles/r3_notes.md).
        long long lane_cap = tune.batch_lanes;
        if (lane_cap == 0) {
            lane_cap = 1LL << 24;
            if (total > (1LL << 22) && exec::memory_held_by_others() > 0.10) lane_cap = 1LL << 22;
        }
        batch.S = std::max(1, std::min(std::min(batch_cap, want), (int)std::max(1LL, lane_cap / P)));
        // ... and what the device can still give: a worker's buffers are ~(400 (max_bounces + 1) + 1200) bytes per lane (2.9 KB
        // measured at max_bounces 4 with both edge estimators), a forward batch adds its staging planes, a gradient batch the
        // per-lane copy of the upstream gradient; gradient batches below 2^20 lanes are driven by two workers (more when the
        // tuning asks).  A process that shares the GPU with a large torch model gets smaller batches instead of an
        // allocation failure.
        auto bytes_needed = [&](int S_try) {
            const double lanes = (double)S_try * P;
            const int wk = !samples_independent ? 1 : (tune.workers > 0 ? tune.workers : 2);
            // (measured, bunny_box at max_bounces 4: 2.9 KB per lane with ray differentials, 1.98 KB for the lean kernels, which keep
            //  neither them nor the uv / colour adjoints)
            double per_lane = (lean == kLean ? 260.0 * (B + 1) + 660.0 : 400.0 * (B + 1) + 1200.0) * wk;
            if (forward_batches) per_lane += 4.0 * lay.nd * (B + 1);
            if (d_image) per_lane += 4.0 * lay.nd;
            return per_lane * lanes;
        };
        // Buffers that do not fit the buffer cache (exec::pool_cap_bytes: 8 GiB unless the caller raised it) are allocated and
        // released by EVERY call, and device memory that another process has used before is scrubbed when it is handed out:
        // ~25 ms per GB on a box that has been in use (a fresh box allocates 48 GB in no time, which is how this went unnoticed
        // for a round).  The 256-spp benchmark in 16-sample batches (48 GB of buffers) ran at 64.3 Msamples/s on a fresh box and
        // at 50.1 after the test suite had run there; a 32-spp gradient render of the config-5 stand-in at 12.4 against 28.6 in
        // 4-sample batches (profiles/r4_notes.md).  So a batch is as large as fits the cache -- 8 samples of 1024 x 1024 by
        // default (63 Msamples/s), 16 once the caller raises the bound (rdr_set_pool_cap_mb(65536): 64-65).
        if (tune.batch_lanes == 0 && tune.mem_available_mb < 0) {
            const double cap = (double)exec::pool_cap_bytes();
            while (batch.S > 1 && bytes_needed(batch.S) > cap) --batch.S;
        }
        if (tune.mem_available_mb >= 0 || bytes_needed(batch.S) > 1073741824.0) {        // small frames: not worth asking the driver
            const double room = 0.8 * (tune.mem_available_mb >= 0 ? tune.mem_available_mb * 1048576.0 : (double)exec::memory_available());      // (override: tests)
            while (batch.S > 1 && bytes_needed(batch.S) > room) batch.S = (batch.S + 1) / 2;
        }
        batch.on = batch.S > 1;
    }
    const int S = batch.S;
    const int PL = S * P;                         // lanes per launch set
    static const bool tell = std::getenv("RDR_DEBUG_BATCH") != nullptr;
    if (tell) std::fprintf(stderr, "[render] %s: %d samples of %d lanes in batches of %d\n", d_image ? "backward" : "forward", opt.num_samples, P, S);
    const int num_batches = (opt.num_samples + S - 1) / S;
    const float *d_image_lanes = d_image;         // the upstream gradient, indexed by lane: one copy per sample of a batch
    if (batch.on && d_image) {
        float *rep = arena.get<float>((size_t)PL * lay.nd);
        for (int k = 0; k < S; ++k)
            exec::copy_dev(rep + (size_t)k * P * lay.nd, d_image, sizeof(float) * (size_t)P * lay.nd);
        d_image_lanes = rep;
    }
    // forward batches with id channels in the output: which components are assigned rather than accumulated
    const int *assign_flags = nullptr;
    if (batch.on && forward_batches && !lay.ch.radiance_only) {
        std::vector<int> flags((size_t)lay.nd, 0);
        int at = 0;
        bool any_id = false;
        for (int k = 0; k < opt.num_channels; ++k) {
            int one = opt.channels[k];
            const int w = compute_num_channels(&one, 1, scene.max_generic_texture_dimension);
            if (one >= RDR_CH_SHAPE_ID) { for (int j = 0; j < w; ++j) flags[at + j] = 1; any_id = true; }
            at += w;
        }
        if (any_id) {
            int *d_flags = arena.get<int>(flags.size());
            exec::upload(d_flags, flags.data(), sizeof(int) * flags.size());
            assign_flags = d_flags;
        }
    }
    int workers = 1;
    if (samples_independent) workers = std::max(1, std::min(tune.workers > 0 ? tune.workers : exec::sample_workers(PL, num_batches, batch.on), num_batches));
    if (chain) workers = 1;            // (also one sample per launch: the scratch chain runs through the samples in order)
    if (d_image != nullptr) { last_schedule()[0].store(S); last_schedule()[1].store(workers); }

    // Everything one sample (or sample batch) needs between its camera rays and its last gradient add.
    struct Worker {
        Arena arena;
        std::vector<VSlice> vs;
        int *active = nullptr;
        Queues q;
        std::vector<exec::Count> num_active;   // live lanes per depth: device-side counts (upper bound PL)
        int *main_dyn = nullptr;               // PCG only: 7 x (bounces that had lanes), counted on the device
        int *seg = nullptr;                    // sample batches: per depth, where each sample's part of the live-lane list starts
        float *stage = nullptr;                // forward batches: B + 1 planes of PL x nd floats (see ResolveBatchImage)
        std::unique_ptr<Backward> bwd;
    };
    auto make_worker = [&](Worker &w) {
        w.vs.resize(B + 1);
        // (the occlusion byte is what the next-event adjoint compacts by: a forward render keeps none and BounceContrib skips the test)
        for (int d = 0; d <= B; ++d) w.vs[d] = make_slice(w.arena, PL, d < B && d_image_lanes != nullptr, lean != kLean);
        w.active = w.arena.get<int>((size_t)(B + 1) * PL);
        w.q.nee = w.arena.get<rt::RayRec>((size_t)2 * PL); w.q.bsdf = w.arena.get<rt::RayRec>((size_t)2 * PL);
        w.q.h_nee = w.arena.get<rt::HitRec>((size_t)2 * PL); w.q.h_bsdf = w.arena.get<rt::HitRec>((size_t)2 * PL);
        w.q.pos[0] = w.arena.get<int>((size_t)2 * PL); w.q.pos[1] = w.arena.get<int>((size_t)2 * PL);
        w.num_active.assign(B + 2, exec::Count(0));
        if (pcg_main) w.main_dyn = w.arena.get<int>(1);
        if (batch.on) w.seg = w.arena.get<int>((size_t)(B + 1) * (kMaxBatch + 1));
        if (batch.on && forward_batches) w.stage = w.arena.get<float>((size_t)(B + 1) * PL * lay.nd);
    };
    auto make_backward = [&](Worker &w) {          // needs the edge structures (their sizes decide what is allocated)
        scene.edge_data();
        w.bwd.reset(new Backward(scene, opt, *grads, PL, B, d_image_lanes, screen_gradient_image, weight, lay.nd, lay.radiance_dim, lay.ch, batch));
    };
    // batches first, first + stride, ... on the calling thread's stream (a batch = S consecutive samples; S = 1: samples)
    auto run_samples = [&](Worker &w, int first, int stride) {
        std::vector<VSlice> &vs = w.vs;
        int *active = w.active;
        const Queues &q = w.q;
        std::vector<exec::Count> &num_active = w.num_active;
        SceneD sd = scene.d;
        if (batch.on) sd.cam.batch_rows = batch.rows;
        for (int b = first; b < num_batches; b += stride) {
            const int s = b * S;
            const int S_now = std::min(S, opt.num_samples - s);
            const int lanes = S_now * P;
            const int sample_id = opt.sample_offset + s;
            SamplerD rng{scene.sobol_table, opt.seed, sample_id, pcg_main, 0};
            if (batch.on) { rng.batch = S_now; rng.batch_lanes = P; }
            Sink sink{image, nullptr, lay.nd, lay.radiance_dim, weight, lay.ch, nullptr};
            const size_t plane = (size_t)PL * lay.nd;
            auto sink_of = [&](int launch) { Sink k = sink; if (w.stage) k.image = w.stage + (size_t)launch * plane; return k; };
            if (w.stage) exec::zero(w.stage, sizeof(float) * plane * (B + 1));
            // a batch: where each sample's lanes start in the live-lane list of depth d (the list ascends in the lane id)
            auto segments = [&](int d) {
                if (batch.on) exec::launch(S_now + 1, SegOffsets{active + (size_t)d * PL, num_active[d].dev, num_active[d].upper, P, S_now,
                                                                 w.seg + (size_t)d * (kMaxBatch + 1)});
            };

            // ---- camera vertex ----
            launch_v(lean, lanes, GenPrimary{sd, rng, opt.sample_pixel_center, vs[0], q.bsdf});
            exec::trace(scene.bvh, q.bsdf, q.h_bsdf, lanes, false, true);        // camera rays: a coherent queue
            launch_v(lean, lanes, ShadePrimary{sd, nullptr, vs[0], q.h_bsdf, sink_of(0)});
            std::fill(num_active.begin(), num_active.end(), exec::Count(0));
            num_active[0] = exec::compact_dev((const int *)nullptr, lanes, active, KeepHit{vs[0].shape});
            if (d_image) segments(0);
            if (w.main_dyn) exec::zero(w.main_dyn, sizeof(int));

            // ---- bounces (src/pathtracer.cpp:292-390).  The reference stops when no lane is left; here every bounce is
            // queued and one without lanes does nothing (the live-lane counts stay on the device, exec::Count) ----
            int dim = opt.sample_pixel_center ? 0 : 2;
            const int dim_first = dim;
            BounceChain chain;
            chain.fused = fuse_bounces(pcg_main != nullptr, lean, 2 * lanes);
            for (int d = 0; d < B && num_active[d].upper > 0 && has_lights; ++d) {
                num_active[d + 1] = run_bounce(scene, sd, rng, dim, 0, active + (size_t)d * PL, num_active[d],
                                               vs[d], vs[d + 1], q, sink_of(d + 1), active + (size_t)(d + 1) * PL, w.main_dyn, 7,
                                               d == 0,         // shadow rays of the camera vertices: neighbouring origins
                                               &chain, d + 1 < B ? &vs[d + 2] : nullptr, d == B - 1, d + 2 == B);
                if (num_active[d + 1].dev && d_image) segments(d + 1);
                dim += 7;
            }
            if (w.stage) exec::launch(P * lay.nd, ResolveBatchImage{image, w.stage, P, lay.nd, S_now, B + 1, plane, assign_flags, vs[0].shape});

            if (d_image && !w.bwd) make_backward(w);       // first sample of this worker: the GPU is busy with the stages queued above
            if (w.bwd) w.bwd->run_sample(sample_id, rng, vs, active, num_active, q, lanes, PL, S_now, w.seg);
            // a batch of an environment-lit scene: did the list of recorded stale reads overflow?  (Asked after the first and after
            // the last batch of a worker.)
            if (w.bwd && w.bwd->hp_violations && (b == first || b + stride >= num_batches)) {
                int bad = 0;
                exec::download(&bad, w.bwd->hp_violations, sizeof(int));
                if (bad > 0) throw RestartUnbatched();
            }
            // every slot drew dim_first + 7 x (bounces that ran) numbers this sample
            if (pcg_main) exec::launch(P, PcgAdvance{pcg_main, dim_first, w.main_dyn, nullptr});
        }
    };

    Worker w0;
    make_worker(w0);
    if (timer.on) exec::sync();
    timer.lap("buffers, accumulators");
    if (workers == 1) {
        run_samples(w0, 0, 1);
    } else {
        exec::sync();                          // accumulators zeroed, tables uploaded: visible to the other streams
        for (int k = 1; k < workers; ++k)
            exec::SecondThread::get(k - 1).start([&, k] {
                TuningScope helper_scope(tune);        // (thread-local: the helper renders under the call's tuning)
                Worker w;
                make_worker(w);
                run_samples(w, k, workers);
                exec::sync();
            });
        std::exception_ptr failure;
        try { run_samples(w0, 0, workers); } catch (...) { failure = std::current_exception(); }
        for (int k = 1; k < workers; ++k) {
            try { exec::SecondThread::get(k - 1).wait(); } catch (...) { if (!failure) failure = std::current_exception(); }
        }
        if (failure) std::rethrow_exception(failure);
    }
    if (timer.on) exec::sync();
    timer.lap("samples");
    if (grads) grads->flush();
    exec::sync();
    timer.lap("gradient flush");
}
} // namespace

} // namespace rdr

// Which transcendental functions this build of the stage kernels calls (include/redner_amd.h: rdr_libm_exact)
extern "C" int rdr_libm_exact(void) {
#ifdef RDR_PLATFORM_LIBM
    return 0;
#else
    return 1;
#endif
}
