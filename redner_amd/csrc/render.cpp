// render.cpp -- host driver of the wavefront differentiable path tracer (rdr_render()).
//
// Counterpart of render() in src/pathtracer.cpp:177-958: per Sobol' sample it runs the forward
// wavefront, and -- when an image gradient is given -- re-walks the stored path backwards,
// adding the two edge-sampling estimators.  The structure of the loops (which lanes are alive,
// which Sobol' dimensions each draw consumes, in which order contributions reach the image) is
// the reference's, because sample-exact parity depends on it; how a bounce is cut into kernels
// and what is kept in HBM is ours (stages_fwd.h / stages_bwd.h / stages_edge.h).
#include "render.h"
#include "stages_fwd.h"
#include "stages_bwd.h"
#include "stages_edge.h"
#include <algorithm>
#include <memory>
#include <stdexcept>
#include <vector>

namespace rdr {

namespace {

// Device arena: one allocation per render() call, carved into typed arrays.
struct Arena {
    std::vector<void *> blocks;
    template <class T> T *get(size_t count) {
        T *p = (T *)exec::dmalloc(sizeof(T) * (count ? count : 1));
        blocks.push_back(p);
        return p;
    }
    ~Arena() { for (void *p : blocks) exec::dfree(p); }
};

VSlice make_slice(Arena &a, int n, bool with_occl) {
    VSlice v;
    v.n = n;
    v.ray = a.get<double>((size_t)6 * n);
    v.rdiff = a.get<double>((size_t)12 * n);
    v.shape = a.get<int>(n); v.tri = a.get<int>(n);
    v.thr = a.get<double>((size_t)3 * n);
    v.mrough = a.get<double>(n);
    v.occl = with_occl ? a.get<unsigned char>(n) : nullptr;
    return v;
}

struct Queues {
    rt::RayRec *nee, *bsdf;
    rt::HitRec *h_nee, *h_bsdf;
};

struct ChannelLayout { int nd, radiance_dim; };
ChannelLayout layout_of(const rdr_render_options &o, int max_generic) {
    ChannelLayout l{0, -1};
    int d = 0;
    for (int i = 0; i < o.num_channels; ++i) {
        if (o.channels[i] == RDR_CH_RADIANCE) {
            if (l.radiance_dim != -1) throw std::runtime_error("Duplicated radiance channel");   // src/channels.cpp:24-26
            l.radiance_dim = d;
        }
        int one = o.channels[i];
        int w = compute_num_channels(&one, 1, max_generic);
        if (w < 0) throw std::runtime_error("render: unknown channel id");
        d += w;
    }
    l.nd = d;
    return l;
}

// One NEE + BSDF bounce over the live lanes of `v`; fills `vn` and the next live-lane list.
int run_bounce(const Scene &scene, const SobolD &rng, int dim, int rng_shift,
               const int *active, int num_active, const VSlice &v, const VSlice &vn,
               const Queues &q, const Sink &sink, int *next_active) {
    exec::launch(num_active, BounceSample{scene.d, rng, dim, rng_shift, active, v, vn, q.nee, q.bsdf});
    exec::trace(scene.bvh, q.nee, q.h_nee, num_active, true);
    exec::trace(scene.bvh, q.bsdf, q.h_bsdf, num_active, false);
    exec::launch(num_active, BounceContrib{scene.d, rng, dim, rng_shift, active, v, vn, q.h_nee, q.h_bsdf, sink});
    return exec::compact(active, num_active, next_active, KeepHit{vn.shape});
}

} // namespace

void render(const Scene &scene, const rdr_render_options &opt, float *image, const float *d_image,
            const rdr_dscene_desc *d_scene, float *screen_gradient_image, float * /*debug_image*/) {
    if (opt.sampler_type != RDR_SAMPLER_SOBOL)
        throw std::runtime_error("render: only SamplerType.sobol is implemented (independent/PCG: SURVEY.md section 8f row 4)");
    if (d_image && !d_scene) throw std::runtime_error("render: d_rendered_image given without d_scene");
    const CameraD &cam = scene.d.cam;
    const int P = (cam.vp_x1 - cam.vp_x0) * (cam.vp_y1 - cam.vp_y0);
    if (P <= 0) return;
    const int B = opt.max_bounces;
    if (B < 0) throw std::runtime_error("render: max_bounces must be >= 0");
    ChannelLayout lay = layout_of(opt, scene.max_generic_texture_dimension);
    if (lay.radiance_dim < 0 || lay.nd != 3)
        throw std::runtime_error("render: only the radiance channel is implemented so far (G-buffer channels: SURVEY.md section 8f row 2)");
    const int total_spp = opt.total_samples > 0 ? opt.total_samples : opt.num_samples;
    const double weight = 1.0 / total_spp;
    const bool has_lights = scene.d.num_lights > 0;
    if (2 + 7 * B > kSobolDims) throw std::runtime_error("render: max_bounces exceeds the Sobol' table");

    Arena arena;
    std::vector<VSlice> vs(B + 1);
    for (int d = 0; d <= B; ++d) vs[d] = make_slice(arena, P, d < B);
    int *active = arena.get<int>((size_t)(B + 1) * P);
    Queues q;
    q.nee = arena.get<rt::RayRec>((size_t)2 * P); q.bsdf = arena.get<rt::RayRec>((size_t)2 * P);
    q.h_nee = arena.get<rt::HitRec>((size_t)2 * P); q.h_bsdf = arena.get<rt::HitRec>((size_t)2 * P);
    std::vector<int> num_active(B + 2, 0);

    std::unique_ptr<Backward> bwd;
    if (d_image) bwd.reset(new Backward(scene, opt, *d_scene, P, B, d_image, screen_gradient_image, weight, lay.nd, lay.radiance_dim));

    for (int s = 0; s < opt.num_samples; ++s) {
        const int sample_id = opt.sample_offset + s;
        SobolD rng{scene.sobol_table, opt.seed, sample_id};
        Sink sink{image, nullptr, lay.nd, lay.radiance_dim, weight};

        // ---- camera vertex ----
        exec::launch(P, GenPrimary{scene.d, rng, opt.sample_pixel_center, vs[0], q.bsdf});
        exec::trace(scene.bvh, q.bsdf, q.h_bsdf, P, false);
        exec::launch(P, ShadePrimary{scene.d, nullptr, vs[0], q.h_bsdf, sink});
        std::fill(num_active.begin(), num_active.end(), 0);
        num_active[0] = exec::compact((const int *)nullptr, P, active, KeepHit{vs[0].shape});

        // ---- bounces (src/pathtracer.cpp:292-390) ----
        int dim = opt.sample_pixel_center ? 0 : 2;
        for (int d = 0; d < B && num_active[d] > 0 && has_lights; ++d) {
            num_active[d + 1] = run_bounce(scene, rng, dim, 0, active + (size_t)d * P, num_active[d],
                                           vs[d], vs[d + 1], q, sink, active + (size_t)(d + 1) * P);
            dim += 7;
        }

        if (bwd) bwd->run_sample(sample_id, vs, active, num_active, q);
    }
    if (bwd) bwd->flush();
    exec::sync();
}

} // namespace rdr
