// sobol.h -- the scrambled Sobol' stream the renderer draws every random number from.
//
// Behavioural spec: src/sobol_sampler.cpp:10-29 (per-slot scramble = hash64shift(seed<<32 | slot)),
// :62-76 (52-bit Gruenschloss matrices, XOR over set bits of the sample index), :97-214 (dimension
// bookkeeping: camera 2, light 4, bsdf 3, primary edge 2, secondary edge 4 numbers per draw).
// The sampler is stateless: value = f(sample index, dimension, scramble[slot]); kernels regenerate
// numbers instead of storing them (the reference stores them in PathBuffer, src/pathtracer.cpp:48-53).
// The direction-number table (1024 dims x 52 u64) is data shipped in redner_amd/data/sobol_1024x52.u64.
#pragma once
#include <stdint.h>

namespace rdr {

constexpr int kSobolBits = 52;
constexpr int kSamplerDims = 1024;
constexpr int kSobolTableWords = kSobolBits * kSamplerDims;

RDR_FN uint64_t hash64shift(uint64_t key) {
    key = (~key) + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

RDR_FN uint64_t sobol_scramble(uint64_t seed, int slot) { return hash64shift((seed << 32) | (uint64_t)slot); }

// The XOR over the set bits of the sample index (src/sobol_sampler.cpp:62-76).  The eight direction numbers of the low byte are
// loaded TOGETHER and unconditionally and masked by their bit -- the same XORs in the same order; the reference-shaped loop (a
// conditional load per set bit, each waited for before the next) cost a kernel 4 dependent L2 round trips per number and 7
// numbers per bounce.  Sample indices above 255 take the loop for their remaining bits.
RDR_FN double sobol_value(const uint64_t *matrices, uint64_t index, uint32_t dim, uint64_t scramble) {
    uint64_t r = scramble & ~-(1ULL << kSobolBits);
    const uint64_t *m = matrices + (size_t)dim * kSobolBits;
    uint64_t w[8];
    for (int k = 0; k < 8; ++k) w[k] = m[k];
    for (int k = 0; k < 8; ++k) r ^= w[k] & (0ULL - ((index >> k) & 1ULL));
    index >>= 8;
    for (uint32_t i = 8; index; index >>= 1, ++i)
        if (index & 1) r ^= m[i];
    return r * (1.0 / (1ULL << kSobolBits));
}

// ---- PCG32 ("independent" sampler) -----------------------------------------------------------------
// Behavioural spec: src/pcg_sampler.cpp:8-50 (XSH-RR output, per-slot stream inc = 2*(slot+1)+1, seeding),
// :76-146 (every next_*_samples call advances the state of each slot in the view by 2/4/3/2/4 numbers).
// The reference stores the drawn numbers; here a number is regenerated from the slot's state at the start
// of the current draw group plus a skip-ahead, so stages stay stateless and the host advances the states.
constexpr uint64_t kPcgMult = 6364136223846793005ULL;
RDR_FN uint64_t pcg_inc(int slot) { return ((((uint64_t)slot + 1) << 1u) | 1u); }
RDR_FN uint64_t pcg_step(uint64_t state, uint64_t inc) { return state * kPcgMult + (inc | 1); }
RDR_FN uint64_t pcg_seed_state(uint64_t seed, int slot) {
    uint64_t inc = pcg_inc(slot);
    uint64_t st = pcg_step(0U, inc);
    st += (0x853c49e6748fea9bULL + seed);
    return pcg_step(st, inc);
}
RDR_FN uint64_t pcg_advance(uint64_t state, uint64_t inc, uint32_t delta) {     // O(log delta) LCG skip-ahead
    uint64_t acc_mult = 1, acc_plus = 0, cur_mult = kPcgMult, cur_plus = inc | 1;
    while (delta > 0) {
        if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    return acc_mult * state + acc_plus;
}
RDR_FN double pcg_output_double(uint64_t oldstate) {
    uint32_t xorshifted = uint32_t(((oldstate >> 18u) ^ oldstate) >> 27u);
    uint32_t rot = uint32_t(oldstate >> 59u);
    uint32_t u = uint32_t((xorshifted >> rot) | (xorshifted << ((-rot) & 31)));
    union { uint64_t u; double d; } x;
    x.u = ((uint64_t)u << 20) | 0x3ff0000000000000ULL;
    return x.d - 1.0;
}

// One sampler "view".  Sobol': which table, which sample of the sequence, which seed.  PCG (pcg_state != null):
// per-slot states valid for dimension `pcg_base`; draw(slot, dim) is the (dim - pcg_base)-th number after it.
// `dyn` (optional, device memory): the part of the dimension counter the host does not know -- the edge sampler advances by
// 7 per EXECUTED bounce of an edge sub-path, and how many are executed depends on live-lane counts that stay on the device
// (render.cpp).  A draw's dimension is `dim + *dyn`; PCG draws are relative to the draw group's start, so they ignore it.
// Sample batches (Sobol' only; render.cpp): the slots of `batch` consecutive samples in one view.  Either every sample owns
// `batch_lanes` consecutive slots (slot v = slot v % batch_lanes of sample sample_id + v / batch_lanes: pixels, primary-edge
// slots), or -- `seg` set -- the slots are the concatenated compacted lists of the samples (seg[s] = where sample s's list
// starts: the secondary-edge sampler numbers its slots by compacted rank WITHIN a sample, src/pathtracer.cpp:504-505).  `dyn`
// then holds one counter per sample of the batch.
struct SamplerD {
    const uint64_t *matrices;
    uint64_t seed;
    int sample_id;
    const uint64_t *pcg_state;
    int pcg_base;
    const int *dyn = nullptr;
    int batch = 0, batch_lanes = 0;
    const int *seg = nullptr;
    // What a slot's numbers share -- which sample of the batch the slot belongs to, its scramble, the dynamic part of its
    // dimension counter: resolved ONCE per lane (`lane(slot)`), then every number of the lane is one `draw(lane, dim)`.  (A
    // stage draws 2 ... 7 numbers per lane; resolving per number repeated an integer division, the 64-bit hash and the segment
    // search each time.)
    struct Lane { uint64_t index, scramble; int slot, dyn; };
    RDR_FN Lane lane(int slot) const {
        Lane l;
        l.slot = slot; l.index = 0; l.scramble = 0; l.dyn = 0;
        if (pcg_state) return l;
        int s = 0;
        if (seg) {
            for (int k = 1; k < batch; ++k) if (slot >= seg[k]) s = k;
            slot -= seg[s];
        } else if (batch_lanes > 0) {
            s = slot / batch_lanes;
            slot -= s * batch_lanes;
        }
        l.index = (uint64_t)(sample_id + s);
        l.scramble = sobol_scramble(seed, slot);
        l.dyn = dyn ? dyn[s] : 0;
        return l;
    }
    RDR_FN double draw(const Lane &l, int dim) const {
        if (pcg_state) return pcg_output_double(pcg_advance(pcg_state[l.slot], pcg_inc(l.slot), (uint32_t)(dim - pcg_base)));
        return sobol_value(matrices, l.index, (uint32_t)(dim + l.dyn), l.scramble);
    }
    RDR_FN double draw(int slot, int dim) const { return draw(lane(slot), dim); }
};

// Host-launched maintenance of the PCG states.
struct PcgInit {
    uint64_t *state; uint64_t seed;
    RDR_FN void operator()(int slot) const { state[slot] = pcg_seed_state(seed, slot); }
};
// `count` numbers, plus `*extra` more (device memory, optional); only if `*gate > 0` when a gate is given (a draw group of a
// bounce that had no lanes to run was never drawn).
struct PcgAdvance {
    uint64_t *state; int count;
    const int *extra = nullptr, *gate = nullptr;
    RDR_FN void operator()(int slot) const {
        if (gate && *gate <= 0) return;
        state[slot] = pcg_advance(state[slot], pcg_inc(slot), (uint32_t)(count + (extra ? *extra : 0)));
    }
};

// Sobol' dimension bookkeeping that depends on live-lane counts the host never sees (`gate`: a device-side count).
// The reference's backward sweep skips a path depth without live lanes BEFORE its edge sampler draws anything
// (src/pathtracer.cpp:432-436), so the 4 numbers of a secondary-edge pass are consumed only by depths that have lanes.
struct BumpDyn {                    // *dyn += inc if the depth had lanes (gate null: it had)
    int *dyn; const int *gate; int inc;
    RDR_FN void operator()(int) const { if (!gate || *gate > 0) *dyn += inc; }
};
constexpr int kDepthGates = 16;
struct CountLiveDepths {            // *out (+)= inc x #{gates whose count is positive}
    int *out; const int *gate[kDepthGates]; int n, inc, add;
    RDR_FN void operator()(int) const {
        int c = add ? *out : 0;
        for (int i = 0; i < kDepthGates; ++i) if (i < n && *gate[i] > 0) c += inc;
        *out = c;
    }
};

// ---- sample batches: per-sample bookkeeping over sorted lane lists (launched with batch (+ 1) lanes) -----------------------
constexpr int kMaxBatch = 16;
// out[s] = position of the first entry >= s * lanes_per_sample in the ascending list (out[batch] = its length)
struct SegOffsets {
    const int *list; const int *count; int upper, lanes_per_sample, batch; int *out;
    RDR_FN void operator()(int s) const {
        int n = *count; n = n < upper ? n : upper;
        const int key = s * lanes_per_sample;
        int lo = 0, hi = n;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (list[mid] < key) lo = mid + 1; else hi = mid; }
        out[s] = s >= batch ? n : lo;
    }
};
// dyn[s] += inc for every sample whose segment of `seg` is not empty
struct BumpDynSeg {
    int *dyn; const int *seg; int inc;
    RDR_FN void operator()(int s) const { if (seg[s + 1] > seg[s]) dyn[s] += inc; }
};
// dyn[s] += inc for every sample that owns an entry of the ascending edge-lane list: sample s owns the lanes
// [2 seg[s], 2 seg[s + 1]) (two lanes per slot of its compacted list) or, seg null, [s * lanes_per_sample, (s + 1) * lanes_per_sample)
struct BumpDynList {
    int *dyn; const int *list; const int *count; int upper; const int *seg; int lanes_per_sample, inc;
    RDR_FN int first_at_least(int key, int n) const {
        int lo = 0, hi = n;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (list[mid] < key) lo = mid + 1; else hi = mid; }
        return lo;
    }
    RDR_FN void operator()(int s) const {
        int n = *count; n = n < upper ? n : upper;
        const int b0 = seg ? 2 * seg[s] : s * lanes_per_sample, b1 = seg ? 2 * seg[s + 1] : (s + 1) * lanes_per_sample;
        if (first_at_least(b1, n) > first_at_least(b0, n)) dyn[s] += inc;
    }
};
// Forward image of a sample batch: every launch of the batch wrote its contributions per LANE into its own plane of `stage`
// (plane k = launch k of the sample: first hit, bounce 1, ...; lanes_stride x nd floats each, zero where nothing was added);
// one lane per pixel component adds them to the image in the order the reference's launches add them -- sample by sample,
// launch by launch (src/pathtracer.cpp:283,378) -- so the fp32 sums round exactly as they do one sample at a time.
// Components of the id channels (shape / triangle / material id) are ASSIGNED by the first-hit stage of every sample whose
// camera ray hits something (src/primary_contribution.cpp: the last sample wins): `assign[c]` != 0 marks them, `first_shape`
// (the batch's first-hit shape ids per lane) says which samples wrote.
struct ResolveBatchImage {
    float *image; const float *stage; int pixels, nd, samples, planes; size_t plane_stride;
    const int *assign; const int *first_shape;
    RDR_FN void operator()(int i) const {
        const int pixel = i / nd, c = i - pixel * nd;
        float acc = image[i];
        if (assign && assign[c]) {
            // ... and what later launches of the same sample ADD to such a component stays on top of it: with the radiance
            // channel listed after them, the reference's bounce contributions land on the component whose index equals the
            // radiance channel's position in the list (src/channels.cpp:27), which may be an id
            for (int s = 0; s < samples; ++s) {
                const size_t at = ((size_t)s * pixels + pixel) * nd + c;
                if (first_shape[(size_t)s * pixels + pixel] >= 0) acc = stage[at];
                for (int k = 1; k < planes; ++k) acc += stage[(size_t)k * plane_stride + at];
            }
        } else {
            for (int s = 0; s < samples; ++s)
                for (int k = 0; k < planes; ++k)
                    acc += stage[(size_t)k * plane_stride + ((size_t)s * pixels + pixel) * nd + c];
        }
        image[i] = acc;
    }
};

// out[s] = inc x #{tables whose segment s is not empty}
struct CountLiveDepthsSeg {
    int *out; const int *seg[kDepthGates]; int n, inc, add;
    RDR_FN void operator()(int s) const {
        int c = add ? out[s] : 0;
        for (int i = 0; i < kDepthGates; ++i) if (i < n && seg[i][s + 1] > seg[i][s]) c += inc;
        out[s] = c;
    }
};

} // namespace rdr
