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
constexpr int kSobolDims = 1024;
constexpr int kSobolTableWords = kSobolBits * kSobolDims;

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

RDR_FN double sobol_value(const uint64_t *matrices, uint64_t index, uint32_t dim, uint64_t scramble) {
    uint64_t r = scramble & ~-(1ULL << kSobolBits);
    for (uint32_t i = dim * kSobolBits; index; index >>= 1, ++i)
        if (index & 1) r ^= matrices[i];
    return r * (1.0 / (1ULL << kSobolBits));
}

// One sampler "view": which table, which sample of the sequence, which seed.
struct SobolD {
    const uint64_t *matrices;
    uint64_t seed;
    int sample_id;
    RDR_FN double draw(int slot, int dim) const {
        return sobol_value(matrices, (uint64_t)sample_id, (uint32_t)dim, sobol_scramble(seed, slot));
    }
};

} // namespace rdr
