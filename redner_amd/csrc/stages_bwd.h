#pragma once
#include "stages_fwd.h"
#include "scene.h"
#include <stdexcept>
namespace rdr {

struct Backward {
    Backward(const Scene &, const rdr_render_options &, const rdr_dscene_desc &, int, int, const float *, float *, double, int, int) {
        throw std::runtime_error("backward pass not implemented yet");
    }
    template <class Q> void run_sample(int, std::vector<VSlice> &, int *, std::vector<int> &, const Q &) {}
    void flush() {}
};
}
