// Out-of-scope preprocessing entry points bound by the reference's redner.cpp:241-255.
#include "automatic_uv_map.h"
#include <stdexcept>
std::vector<int> automatic_uv_map(const std::vector<UVTriMesh> &, TextureAtlas &, bool) {
    throw std::runtime_error("automatic_uv_map: xatlas is not part of the oracle build");
}
void copy_texture_atlas(const TextureAtlas &, std::vector<UVTriMesh> &) {
    throw std::runtime_error("copy_texture_atlas: xatlas is not part of the oracle build");
}
