// render.h -- entry point of the host driver (render.cpp).
#pragma once
#include "scene.h"
namespace rdr {
void render(const Scene &scene, const rdr_render_options &opt, float *image, const float *d_image,
            const rdr_dscene_desc *d_scene, float *screen_gradient_image, float *debug_image);
}
