// edges.h -- edge-sampling structures (placeholder until the edge estimator lands).
#pragma once
namespace rdr {
struct Scene;
struct EdgeData;
EdgeData *build_edge_data(Scene &scene);
void delete_edge_data(EdgeData *e);
}
