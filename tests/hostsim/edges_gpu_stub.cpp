// TEST INFRASTRUCTURE: the CPU debugging harness has no kernels, so the two entry points of redner_amd/csrc/edges_gpu.cpp are
// stubs here (exec::kDeviceEdgeTrees is false in the harness's exec.h: they are never reached).
#include "edges.h"
#include <stdexcept>
namespace rdr {
void build_edge_trees_device(EdgeData &) { throw std::runtime_error("harness: no device edge builder"); }
void download_edge_trees(EdgeData &) { throw std::runtime_error("harness: no device edge builder"); }
void gather_hierarchy_device(EdgeData &) { throw std::runtime_error("harness: no device edge builder"); }
void drop_gather_cache() {}
}

#include "bvh_gpu.h"
namespace rt {
BvhDev::~BvhDev() {}
void build_tri_bvh_device(const void *, const int *, int, const BvhBuildParams &, BvhDev &) { throw std::runtime_error("harness: no device hierarchy builder"); }
void build_box_bvh_device(const float *, int, const BvhBuildParams &, BvhDev &) { throw std::runtime_error("harness: no device hierarchy builder"); }
void refit_tri_bvh_device(const BvhDev &, const void *, BvhDev &) { throw std::runtime_error("harness: no device hierarchy builder"); }
void refit_box_bvh_device(const BvhDev &, const float *, BvhDev &) { throw std::runtime_error("harness: no device hierarchy builder"); }
}
