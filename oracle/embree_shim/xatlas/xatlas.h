// Minimal declaration so the reference's src/automatic_uv_map.h (which redner.cpp includes) parses.
// UV atlas generation is out of scope; the oracle build replaces
// automatic_uv_map.cpp with oracle/embree_shim/uvstub.cpp.
#pragma once
namespace xatlas {
struct Atlas { int unused; };
inline Atlas *Create() { return new Atlas{0}; }
inline void Destroy(Atlas *a) { delete a; }
}
