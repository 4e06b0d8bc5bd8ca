#include "edges.h"
namespace rdr {
EdgeData *build_edge_data(Scene &) { return nullptr; }
void delete_edge_data(EdgeData *) {}
}
