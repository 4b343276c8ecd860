// oracle_module.cpp -- the CPU oracle behind the reference's pybind11 module surface.
//
// TEST INFRASTRUCTURE ONLY (like everything under oracle/): only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline / dropin_host legs load it.  PARITY STATUS: as pp_oracle.h -- parity unpinned against the real Boost build.
//
// /root/reference data/pillars.cpp:429-435 exports `create_pillars` and `make_ious` from a pybind11 module; BASELINE.md
// section 4 asks for the CPU baseline to be timed "through the same pybind11 signatures".  This is that surface for the
// oracle's reference-style C functions (oracle/pp_oracle.c: hash map of heap nodes, per-point allocation, pillars.cpp:236-398;
// the gate loop of pillars.cpp:400-427): module `pillars_oracle`, the same two names, positional signatures
// (pillars.cpp:236-249, :400-404), array_t<double> arguments with forcecast, None returned, outputs mutated in place.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <stdexcept>

#include "pp_oracle.h"

namespace py = pybind11;
using arr = py::array_t<double, py::array::forcecast>;

namespace {
void raise_for(int rc, const char *what) {
  if (rc == PPO_OK) return;
  if (rc == PPO_ERR_INDEX) throw py::index_error(what);  // what pybind11's .at() raises in the reference
  throw py::value_error(what);
}
}  // namespace

PYBIND11_MODULE(pillars_oracle, m) {
  m.doc() = "point pillars data prep functions (CPU oracle, test infrastructure)";
  m.def("make_ious", [](arr a_corners, arr g_corners, arr a_centers, arr g_centers, arr ious) {
    if (a_corners.ndim() != 3 || g_corners.ndim() != 3 || a_centers.ndim() != 2 || g_centers.ndim() != 2 || ious.ndim() != 2)
      throw py::index_error("make_ious: wrong number of dimensions");
    const int64_t A = a_corners.shape(0), G = g_corners.shape(0);
    if (a_centers.shape(0) < A || g_centers.shape(0) < G || ious.shape(0) < A || ious.shape(1) < G)
      throw py::index_error("make_ious: shape mismatch");
    raise_for(ppo_make_ious(a_corners.data(), A, a_corners.strides(0), a_corners.strides(1), a_corners.strides(2),
                            g_corners.data(), G, g_corners.strides(0), g_corners.strides(1), g_corners.strides(2),
                            a_centers.data(), a_centers.strides(0), a_centers.strides(1), g_centers.data(),
                            g_centers.strides(0), g_centers.strides(1), ious.mutable_data(), ious.strides(0),
                            ious.strides(1)),
              "make_ious");
  });
  m.def("create_pillars", [](arr points, arr tensor, arr indices, int max_points_per_pillar, int max_pillars,
                             double x_step, double y_step, double x_min, double y_min, double z_min, double x_max,
                             double y_max, double z_max, double canvas_height) {
    if (points.ndim() != 2 || tensor.ndim() != 3 || indices.ndim() != 2)
      throw py::index_error("create_pillars: wrong number of dimensions");
    int64_t cells = 0;
    raise_for(ppo_create_pillars(points.data(), points.shape(0), points.strides(0), points.strides(1),
                                 tensor.mutable_data(), tensor.shape(0), tensor.shape(1), tensor.shape(2),
                                 tensor.strides(0), tensor.strides(1), tensor.strides(2), indices.mutable_data(),
                                 indices.shape(0), indices.shape(1), indices.strides(0), indices.strides(1),
                                 max_points_per_pillar, max_pillars, x_step, y_step, x_min, y_min, z_min, x_max, y_max,
                                 z_max, canvas_height, PPO_ORDER_HASH /* the reference-style hash map */, &cells),
              "create_pillars");
  });
}
