// pillars_module.cpp -- the reference's pybind11 module `pillars`, on libpp_hip.so.
//
// /root/reference data/pillars.cpp:429-435 exports exactly two functions from a module named
// `pillars` with the doc string "point pillars data prep functions"; data/dataset.py:6 and
// utils/box_utils.py:11 import it as `data.pillars` (install_mods.sh:8-10 builds
// pillars*.so and moves it into data/).  This file is that module for the HIP path: the same
// names, positional signatures, in-place semantics and exception types, a thin host-only C++
// layer over the C ABI of include/pp_hip.h (built with g++, linked against libpp_hip.so).
// Copy the built pillars*.so next to libpp_hip.so into the reference's data/ directory and
// nothing else changes.
//
// Documented tightenings (same as the ctypes mirror, pillars.py): outputs must be genuine
// writable float64 arrays (the reference's forcecast would write into a temporary and lose
// the results), wrong corner winding raises ValueError instead of std::exit(1), pillar order
// is deterministic (env PP_PILLAR_ORDER: 0 row-major, 1 scrambled = default), NaN points are dropped.
// HIP is initialised on the first call, never at import (DataLoader workers: use spawn).

#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <unistd.h>

#include <cstdlib>
#include <mutex>
#include <string>

#include "pp_hip.h"

namespace py = pybind11;

namespace {

pp_ctx_t *g_ctx = nullptr;
pid_t g_pid = 0;
// The context (staging buffers, one stream) serves one call at a time; the GIL is released around the calls below, so
// Python threads are serialised here.  (The reference holds the GIL for its whole call: its callers never overlap.)
std::mutex g_call_mutex;

// (called with g_call_mutex held and the GIL released: reports through *err instead of throwing)
pp_ctx_t *context_nothrow(std::string *err) {
  if (!g_ctx || g_pid != getpid()) {  // a forked child must not reuse the parent's context
    const char *dev = std::getenv("PP_HIP_DEVICE");
    pp_ctx_t *c = nullptr;
    if (pp_ctx_create(dev ? std::atoi(dev) : 0, &c) != PP_OK) {
      *err = pp_last_error();
      return nullptr;
    }
    g_ctx = c;
    g_pid = getpid();
  }
  return g_ctx;
}

[[noreturn]] void raise_for(int rc, const char *what, const std::string &detail) {
  const std::string msg = std::string(what) + ": " + detail;
  switch (rc) {
    case PP_ERR_INDEX:
      throw py::index_error(msg);  // what pybind11's bounds-checked .at() raises
    case PP_ERR_VALUE:
    case PP_ERR_WINDING:
      throw py::value_error(msg);
    default:
      throw std::runtime_error(msg);
  }
}

using in_array = py::array_t<double, py::array::forcecast>;  // inputs: any dtype/strides, like the reference

py::array out_array(py::object o, const char *name, int ndim) {
  if (!py::isinstance<py::array>(o)) throw py::type_error(std::string(name) + " must be a numpy.ndarray");
  py::array a = py::reinterpret_borrow<py::array>(o);
  if (!py::isinstance<py::array_t<double>>(a) || !a.dtype().is(py::dtype::of<double>()))
    throw py::type_error(std::string(name) +
                         " must be a float64 numpy.ndarray: the reference would write into a "
                         "converted temporary and lose every result");
  if (!a.writeable()) throw py::type_error(std::string(name) + " must be writable");
  if (a.ndim() != ndim)
    throw py::index_error(std::string(name) + ": index dimension mismatch; " + std::to_string(ndim) +
                          " expected, got " + std::to_string(a.ndim()));
  return a;
}

void need_ndim(const py::array &a, const char *name, int ndim) {
  if (a.ndim() != ndim)
    throw py::index_error(std::string(name) + ": index dimension mismatch; " + std::to_string(ndim) +
                          " expected, got " + std::to_string(a.ndim()));
}

void create_pillars(in_array points, py::object tensor_o, py::object indices_o,
                    int max_points_per_pillar, int max_pillars, double x_step, double y_step,
                    double x_min, double y_min, double z_min, double x_max, double y_max,
                    double z_max, double canvas_height) {
  need_ndim(points, "points", 2);
  py::array tensor = out_array(tensor_o, "tensor", 3);
  py::array indices = out_array(indices_o, "indices", 2);
  if (points.shape(0) > 0 && points.shape(1) < 4)
    throw py::index_error("points: index 3 is out of bounds for axis 1");
  pp_voxel_params_t prm{};
  prm.max_points_per_pillar = max_points_per_pillar;
  prm.max_pillars = max_pillars;
  prm.x_step = x_step;
  prm.y_step = y_step;
  prm.x_min = x_min;
  prm.y_min = y_min;
  prm.z_min = z_min;
  prm.x_max = x_max;
  prm.y_max = y_max;
  prm.z_max = z_max;
  prm.canvas_height = canvas_height;
  const char *ord = std::getenv("PP_PILLAR_ORDER");
  prm.order = ord ? std::atoi(ord) : PP_ORDER_SCRAMBLED;
  const int64_t ts[3] = {tensor.shape(0), tensor.shape(1), tensor.shape(2)};
  const int64_t tst[3] = {tensor.strides(0), tensor.strides(1), tensor.strides(2)};
  const int64_t is[2] = {indices.shape(0), indices.shape(1)};
  const int64_t ist[2] = {indices.strides(0), indices.strides(1)};
  int64_t ncell = 0;
  int rc;
  std::string err;
  {
    py::gil_scoped_release nogil;  // the reference holds the GIL for the whole call; no need to
    std::lock_guard<std::mutex> one_call(g_call_mutex);
    pp_ctx_t *c = context_nothrow(&err);
    rc = c ? pp_create_pillars_f64(c, points.data(), points.shape(0), points.strides(0), points.strides(1),
                                   tensor.mutable_data(), ts, tst, indices.mutable_data(), is, ist, &prm, &ncell)
           : PP_ERR_HIP;
    if (rc != PP_OK && err.empty()) err = pp_last_error();
  }
  if (rc != PP_OK) raise_for(rc, "create_pillars", err);
}

void make_ious(in_array a_corners, in_array g_corners, in_array a_centers, in_array g_centers,
               py::object ious_o) {
  need_ndim(a_corners, "a_corners", 3);
  need_ndim(g_corners, "g_corners", 3);
  need_ndim(a_centers, "a_centers", 2);
  need_ndim(g_centers, "g_centers", 2);
  py::array ious = out_array(ious_o, "ious", 2);
  const int64_t A = a_corners.shape(0), G = g_corners.shape(0);
  if (A == 0 || G == 0) return;
  if (a_corners.shape(1) < 4 || a_corners.shape(2) < 2 || g_corners.shape(1) < 4 ||
      g_corners.shape(2) < 2 || a_centers.shape(0) < A || a_centers.shape(1) < 2 ||
      g_centers.shape(0) < G || g_centers.shape(1) < 2 || ious.shape(0) < A || ious.shape(1) < G)
    throw py::index_error("make_ious: index out of bounds for the given array shapes");
  const int64_t ac[3] = {a_corners.strides(0), a_corners.strides(1), a_corners.strides(2)};
  const int64_t gc[3] = {g_corners.strides(0), g_corners.strides(1), g_corners.strides(2)};
  const int64_t an[2] = {a_centers.strides(0), a_centers.strides(1)};
  const int64_t gn[2] = {g_centers.strides(0), g_centers.strides(1)};
  const int64_t io[2] = {ious.strides(0), ious.strides(1)};
  int rc;
  std::string err;
  {
    py::gil_scoped_release nogil;
    std::lock_guard<std::mutex> one_call(g_call_mutex);
    pp_ctx_t *c = context_nothrow(&err);
    rc = c ? pp_make_ious_f64(c, a_corners.data(), A, ac, g_corners.data(), G, gc, a_centers.data(), an,
                              g_centers.data(), gn, ious.mutable_data(), io)
           : PP_ERR_HIP;
    if (rc != PP_OK && err.empty()) err = pp_last_error();
  }
  if (rc != PP_OK) raise_for(rc, "make_ious", err);
}

}  // namespace

PYBIND11_MODULE(pillars, m) {
  m.doc() = "point pillars data prep functions";  // pillars.cpp:431
  m.def("make_ious", &make_ious, "ious");              // pillars.cpp:432
  m.def("create_pillars", &create_pillars, "pillars");  // pillars.cpp:433
}
