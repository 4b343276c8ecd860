// pp_common.h -- shared host-side plumbing of libpp_hip.so (context, errors,
// workspace).  gfx950 / ROCm only; there is deliberately no CPU path here.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "pp_hip.h"

namespace pp {

void set_error(const char *fmt, ...);

#define PP_HIP_TRY(expr)                                                        \
  do {                                                                          \
    hipError_t e__ = (expr);                                                    \
    if (e__ != hipSuccess) {                                                    \
      pp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),     \
                    __FILE__, __LINE__);                                        \
      return PP_ERR_HIP;                                                        \
    }                                                                           \
  } while (0)

// A grow-only device buffer.
struct DevBuf {
  void *ptr = nullptr;
  size_t bytes = 0;
  int ensure(size_t need, bool *grew = nullptr);
  void release();
};

// A grow-only pinned host buffer (staging for the host drop-in entry points).
struct PinBuf {
  void *ptr = nullptr;
  size_t bytes = 0;
  int ensure(size_t need);
  void release();
};

// A small persistent pool of host threads for the host drop-in entry points (pp_create_pillars_f64, pp_make_ious_f64):
// the gathers from and the scatters into the caller's NumPy arrays are chains of cache misses in arrays no CPU cache
// holds (an 86 MB tensor, a 40 MB matrix) -- disjoint rows, so they split across threads.  The reference's caller holds
// the GIL for the whole call (pillars.cpp:429-435); the module releases it, so the threads are free to run.  Workers
// spin briefly between the jobs of one call and sleep on a condition variable between calls.  Created lazily, per
// context (a forked DataLoader worker creates its own context, hence its own pool); PP_HOST_THREADS sets the size
// (default min(8, hardware threads); 1: everything on the calling thread).
class HostPool {
 public:
  explicit HostPool(int threads);
  ~HostPool();
  int size() const { return n_; }  // parts per job, the calling thread included
  // fn(part, parts) for part in [0, parts): part 0 on the calling thread; returns when every part is done
  void run(const std::function<void(int, int)> &fn);
  // the same on the workers only (parts = size() - 1, or inline at wait() when there is no worker): the caller goes on
  // (HIP calls, a stream synchronize) and joins with wait()
  void start(const std::function<void(int, int)> &fn);
  void wait();

 private:
  struct Impl;
  Impl *impl_;
  int n_;
};

// Geometry of the implied cell grid, derived from the create_pillars scalars.
struct GridGeom {
  double x_step, y_step, x_min, y_min, z_min, x_max, y_max, z_max, canvas_height;
  int nx, ny;         // columns / rows of the cell grid
  int ncells;         // nx * ny
  int tile_shift;     // a tile = 1 << tile_shift consecutive slots (64..1024)
  int ntiles;         // ceil(ncells / tile slots) <= 4096
  int tile_bits;      // bits of a tile id
  int order;          // PP_ORDER_*
  unsigned long long mult, mult_inv;  // scrambled order: slot = cell*mult % ncells
  unsigned long long barrett;         // floor(2^64 / ncells), for the device-side modulo
};

// step_mode: tile sizes for k_step's 4-wave tile role (half as many, fatter tiles: kTargetTiles / 2)
int make_grid(const pp_voxel_params_t *prm, GridGeom *g, int step_mode = 0);

#ifdef __HIPCC__
// Inclusive prefix sum over the wave's 64 lanes on the DPP path (row shifts inside the rows of 16,
// then the two row broadcasts of gfx9): 6 VALU adds with DPP operands instead of 6 ds_bpermute
// round trips through the LDS pipeline.  ALL 64 lanes must be active at the call (a DPP read of an
// inactive lane leaves the destination unchanged, which would cut the carry chain).
__device__ __forceinline__ unsigned wave_scan_u32(unsigned v) {
#define PP_DPP_ADD(ctrl, rmask)                                                              \
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xF, false)
  PP_DPP_ADD(0x111, 0xF);  // row_shr:1
  PP_DPP_ADD(0x112, 0xF);  // row_shr:2
  PP_DPP_ADD(0x114, 0xF);  // row_shr:4
  PP_DPP_ADD(0x118, 0xF);  // row_shr:8
  PP_DPP_ADD(0x142, 0xA);  // row_bcast:15 -> rows 1 and 3
  PP_DPP_ADD(0x143, 0xC);  // row_bcast:31 -> rows 2 and 3
#undef PP_DPP_ADD
  return v;
}
// two independent 32-bit counts packed in one word (no carry between the halves)
__device__ __forceinline__ unsigned long long wave_scan_2x32(unsigned long long v) {
  const unsigned lo = wave_scan_u32((unsigned)(v & 0xFFFFFFFFull)), hi = wave_scan_u32((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// OR of a word over the wave's 64 lanes, the same way; the result is wave-uniform (an SGPR)
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
#define PP_DPP_OR(ctrl, rmask) v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xF, false)
  PP_DPP_OR(0x111, 0xF);
  PP_DPP_OR(0x112, 0xF);
  PP_DPP_OR(0x114, 0xF);
  PP_DPP_OR(0x118, 0xF);
  PP_DPP_OR(0x142, 0xA);
  PP_DPP_OR(0x143, 0xC);
#undef PP_DPP_OR
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// minimum / maximum of a double over the wave's 64 lanes on the same DPP steps (a lane without a
// source keeps its own value); the result is read from lane 63 and is wave-uniform
template <bool MAX>
__device__ __forceinline__ double wave_minmax_f64(double v) {
#define PP_DPP_MM(ctrl, rmask)                                                                         \
  do {                                                                                                 \
    const int lo_ = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), ctrl, rmask, 0xF, false); \
    const int hi_ = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), ctrl, rmask, 0xF, false); \
    const double o_ = __hiloint2double(hi_, lo_);                                                       \
    v = MAX ? fmax(v, o_) : fmin(v, o_);                                                                \
  } while (0)
  PP_DPP_MM(0x111, 0xF);
  PP_DPP_MM(0x112, 0xF);
  PP_DPP_MM(0x114, 0xF);
  PP_DPP_MM(0x118, 0xF);
  PP_DPP_MM(0x142, 0xA);
  PP_DPP_MM(0x143, 0xC);
#undef PP_DPP_MM
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
#endif  // __HIPCC__

}  // namespace pp

// A batch on its way through k_step's four roles (pp_voxelize_step_dev): split, tile, order, emit.
struct pp_step_batch {
  bool valid = false;
  int slot = 0;      // workspace slot (1..4)
  int batch = 0, maxn = 0;
  int n_points[PP_MAX_BATCH] = {0};
  int64_t points_stride = 0;
  pp_voxel_params_t prm = {};
};

struct pp_ctx {
  int device = 0;
  // voxelizer scratch, laid out by VoxLayout (pp_voxelize.hip).  Slot 0 serves the plain calls; the
  // batches of the software-pipelined mode (pp_voxelize_step_dev: four of them are in flight, one
  // per stage role of k_step) rotate through slots 1..4.
  static constexpr int kVoxSlots = 5;
  pp::DevBuf vox_ws[kVoxSlots];
  unsigned long long vox_layout_key[kVoxSlots][6] = {};
  pp_step_batch step_batch[3];   // [0]: split done, waits for its tile role; [1]: tiled, waits for the order role;
                                 // [2]: descriptors in pillar order, waits for its emit role
  int step_next_slot = 1;
  hipStream_t step_stream = nullptr;  // the stream the batches in flight were submitted on
  bool sort_lds_armed = false;     // k_sort_runs' dynamic-LDS attribute set on this context's device
  int force_tile_waves = 0;  // development knob: PP_TILE_WAVES in the environment
  size_t dbg_stamps_off = 0, dbg_stamps_bytes = 0;  // PP_STAMPS builds (tools/lab)
  // host drop-in staging
  pp::DevBuf stage_in, stage_out, stage_out2;
  pp::PinBuf pin_in, pin_out, pin_meta;
  pp::HostPool *pool = nullptr;       // host threads of the drop-in entry points (lazily created)
  // pp_make_ious_f64: the anchors of the previous call stay on the device (and, bit for bit, in a pinned mirror): the
  // reference hands the SAME anchor arrays in with every sample (utils/box_utils.py:181-183), and 11 MB of them per call
  // was two thirds of the call.  The gather compares while it copies; only a changed set is uploaded again.
  pp::PinBuf anchors_pin;
  pp::DevBuf anchors_dev;
  int64_t anchors_A = -1;              // rows the mirror holds (-1: nothing)
  hipEvent_t chunk_ev[12] = {};        // one per chunk of the features' device-to-host copy (lazily created)
  int dropin_last_n = 0;               // pp_create_pillars_f64: the previous call's point count and how many points
  int64_t dropin_last_end = 0;         // it emitted -- sizes the feature copy that is sent ahead of the descriptors
  // IoU / target scratch
  pp::DevBuf iou_ws;
  // the layout the target scratch was last armed for: {A, gcap, batch, units, form, splits, cand_per_wg, off_best,
  // ticket groups}, compared field by field (all zero: not armed)
  unsigned long long tgt_key[9] = {};
  pp::DevBuf decode_ws;            // post-processing: sort keys + rocPRIM temporary storage
  pp::DevBuf pfn_ws;               // training feature net: per-workgroup partial sums
  // emit-kernel timing ring (bench.py)
  std::vector<hipEvent_t> ev_start[3], ev_stop[3];  // [PP_KERNEL_*][slot]
  int ev_slots = 0;
  int ev_next = 0;
  int ev_count = 0;
  int ev_columns = 7;    // bit k: column PP_KERNEL_k was recorded (k_step launches record the EMIT column only)
};

namespace pp {
HostPool *host_pool(pp_ctx *ctx);  // the context's pool, created on first use
}
