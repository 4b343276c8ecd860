// pp_iou.hip -- rotated-box anchor/ground-truth IoU and the fused anchor-target
// assignment as hand-written HIP for gfx950.
//
// Replaces make_ious + iou (/root/reference data/pillars.cpp:400-427, 132-172)
// and, fused on the device-resident path, create_target + make_target
// (utils/box_utils.py:162-232, 70-109) so that the [A,G] IoU matrix is never
// materialised.
//
// Boost.Geometry (bg::intersection / bg::area, pillars.cpp:160,164,165) is an
// absent, unpinned third-party dependency of the reference; its published
// behaviour for convex quads is restated as Sutherland-Hodgman clipping plus a
// shoelace area, executed in f64 with contraction disabled in exactly the
// operation order of oracle/pp_oracle.c, so HIP-vs-oracle parity is bit-exact.

#include "pp_common.h"

#include <algorithm>
#include <climits>
#include <cstring>

namespace pp {

constexpr int kIouThreads = 128;  // 16 LDS vertex slots x 16 B per thread = 32 KB per workgroup
using u64 = unsigned long long;

__device__ __forceinline__ double shoelace_dev(const double *q, int n) {
  double s = 0.0;
  for (int k = 0; k < n; ++k) {
    const int j = (k + 1 == n) ? 0 : k + 1;
    s = s + (q[2 * k] * q[2 * j + 1] - q[2 * j] * q[2 * k + 1]);
  }
  return 0.5 * s;
}

// pillars.cpp:132-172 for one pair.  a: anchor corners, declared counter-
// clockwise; g: ground-truth corners, declared clockwise.  *bad is set when a
// declared-orientation area is negative (the reference's "IOU < 0" exit).
// The two ping-pong vertex lists (at most 8 vertices each) live in LDS, slot-major
// ([16 slots][block threads], 16 B per vertex): lanes of a wave touch consecutive
// 16-byte words, and a dynamically indexed private array would go to scratch
// (global memory) -- the clip is a chain of dependent reads and writes.
struct PolyLds {
  double2 *base;  // &lds[threadIdx.x]
  int stride;     // block threads
  __device__ __forceinline__ double2 &at(int list, int v) const {
    return base[(list * 8 + v) * stride];
  }
};

__device__ double iou_pair_dev(const double a[8], const double g[8], bool *bad, const PolyLds &pl) {
  const double area_a = shoelace_dev(a, 4);
  const double area_g = -shoelace_dev(g, 4);
  if (area_a < 0.0 || area_g < 0.0) {
    *bad = true;
    return -1.0;
  }
  int n = 4, cur = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) pl.at(0, k) = make_double2(a[2 * k], a[2 * k + 1]);
  for (int e = 0; e < 4 && n > 0; ++e) {
    const int ia = (4 - e) & 3, ib = (3 - e) & 3;
    double ax = 0, ay = 0, bx = 0, by = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // static indexing keeps g[] in registers
      ax = (k == ia) ? g[2 * k] : ax;
      ay = (k == ia) ? g[2 * k + 1] : ay;
      bx = (k == ib) ? g[2 * k] : bx;
      by = (k == ib) ? g[2 * k + 1] : by;
    }
    const double ex = bx - ax, ey = by - ay;
    int m = 0;
    const double2 last = pl.at(cur, n - 1);
    double px = last.x, py = last.y;
    double dp = ex * (py - ay) - ey * (px - ax);
    for (int i = 0; i < n; ++i) {
      const double2 c = pl.at(cur, i);
      const double cx = c.x, cy = c.y;
      const double dc = ex * (cy - ay) - ey * (cx - ax);
      if ((dc >= 0.0) != (dp >= 0.0)) {
        const double t = dp / (dp - dc);
        pl.at(cur ^ 1, m) = make_double2(px + t * (cx - px), py + t * (cy - py));
        ++m;
      }
      if (dc >= 0.0) {
        pl.at(cur ^ 1, m) = make_double2(cx, cy);
        ++m;
      }
      px = cx;
      py = cy;
      dp = dc;
    }
    n = m;
    cur ^= 1;
  }
  if (n < 3) return 0.0;
  // shoelace over the clipped ring, same operation order as shoelace_dev
  double s = 0.0;
  for (int k = 0; k < n; ++k) {
    const int j = (k + 1 == n) ? 0 : k + 1;
    const double2 vk = pl.at(cur, k), vj = pl.at(cur, j);
    s = s + (vk.x * vj.y - vj.x * vk.y);
  }
  const double inter = 0.5 * s;
  if (!(inter > 0.0)) return 0.0;
  return inter / (area_a + area_g - inter);
}

// the +-10 cell centre gate of pillars.cpp:418-419
__device__ __forceinline__ bool gate_far(double acx, double acy, double gcx, double gcy) {
  return (fabs(acx - gcx) > 10.0) || (fabs(acy - gcy) > 10.0);
}

// ------------------------------------------------------------------------- //
// make_ious: one lane per (anchor, gt) entry, coalesced f64 stores           //
// ------------------------------------------------------------------------- //
__global__ __launch_bounds__(kIouThreads) void k_make_ious(
    const double *__restrict__ a_corners, const double *__restrict__ a_centers, int acols,
    int64_t A, const double *__restrict__ g_corners, const double *__restrict__ g_centers,
    int gcols, int G, double *__restrict__ ious, int *errflag) {
  __shared__ double2 s_poly[16 * kIouThreads];
  const PolyLds pl{s_poly + threadIdx.x, kIouThreads};
  const int64_t total = A * G;
  for (int64_t e = (int64_t)blockIdx.x * kIouThreads + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * kIouThreads) {
    const int64_t i = e / G;
    const int j = (int)(e - i * G);
    const double acx = a_centers[i * acols], acy = a_centers[i * acols + 1];
    const double gcx = g_centers[(int64_t)j * gcols], gcy = g_centers[(int64_t)j * gcols + 1];
    double v = 0.0;
    if (!gate_far(acx, acy, gcx, gcy)) {
      double a[8], g[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a[k] = a_corners[i * 8 + k];
        g[k] = g_corners[(int64_t)j * 8 + k];
      }
      bool bad = false;
      v = iou_pair_dev(a, g, &bad, pl);
      if (bad) atomicExch(errflag, 1);
    }
    ious[e] = v;
  }
}

// ------------------------------------------------------------------------- //
// fused target assignment                                                    //
// ------------------------------------------------------------------------- //
struct TargetArgs {
  int64_t A;
  int G;
  const double *a_corners, *a_centers, *a_wlh, *a_yaw;  // anchor arrays, or all NULL with ...
  // ... the anchor grid of make_anchor_boxes (box_utils.py:111-159) evaluated on the fly:
  // anchor i = (y*fm_w + x)*per_cell + d; centre ((x+.5)/fm_scale, (y+.5)/fm_scale, z_d)
  int grid, fm_w, per_cell;
  double fm_scale;
  const double *types;  // [per_cell][kTypeCols]: corner offsets x0,y0..x3,y3, w, l, h, yaw, z
  const double *g_corners, *g_centers_img, *g_centers, *g_wlh, *g_yaw;
  const int *g_class;
  double pos_thresh, canvas_height;
  int num_classes;
  // scratch
  u64 *col_max;  // [G] bit pattern of the column maximum IoU (0 = all zero): the tail's scratch when
  int *col_arg;  // [G] first anchor reaching it                 G is beyond its LDS
  int *errflag;
  // per workgroup and ground truth it reaches with IoU > 0: {gt, first anchor of the workgroup's
  // column maximum, maximum lo, hi}; at most G entries per workgroup
  int4 *cand;            // [workgroups * G]
  unsigned *cand_count;  // entries appended
  unsigned *ticket;      // workgroups finished
  // outputs
  float *cls_targets;  // [A][num_classes]
  float *reg_targets;  // [A][9]
};

constexpr int kTypeCols = 13;

struct AnchorId {
  int d;          // anchor type within the cell
  double cx, cy;  // centre
};

__device__ __forceinline__ AnchorId anchor_id(const TargetArgs &t, int64_t i) {
  AnchorId a;
  if (t.grid) {
    const int64_t cell = i / t.per_cell;
    a.d = (int)(i - cell * t.per_cell);
    const int64_t y = cell / t.fm_w, x = cell - y * t.fm_w;
    a.cx = ((double)x + 0.5) / t.fm_scale;  // box_utils.py:137-138, same f64 operations
    a.cy = ((double)y + 0.5) / t.fm_scale;
  } else {
    a.d = 0;
    a.cx = t.a_centers[i * 3];
    a.cy = t.a_centers[i * 3 + 1];
  }
  return a;
}

// corners = per-type rotated offsets + centre: the last addition of Box.bottom_corners
// (boxes.bottom_corners_xy), so the values equal the uploaded arrays' bit for bit
__device__ __forceinline__ void anchor_corners(const TargetArgs &t, int64_t i, double a[8]) {
  if (t.grid) {
    const AnchorId id = anchor_id(t, i);
    const double *ty = t.types + id.d * kTypeCols;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[2 * k] = ty[2 * k] + id.cx;
      a[2 * k + 1] = ty[2 * k + 1] + id.cy;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = t.a_corners[i * 8 + k];
  }
}

// utils/box_utils.py:70-109
__device__ void make_target_dev(const TargetArgs &t, int64_t i, int j, float out[9]) {
  double ax, ay, az, aw, al, ah, at;
  if (t.grid) {
    const AnchorId id = anchor_id(t, i);
    const double *ty = t.types + id.d * kTypeCols;
    ax = id.cx, ay = id.cy, az = ty[12];
    aw = ty[8], al = ty[9], ah = ty[10], at = ty[11];
  } else {
    ax = t.a_centers[i * 3], ay = t.a_centers[i * 3 + 1], az = t.a_centers[i * 3 + 2];
    aw = t.a_wlh[i * 3], al = t.a_wlh[i * 3 + 1], ah = t.a_wlh[i * 3 + 2];
    at = t.a_yaw[i];
  }
  const double gx = t.g_centers[j * 3];
  double gy = t.g_centers[j * 3 + 1];
  const double gz = t.g_centers[j * 3 + 2];
  const double gw = t.g_wlh[j * 3], gl = t.g_wlh[j * 3 + 1], gh = t.g_wlh[j * 3 + 2];
  const double ad = sqrt(aw * aw + al * al);
  double gt = t.g_yaw[j];
  const double pi = 3.141592653589793;  // np.pi
  gy = (t.canvas_height - 1) - gy;       // box_utils.py:83
  const double dx = (gx - ax) / ad;
  const double dy = (gy - ay) / ad;
  const double dz = (gz - az) / ah;
  const double dw = log(gw / aw);
  const double dl = log(gl / al);
  const double dh = log(gh / ah);
  if (gt <= pi && gt >= pi / 2)  // box_utils.py:92-95
    gt -= pi;
  else if (gt >= -pi && gt <= -pi / 2)
    gt += pi;
  const double dt = sin(gt - at);
  const double df = gt - at;
  const double ort =
      ((df <= pi && df >= pi / 2) || (df >= -pi && df <= -pi / 2)) ? 1.0 : 0.0;  // :99-102
  out[0] = 1.0f;
  out[1] = (float)dx;
  out[2] = (float)dy;
  out[3] = (float)dz;
  out[4] = (float)dw;
  out[5] = (float)dl;
  out[6] = (float)dh;
  out[7] = (float)dt;
  out[8] = (float)ort;
}

__device__ __forceinline__ double pair_iou(const TargetArgs &t, int64_t i, int j, bool *bad,
                                           const PolyLds &pl) {
  double a[8], g[8];
  anchor_corners(t, i, a);
#pragma unroll
  for (int k = 0; k < 8; ++k) g[k] = t.g_corners[(int64_t)j * 8 + k];
  return iou_pair_dev(a, g, bad, pl);
}

// ------------------------------------------------------------------------- //
// k_targets: create_target (utils/box_utils.py:162-232) in ONE launch         //
// ------------------------------------------------------------------------- //
// One workgroup per 256 consecutive anchors:
//   gate     every anchor tests the +-10 centre gate against a chunk of 64 ground truths (centres in
//            LDS) and keeps the survivors as a 64-bit mask; only ~0.16 % of the pairs pass;
//   queue    a workgroup prefix sum lines the surviving (anchor, gt) pairs up in LDS, pairs of one
//            anchor in ascending gt order, and they are clipped a window of 512 at a time;
//   clip     EIGHT lanes per pair, one polygon vertex each (clip_group): a Sutherland-Hodgman pass
//            is one step for the whole ring instead of a loop over it -- the serial clip was a
//            10 us dependent chain, half of the old kernel;
//   reduce   row maximum / first argmax per anchor (box_utils.py:193-196); per ground truth the
//            workgroup's column maximum and the first anchor reaching it, reduced in LDS and
//            appended to a list (at most one entry per workgroup and ground truth: the list can
//            never overflow its [workgroups x G] slots, no global atomic on the data path except
//            one counter bump per workgroup and chunk);
//   store    class and regression rows (:211, :219-221; zero rows otherwise) staged in LDS and
//            written as whole 16-byte groups;
//   tail     the LAST workgroup to finish (ticket) reduces the list to the column argmax
//            (np.argmax over the transposed matrix, :199-200) and writes the forced rows
//            (:204-205, :212-213, :223-228).  Two more launches cost more than this tail.
constexpr int kTgtThreads = 256;
constexpr int kTgtWaves = kTgtThreads / 64;
constexpr int kGtChunk = 64;    // ground truths per gate pass: one mask bit each
constexpr int kPairCap = 512;   // pairs per window
constexpr int kGroup = 8;       // lanes per pair: a quad clipped by a quad has at most 8 vertices
constexpr int kPairsPerRound = kTgtThreads / kGroup;
constexpr int kStageCols = 16;  // widest target row staged in LDS
constexpr int kForcedLds = 2048;  // ground truths whose column results the tail keeps in LDS

struct TgtLds {
  double2 gc[kGtChunk];                   // image-space centres of the chunk's ground truths
  double2 gk[kGtChunk][4];                // their corners
  double2 poly[kPairsPerRound][kGroup];   // hand-over of a clip pass's output ring
  double iou[kPairCap];
  u64 cmax[kGtChunk], cseen[kGtChunk];    // column maximum of this workgroup / as of the last window
  int carg[kGtChunk];                     // first anchor reaching it
  unsigned short pair_lane[kPairCap], pair_gt[kPairCap];
  int woff[kTgtWaves];
  int is_last;
  float stage[kTgtThreads * kStageCols];
};
struct TailLds {
  u64 colmax[kForcedLds];
  int colarg[kForcedLds];
  unsigned char cls[kForcedLds];
};
constexpr size_t kTgtLdsBytes = sizeof(TgtLds) > sizeof(TailLds) ? sizeof(TgtLds) : sizeof(TailLds);

__device__ __forceinline__ void iou_wave_sync() {
  // LDS operations of one wave execute in program order; this only stops the compiler from
  // moving them across a cross-lane hand-off
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// shoelace_dev over a ring held one vertex per lane (lanes gbase .. gbase+n-1): the terms are
// formed in parallel, the sum runs in the serial order
__device__ __forceinline__ double group_shoelace(double x, double y, int n, int v, int gbase) {
  const int succ = gbase + ((v + 1 >= n) ? 0 : v + 1);
  const double jx = __shfl(x, succ, 64), jy = __shfl(y, succ, 64);
  const double term = x * jy - jx * y;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < kGroup; ++k) {
    const double tk = __shfl(term, gbase + k, 64);
    if (k < n) s = s + tk;
  }
  return 0.5 * s;
}

// iou_pair_dev with the ring spread over the 8 lanes of a group: lane v holds vertex v.  Every
// vertex goes through the same operations in the same order as in the serial loop (dp of vertex
// i is dc of vertex i-1, the same function of the same operands), and the output ring is laid
// out in the serial order (crossing point before the kept vertex, vertices ascending): same
// bits.  (cx, cy): this lane's anchor corner (v < 4); gk: the ground truth's corners in LDS.
// All 64 lanes must call it together; lanes of idle groups pass zeros.
__device__ __forceinline__ double clip_group(double cx, double cy, const double2 *gk, double2 *poly,
                                             int v, int gbase, bool *wrong) {
  const double area_a = group_shoelace(cx, cy, 4, v, gbase);
  const double2 gv = gk[v & 3];
  const double area_g = -group_shoelace(gv.x, gv.y, 4, v, gbase);
  *wrong = (area_a < 0.0 || area_g < 0.0);
  int n = 4;
  const unsigned below = (1u << v) - 1u;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ia = (4 - e) & 3, ib = (3 - e) & 3;
    const double2 a = gk[ia], b = gk[ib];
    const double ex = b.x - a.x, ey = b.y - a.y;
    const bool act = v < n;
    const double dc = ex * (cy - a.y) - ey * (cx - a.x);
    const int pred = gbase + ((v == 0) ? max(n - 1, 0) : v - 1);
    const double dp = __shfl(dc, pred, 64), px = __shfl(cx, pred, 64), py = __shfl(cy, pred, 64);
    const bool in_c = dc >= 0.0, in_p = dp >= 0.0;
    const bool cross = act && (in_c != in_p), keep = act && in_c;
    const unsigned cb = (unsigned)(__ballot(cross) >> gbase) & 0xFFu;
    const unsigned kb = (unsigned)(__ballot(keep) >> gbase) & 0xFFu;
    int pos = __popc(cb & below) + __popc(kb & below);
    if (cross) {
      const double tt = dp / (dp - dc);
      // (a ninth vertex cannot come from two convex quads; never write past the ring)
      if (pos < kGroup) poly[pos] = make_double2(px + tt * (cx - px), py + tt * (cy - py));
      ++pos;
    }
    if (keep && pos < kGroup) poly[pos] = make_double2(cx, cy);
    n = min(__popc(cb) + __popc(kb), kGroup);
    iou_wave_sync();
    if (v < n) {
      const double2 c = poly[v];
      cx = c.x;
      cy = c.y;
    }
    iou_wave_sync();
  }
  const double inter = group_shoelace(cx, cy, n, v, gbase);
  if (*wrong) return -1.0;
  if (n < 3) return 0.0;
  if (!(inter > 0.0)) return 0.0;
  return inter / (area_a + area_g - inter);
}

__device__ __forceinline__ u64 ld_agent(const u64 *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_agent(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Stores another XCD must be able to order against (the tail overwrites forced rows, and reads
// the workgroups' list entries): write-through (sc1), drained with s_waitcnt vmcnt(0) before the
// workgroup's ticket.  An agent-scope release fence instead writes the XCD's whole L2 back, once
// per wave: measured 39 us of a 55 us kernel.
typedef unsigned v4u __attribute__((__vector_size__(4 * sizeof(unsigned))));
constexpr int kAuxSc1 = 16;

// `total` floats from the LDS stage to dst (16-byte aligned), 16 bytes per store
__device__ __forceinline__ void store_rows(float *dst, const float *stage, int total, int tid) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, total * 4, 0x00020000);
  const int n4 = total >> 2;
  for (int k = tid; k < n4; k += kTgtThreads)
    __builtin_amdgcn_raw_buffer_store_b128(reinterpret_cast<const v4u *>(stage)[k], rs, k * 16, 0, kAuxSc1);
  for (int k = (n4 << 2) + tid; k < total; k += kTgtThreads)
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(stage[k]), rs, k * 4, 0, kAuxSc1);
}
__device__ __forceinline__ void store_f32_sc1(float *p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The last workgroup: column argmax from the workgroups' entries, then the forced rows.  A
// column whose argmax is anchor 0 -- all-zero columns included -- is dropped, exactly like the
// reference's np.nonzero filter (box_utils.py:204-205).  IN_LDS: the column words live in LDS
// (G <= kForcedLds), else in the armed global scratch (0 / INT_MAX), which only this workgroup
// touches: atomics and sc1 loads meet in its XCD's L2.
template <bool IN_LDS>
__device__ void targets_tail(const TargetArgs &t, TailLds &T) {
  const int G = t.G, tid = threadIdx.x;
  const unsigned n = __hip_atomic_load(t.cand_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  auto cmax_at = [&](int j) -> u64 * { return IN_LDS ? &T.colmax[j] : &t.col_max[j]; };
  auto carg_at = [&](int j) -> int * { return IN_LDS ? &T.colarg[j] : &t.col_arg[j]; };
  auto cmax_ld = [&](int j) -> u64 { return IN_LDS ? T.colmax[j] : ld_agent(&t.col_max[j]); };
  auto carg_ld = [&](int j) -> int { return IN_LDS ? T.colarg[j] : ld_agent(&t.col_arg[j]); };
  auto sync = [&]() {
    if (!IN_LDS) __threadfence();
    __syncthreads();
  };
  if (IN_LDS)
    for (int j = tid; j < G; j += kTgtThreads) {
      T.colmax[j] = 0ull;
      T.colarg[j] = INT_MAX;
    }
  __syncthreads();
  auto entry = [&](unsigned e, int &j, int &anchor, u64 &bits) {
    const u64 *q = reinterpret_cast<const u64 *>(t.cand + e);
    const u64 lo = ld_agent(q), hi = ld_agent(q + 1);
    j = (int)(lo & 0xFFFFFFFFull);
    anchor = (int)(lo >> 32);
    bits = hi;
  };
  for (unsigned e = tid; e < n; e += kTgtThreads) {
    int j, anchor;
    u64 bits;
    entry(e, j, anchor, bits);
    atomicMax(cmax_at(j), bits);
  }
  sync();
  for (unsigned e = tid; e < n; e += kTgtThreads) {
    int j, anchor;
    u64 bits;
    entry(e, j, anchor, bits);
    if (bits == cmax_ld(j)) atomicMin(carg_at(j), anchor);
  }
  sync();
  // carg[j] <- the anchor ground truth j forces, 0 for none
  for (int j = tid; j < G; j += kTgtThreads) {
    const int i = (cmax_ld(j) != 0ull) ? carg_ld(j) : 0;
    if (IN_LDS) {
      T.colarg[j] = i;
      T.cls[j] = (unsigned char)t.g_class[j];
    } else {
      __hip_atomic_store(&t.col_arg[j], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  sync();
  if (IN_LDS && t.num_classes <= 64) {
    for (int j = tid; j < G; j += kTgtThreads) {
      const int i = T.colarg[j];
      if (i == 0) continue;
      // class row of anchor i: ones at the classes of ALL ground truths forcing it (every
      // duplicate writes the same full row); regression row: the last ground truth wins
      u64 mask = 0;
      bool later = false;
      for (int j2 = 0; j2 < G; ++j2)
        if (T.colarg[j2] == i) {
          mask |= 1ull << (T.cls[j2] & 63);
          later = later || (j2 > j);
        }
      float *cls = t.cls_targets + (int64_t)i * t.num_classes;
      for (int c = 0; c < t.num_classes; ++c) cls[c] = ((mask >> c) & 1ull) ? 1.0f : 0.0f;
      if (!later) {
        float r[9];
        make_target_dev(t, i, j, r);
        float *reg = t.reg_targets + (int64_t)i * 9;
        for (int d = 0; d < 9; ++d) reg[d] = r[d];
      }
    }
  } else {
    // phase A: clear the class rows of all forced anchors
    for (int j = tid; j < G; j += kTgtThreads) {
      const int i = carg_ld(j);
      if (i != 0) {
        float *cls = t.cls_targets + (int64_t)i * t.num_classes;
        for (int c = 0; c < t.num_classes; ++c) cls[c] = 0.0f;
      }
    }
    __threadfence_block();
    __syncthreads();
    // phase B: set the classes (duplicates of one anchor set several ones, as numpy's fancy
    // assignment does); regression row: the last ground truth wins
    for (int j = tid; j < G; j += kTgtThreads) {
      const int i = carg_ld(j);
      if (i == 0) continue;
      if ((unsigned)t.g_class[j] < (unsigned)t.num_classes)  // a class outside the row is ignored, never written
        t.cls_targets[(int64_t)i * t.num_classes + t.g_class[j]] = 1.0f;
      bool later = false;
      for (int j2 = j + 1; j2 < G; ++j2) later = later || (carg_ld(j2) == i);
      if (!later) {
        float r[9];
        make_target_dev(t, i, j, r);
        float *reg = t.reg_targets + (int64_t)i * 9;
        for (int d = 0; d < 9; ++d) reg[d] = r[d];
      }
    }
  }
  // re-arm the scratch words for the next call on this context
  __syncthreads();
  if (!IN_LDS)
    for (int j = tid; j < G; j += kTgtThreads) {
      t.col_max[j] = 0ull;
      t.col_arg[j] = INT_MAX;
    }
  if (tid == 0) {
    *t.cand_count = 0u;
    *t.ticket = 0u;
  }
}

__global__ __launch_bounds__(kTgtThreads) void k_targets(TargetArgs t) {
  __shared__ __align__(16) unsigned char smem[kTgtLdsBytes];
  TgtLds &S = *reinterpret_cast<TgtLds *>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int v = lane & (kGroup - 1), gbase = lane & ~(kGroup - 1);
  const int64_t i0 = (int64_t)blockIdx.x * kTgtThreads;
  const int64_t i = i0 + tid;
  const bool live = i < t.A;
  double acx = 0, acy = 0;
  if (live) {
    const AnchorId id = anchor_id(t, i);
    acx = id.cx;
    acy = id.cy;
  }
  double best = 0.0;  // np.max over a row that is all zeros is 0, argmax 0
  int best_j = 0;
  bool bad = false;
  for (int j0 = 0; j0 < t.G; j0 += kGtChunk) {
    const int gn = min(kGtChunk, t.G - j0);
    __syncthreads();
    if (tid < gn) {
      S.gc[tid] = make_double2(t.g_centers_img[(int64_t)(j0 + tid) * 3], t.g_centers_img[(int64_t)(j0 + tid) * 3 + 1]);
      S.cmax[tid] = 0ull;
      S.cseen[tid] = 0ull;
      S.carg[tid] = INT_MAX;
    }
    for (int k = tid; k < gn * 4; k += kTgtThreads) {
      const double *gp = t.g_corners + ((int64_t)j0 * 4 + k) * 2;
      S.gk[k >> 2][k & 3] = make_double2(gp[0], gp[1]);
    }
    __syncthreads();
    // gate: the survivors of this chunk as a mask
    u64 mask = 0;
    if (live)
      for (int j = 0; j < gn; ++j)
        mask |= (u64)(gate_far(acx, acy, S.gc[j].x, S.gc[j].y) ? 0 : 1) << j;
    const int cnt = __popcll(mask);
    // exclusive prefix sum of the counts over the workgroup (wave scans + wave totals)
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (lane == 63) S.woff[wv] = inc;
    __syncthreads();
    int wave_base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kTgtWaves; ++w) {
      if (w < wv) wave_base += S.woff[w];
      total += S.woff[w];
    }
    const int my_off = wave_base + inc - cnt;
    if (total == 0) continue;
    for (int wb = 0; wb < total; wb += kPairCap) {
      const int wn = min(kPairCap, total - wb);
      // queue: this window's pairs (ascending gt inside each anchor's slice)
      if (cnt > 0 && my_off < wb + wn && my_off + cnt > wb) {
        u64 m = mask;
        int q = my_off - wb;
        while (m) {
          const int j = __ffsll((long long)m) - 1;
          m &= m - 1;
          if (q >= 0 && q < wn) {
            S.pair_lane[q] = (unsigned short)tid;
            S.pair_gt[q] = (unsigned short)j;
          }
          ++q;
        }
      }
      __syncthreads();
      // clip: 32 pairs per round
      for (int r0 = 0; r0 < wn; r0 += kPairsPerRound) {
        const int q = r0 + (tid >> 3);
        const bool on = q < wn;
        const int pl = on ? S.pair_lane[q] : 0, pg = on ? S.pair_gt[q] : 0;
        double cx = 0, cy = 0;
        if (on && v < 4) {
          if (t.grid) {
            const AnchorId id = anchor_id(t, i0 + pl);
            const double *ty = t.types + id.d * kTypeCols;
            cx = ty[2 * v] + id.cx;
            cy = ty[2 * v + 1] + id.cy;
          } else {
            cx = t.a_corners[(i0 + pl) * 8 + 2 * v];
            cy = t.a_corners[(i0 + pl) * 8 + 2 * v + 1];
          }
        }
        bool wrong;
        const double iou = clip_group(cx, cy, S.gk[pg], S.poly[tid >> 3], v, gbase, &wrong);
        if (on && v == 0) {
          S.iou[q] = iou;
          bad = bad || wrong;
        }
      }
      __syncthreads();
      // rows: this anchor's slice of the window, ascending gt; strict >: first maximum wins
      if (cnt > 0) {
        const int qa = max(my_off, wb), qb = min(my_off + cnt, wb + wn);
        for (int q = qa; q < qb; ++q) {
          const double val = S.iou[q - wb];
          if (val > best) {
            best = val;
            best_j = j0 + S.pair_gt[q - wb];
          }
        }
      }
      // columns: maximum, then the first anchor that reaches it (anchors ascend with the windows:
      // a column whose maximum grew in this window forgets the earlier windows' anchor)
      for (int q = tid; q < wn; q += kTgtThreads) {
        const double val = S.iou[q];
        if (val > 0.0) atomicMax(&S.cmax[S.pair_gt[q]], (u64)__double_as_longlong(val));
      }
      __syncthreads();
      if (tid < gn && S.cmax[tid] != S.cseen[tid]) {
        S.cseen[tid] = S.cmax[tid];
        S.carg[tid] = INT_MAX;
      }
      __syncthreads();
      for (int q = tid; q < wn; q += kTgtThreads) {
        const double val = S.iou[q];
        if (val > 0.0 && (u64)__double_as_longlong(val) == S.cmax[S.pair_gt[q]])
          atomicMin(&S.carg[S.pair_gt[q]], (int)(i0 + S.pair_lane[q]));
      }
      __syncthreads();
    }
    // this workgroup's columns of the chunk -> the list
    if (wv == 0) {
      const bool touched = lane < gn && S.cmax[lane] != 0ull;
      const u64 tb = __ballot(touched);
      if (tb) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(t.cand_count, (unsigned)__popcll(tb));
        base = (unsigned)__shfl((int)base, 0, 64);
        if (touched) {
          const unsigned pos = base + (unsigned)__popcll(tb & ((1ull << lane) - 1ull));
          const u64 bits = S.cmax[lane];
          u64 *q = reinterpret_cast<u64 *>(t.cand + pos);  // {gt, anchor}, bits
          __hip_atomic_store(q, (u64)(unsigned)(j0 + lane) | ((u64)(unsigned)S.carg[lane] << 32), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(q + 1, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
  if (bad) atomicExch(t.errflag, 1);
  // rows: positives of both targets, zero rows otherwise
  const int nc = t.num_classes;
  const int nrows = (int)min((int64_t)kTgtThreads, t.A - i0);
  const bool pos = live && best > t.pos_thresh;  // box_utils.py:195 (strict >)
  const int cj = pos ? t.g_class[best_j] : -1;
  float r[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (pos) make_target_dev(t, i, best_j, r);
  float *cls_dst = t.cls_targets + i0 * nc, *reg_dst = t.reg_targets + i0 * 9;
  __syncthreads();
  if (nc <= kStageCols && ((uintptr_t)t.cls_targets & 15) == 0) {
    if (live)
      for (int c = 0; c < nc; ++c) S.stage[tid * nc + c] = (c == cj) ? 1.0f : 0.0f;
    __syncthreads();
    store_rows(cls_dst, S.stage, nrows * nc, tid);
    __syncthreads();
  } else if (live) {
    for (int c = 0; c < nc; ++c) store_f32_sc1(&cls_dst[(int64_t)tid * nc + c], (c == cj) ? 1.0f : 0.0f);
  }
  if (((uintptr_t)t.reg_targets & 15) == 0) {
    if (live) {
#pragma unroll
      for (int d = 0; d < 9; ++d) S.stage[tid * 9 + d] = r[d];
    }
    __syncthreads();
    store_rows(reg_dst, S.stage, nrows * 9, tid);
  } else if (live) {
#pragma unroll
    for (int d = 0; d < 9; ++d) store_f32_sc1(&reg_dst[(int64_t)tid * 9 + d], r[d]);
  }
  if (t.G == 0) return;
  // The last workgroup to get here finishes the job.  Every store above that the tail depends on
  // is write-through; drained per wave, then one agent-scope add per workgroup: the workgroup
  // whose add comes last reads the others' entries with sc1 loads and may overwrite their rows.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) S.is_last = (atomicAdd(t.ticket, 1u) == gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (!S.is_last) return;
  if (t.G <= kForcedLds)
    targets_tail<true>(t, *reinterpret_cast<TailLds *>(smem));
  else
    targets_tail<false>(t, *reinterpret_cast<TailLds *>(smem));
}

__global__ void k_targets_init(u64 *col_max, int *col_arg, int G, int *errflag,
                               unsigned *cand_count, unsigned *ticket) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < G) {
    col_max[j] = 0ull;
    col_arg[j] = INT_MAX;
  }
  if (j == 0) {
    *errflag = 0;
    *cand_count = 0u;
    *ticket = 0u;
  }
}

namespace {
struct DeviceGuard2 {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard2(int dev) {
    if (hipGetDevice(&prev) == hipSuccess) {
      ok = true;
      if (prev != dev) (void)hipSetDevice(dev);
    } else {
      (void)hipGetLastError();
    }
  }
  ~DeviceGuard2() {
    if (ok) (void)hipSetDevice(prev);
  }
};
}  // namespace

}  // namespace pp

using namespace pp;

extern "C" int pp_make_ious_dev(pp_ctx_t *ctx, void *stream_, const double *a_corners_dev,
                                const double *a_centers_dev, int64_t a_center_cols, int64_t A,
                                const double *g_corners_dev, const double *g_centers_dev,
                                int64_t g_center_cols, int64_t G, double *ious_dev) {
  if (!ctx) {
    set_error("ctx is NULL");
    return PP_ERR_VALUE;
  }
  if (A < 0 || G < 0 || G > INT_MAX / 16 || A > (1ll << 40) || a_center_cols < 2 ||
      g_center_cols < 2) {
    set_error("pp_make_ious_dev: bad sizes (A=%lld G=%lld)", (long long)A, (long long)G);
    return PP_ERR_VALUE;
  }
  if (A == 0 || G == 0) return PP_OK;
  if (!a_corners_dev || !a_centers_dev || !g_corners_dev || !g_centers_dev || !ious_dev) {
    set_error("pp_make_ious_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DeviceGuard2 guard(ctx->device);
  int rc = ctx->iou_ws.ensure(4096);
  if (rc) return rc;
  int *errflag = static_cast<int *>(ctx->iou_ws.ptr);
  PP_HIP_TRY(hipMemsetAsync(errflag, 0, 4, stream));
  const int64_t total = A * G;
  const unsigned blocks = (unsigned)std::min<int64_t>((total + kIouThreads - 1) / kIouThreads, 1 << 20);
  hipLaunchKernelGGL(k_make_ious, dim3(blocks), dim3(kIouThreads), 0, stream, a_corners_dev,
                     a_centers_dev, (int)a_center_cols, A, g_corners_dev, g_centers_dev,
                     (int)g_center_cols, (int)G, ious_dev, errflag);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}

// error flag of the last IoU / target launch on this context (synchronises)
extern "C" int pp_iou_check(pp_ctx_t *ctx, void *stream_) {
  if (!ctx || !ctx->iou_ws.ptr) return PP_OK;
  DeviceGuard2 guard(ctx->device);
  int flag = 0;
  PP_HIP_TRY(hipMemcpyAsync(&flag, ctx->iou_ws.ptr, 4, hipMemcpyDeviceToHost,
                            static_cast<hipStream_t>(stream_)));
  PP_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream_)));
  if (flag) {
    // the flag is sticky across launches until it has been reported once
    PP_HIP_TRY(hipMemsetAsync(ctx->iou_ws.ptr, 0, 4, static_cast<hipStream_t>(stream_)));
    set_error("IOU < 0: a box has the wrong corner winding (pillars.cpp:166-169)");
    return PP_ERR_WINDING;
  }
  return PP_OK;
}

extern "C" int pp_make_ious_f64(pp_ctx_t *ctx, const void *a_corners, int64_t A,
                                const int64_t ac[3], const void *g_corners, int64_t G,
                                const int64_t gc[3], const void *a_centers, const int64_t an[2],
                                const void *g_centers, const int64_t gn[2], void *ious,
                                const int64_t io[2]) {
  if (!ctx || !ac || !gc || !an || !gn || !io) {
    set_error("pp_make_ious_f64: NULL argument");
    return PP_ERR_VALUE;
  }
  if (A < 0 || G < 0) {
    set_error("negative size");
    return PP_ERR_VALUE;
  }
  if (A == 0 || G == 0) return PP_OK;
  if (!a_corners || !g_corners || !a_centers || !g_centers || !ious) {
    set_error("pp_make_ious_f64: NULL array");
    return PP_ERR_VALUE;
  }
  DeviceGuard2 guard(ctx->device);
  hipStream_t stream = nullptr;
  // pinned staging: anchors [A][8] + [A][2], gts [G][8] + [G][2]
  const size_t in_bytes = ((size_t)A * 10 + (size_t)G * 10) * 8;
  int rc = ctx->pin_in.ensure(in_bytes);
  if (rc) return rc;
  rc = ctx->stage_in.ensure(in_bytes);
  if (rc) return rc;
  rc = ctx->stage_out2.ensure((size_t)A * G * 8);
  if (rc) return rc;
  rc = ctx->pin_out.ensure((size_t)A * G * 8);
  if (rc) return rc;
  double *h = static_cast<double *>(ctx->pin_in.ptr);
  double *h_ac = h, *h_an = h_ac + A * 8, *h_gc = h_an + A * 2, *h_gn = h_gc + G * 8;
  auto rd = [](const void *base, int64_t off) {
    double v;
    std::memcpy(&v, static_cast<const char *>(base) + off, 8);
    return v;
  };
  for (int64_t i = 0; i < A; ++i) {
    for (int k = 0; k < 4; ++k)
      for (int c = 0; c < 2; ++c) h_ac[i * 8 + k * 2 + c] = rd(a_corners, i * ac[0] + k * ac[1] + c * ac[2]);
    h_an[i * 2] = rd(a_centers, i * an[0]);
    h_an[i * 2 + 1] = rd(a_centers, i * an[0] + an[1]);
  }
  for (int64_t j = 0; j < G; ++j) {
    for (int k = 0; k < 4; ++k)
      for (int c = 0; c < 2; ++c) h_gc[j * 8 + k * 2 + c] = rd(g_corners, j * gc[0] + k * gc[1] + c * gc[2]);
    h_gn[j * 2] = rd(g_centers, j * gn[0]);
    h_gn[j * 2 + 1] = rd(g_centers, j * gn[0] + gn[1]);
  }
  double *d = static_cast<double *>(ctx->stage_in.ptr);
  PP_HIP_TRY(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, stream));
  rc = pp_make_ious_dev(ctx, stream, d, d + A * 8, 2, A, d + A * 10, d + A * 10 + G * 8, 2, G,
                        static_cast<double *>(ctx->stage_out2.ptr));
  if (rc) return rc;
  PP_HIP_TRY(hipMemcpyAsync(ctx->pin_out.ptr, ctx->stage_out2.ptr, (size_t)A * G * 8,
                            hipMemcpyDeviceToHost, stream));
  PP_HIP_TRY(hipStreamSynchronize(stream));
  // every entry is written (pillars.cpp:421,424)
  const double *src = static_cast<const double *>(ctx->pin_out.ptr);
  char *dst = static_cast<char *>(ious);
  if (io[1] == 8 && io[0] == G * 8) {
    std::memcpy(dst, src, (size_t)A * G * 8);
  } else {
    for (int64_t i = 0; i < A; ++i)
      for (int64_t j = 0; j < G; ++j) std::memcpy(dst + i * io[0] + j * io[1], &src[i * G + j], 8);
  }
  return pp_iou_check(ctx, stream);
}

struct AnchorSource {
  const double *corners = nullptr, *centers = nullptr, *wlh = nullptr, *yaw = nullptr;
  int grid = 0, fm_w = 0, per_cell = 0;
  double fm_scale = 1.0;
  const double *types = nullptr;
};

static int assign_targets_impl(pp_ctx_t *ctx, void *stream_, int64_t A, const AnchorSource &an,
                               int64_t G, const double *g_corners, const double *g_centers_img,
                               const double *g_centers, const double *g_wlh, const double *g_yaw,
                               const int32_t *g_class, const pp_target_params_t *prm,
                               float *cls_targets, float *reg_targets) {
  if (!ctx || !prm) {
    set_error("pp_assign_targets*_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  if (A < 1 || A > INT_MAX / 2 || G < 0 || G > 65535 || prm->num_classes < 1 ||
      prm->num_classes > 1024) {
    set_error("pp_assign_targets*_dev: bad sizes (A=%lld G=%lld classes=%d)", (long long)A,
              (long long)G, prm->num_classes);
    return PP_ERR_VALUE;
  }
  if (!cls_targets || !reg_targets ||
      (G > 0 && (!g_corners || !g_centers_img || !g_centers || !g_wlh || !g_yaw || !g_class))) {
    set_error("pp_assign_targets*_dev: NULL array");
    return PP_ERR_VALUE;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DeviceGuard2 guard(ctx->device);
  // scratch: [0,4096) flags/counters | col_max[Gcap] | col_arg[Gcap] | cand[workgroups * Gcap]
  const size_t gcap = (size_t)std::max<int64_t>(G, 1);
  const size_t nwg = (size_t)((A + kTgtThreads - 1) / kTgtThreads);
  const size_t off_cmax = 4096, off_carg = off_cmax + gcap * 8;
  const size_t off_cand = (off_carg + gcap * 4 + 255) / 256 * 256;
  const size_t need = off_cand + nwg * gcap * 16;
  bool grew = false;
  int rc = ctx->iou_ws.ensure(need, &grew);
  if (rc) return rc;
  char *ws = static_cast<char *>(ctx->iou_ws.ptr);
  TargetArgs t;
  t.A = A;
  t.G = (int)G;
  t.a_corners = an.corners;
  t.a_centers = an.centers;
  t.a_wlh = an.wlh;
  t.a_yaw = an.yaw;
  t.grid = an.grid;
  t.fm_w = an.fm_w;
  t.per_cell = an.per_cell;
  t.fm_scale = an.fm_scale;
  t.types = an.types;
  t.g_corners = g_corners;
  t.g_centers_img = g_centers_img;
  t.g_centers = g_centers;
  t.g_wlh = g_wlh;
  t.g_yaw = g_yaw;
  t.g_class = g_class;
  t.pos_thresh = prm->pos_thresh;
  t.canvas_height = prm->canvas_height;
  t.num_classes = prm->num_classes;
  t.errflag = reinterpret_cast<int *>(ws);
  t.cand_count = reinterpret_cast<unsigned *>(ws + 64);
  t.ticket = reinterpret_cast<unsigned *>(ws + 128);
  t.col_max = reinterpret_cast<u64 *>(ws + off_cmax);
  t.col_arg = reinterpret_cast<int *>(ws + off_carg);
  t.cand = reinterpret_cast<int4 *>(ws + off_cand);
  t.cls_targets = cls_targets;
  t.reg_targets = reg_targets;
  // The scratch words are re-armed by the kernel's tail at the end of every call; only a
  // fresh / regrown / re-shaped workspace needs the init.
  const unsigned long long key = ((unsigned long long)A << 20) ^ (unsigned long long)gcap;
  if (grew || ctx->tgt_key != key) {
    const unsigned gb = (unsigned)((gcap + 255) / 256);
    hipLaunchKernelGGL(k_targets_init, dim3(gb), dim3(256), 0, stream, t.col_max, t.col_arg,
                       (int)G, t.errflag, t.cand_count, t.ticket);
    ctx->tgt_key = key;
  }
  hipLaunchKernelGGL(k_targets, dim3((unsigned)nwg), dim3(kTgtThreads), 0, stream, t);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}

extern "C" int pp_assign_targets_dev(pp_ctx_t *ctx, void *stream_, int64_t A,
                                     const double *a_corners, const double *a_centers,
                                     const double *a_wlh, const double *a_yaw, int64_t G,
                                     const double *g_corners, const double *g_centers_img,
                                     const double *g_centers, const double *g_wlh,
                                     const double *g_yaw, const int32_t *g_class,
                                     const pp_target_params_t *prm, float *cls_targets,
                                     float *reg_targets) {
  if (!a_corners || !a_centers || !a_wlh || !a_yaw) {
    set_error("pp_assign_targets_dev: NULL anchor array");
    return PP_ERR_VALUE;
  }
  AnchorSource an;
  an.corners = a_corners;
  an.centers = a_centers;
  an.wlh = a_wlh;
  an.yaw = a_yaw;
  return assign_targets_impl(ctx, stream_, A, an, G, g_corners, g_centers_img, g_centers, g_wlh,
                             g_yaw, g_class, prm, cls_targets, reg_targets);
}

extern "C" int pp_assign_targets_grid_dev(pp_ctx_t *ctx, void *stream_, int fm_height, int fm_width,
                                          double fm_scale, int per_cell,
                                          const double *anchor_types_dev, int64_t G,
                                          const double *g_corners, const double *g_centers_img,
                                          const double *g_centers, const double *g_wlh,
                                          const double *g_yaw, const int32_t *g_class,
                                          const pp_target_params_t *prm, float *cls_targets,
                                          float *reg_targets) {
  if (fm_height < 1 || fm_width < 1 || per_cell < 1 || per_cell > 1024 || !(fm_scale > 0.0) ||
      !anchor_types_dev || (int64_t)fm_height * fm_width * per_cell > INT_MAX / 2) {
    set_error("pp_assign_targets_grid_dev: bad anchor grid (%dx%d, %d per cell, scale %g)",
              fm_height, fm_width, per_cell, fm_scale);
    return PP_ERR_VALUE;
  }
  AnchorSource an;
  an.grid = 1;
  an.fm_w = fm_width;
  an.per_cell = per_cell;
  an.fm_scale = fm_scale;
  an.types = anchor_types_dev;
  return assign_targets_impl(ctx, stream_, (int64_t)fm_height * fm_width * per_cell, an, G, g_corners,
                             g_centers_img, g_centers, g_wlh, g_yaw, g_class, prm, cls_targets,
                             reg_targets);
}
