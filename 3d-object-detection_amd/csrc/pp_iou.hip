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
  u64 *col_max;  // [G] bit pattern of the column maximum IoU (0 = all zero)
  int *col_arg;  // [G] first anchor reaching the column maximum
  int *errflag;
  // (anchor, gt, IoU bits) of every pair with IoU > 0, appended by k_targets_rows
  int4 *cand;       // [cand_cap] {anchor, gt, iou lo, iou hi}
  int cand_cap;
  unsigned *cand_count;  // entries appended (may exceed cand_cap: then k_targets_cols re-scans)
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

// T1: one lane per anchor.  Row maximum / first argmax over the ground truths
// (box_utils.py:193-196), positive rows of both targets (:211, :219-221),
// zero rows otherwise, the column maxima via 64-bit atomicMax on the f64 bit
// pattern (IoU >= 0, so the patterns order like the values), and the list of
// pairs with IoU > 0 for the column-argmax pass.
//
// Only ~0.16 % of the (anchor, gt) pairs pass the centre gate, but an anchor near
// two or three boxes would clip them one after the other while the rest of its
// wave idles.  The workgroup therefore queues its gated pairs in LDS (gate pass:
// count, prefix sum, fill -- pairs of one anchor stay in ascending gt order) and
// clips them one pair per lane, all lanes busy; each anchor then reduces its own
// slice of the results.  A workgroup with more gated pairs than the queue holds
// falls back to the serial loop.
constexpr int kPairCap = 512;
constexpr int kGtChunk = 256;  // ground-truth centres staged in LDS per pass (uniform global loads in
                               // the gate loop are a 500-cycle round trip each: the loop is load-latency-bound)

__global__ __launch_bounds__(kIouThreads) void k_targets_rows(TargetArgs t) {
  // LDS budget decides how many workgroups are resident (the grid should fit in one
  // round): vertex lists for ONE clipping wave (16 KB), centres, queue, results = 26 KB
  __shared__ double2 s_poly[16 * 64];
  __shared__ double2 s_gc[kGtChunk];
  __shared__ double s_iou[kPairCap];
  __shared__ unsigned short s_pair_lane[kPairCap], s_pair_gt[kPairCap];
  __shared__ int s_off[kIouThreads / 64];
  __shared__ u64 s_cmax[kGtChunk];        // column maxima of this workgroup
  __shared__ unsigned s_npos[2], s_base;  // positives of this workgroup / their first list slot
  const PolyLds pl{s_poly + (threadIdx.x & 63), 64};
  const int tid = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * kIouThreads + tid;
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
  auto row_max = [&](int j, double v) {
    if (v > best) {  // strict: first maximum wins, like np.argmax
      best = v;
      best_j = j;
    }
  };
  auto consume = [&](int j, double v) {  // overflow path only: one global atomic per pair
    row_max(j, v);
    if (v > 0.0) {
      const u64 bits = (u64)__double_as_longlong(v);
      atomicMax(&t.col_max[j], bits);
      const unsigned pos = atomicAdd(t.cand_count, 1u);
      if (pos < (unsigned)t.cand_cap)
        t.cand[pos] = make_int4((int)i, j, (int)(bits & 0xFFFFFFFFull), (int)(bits >> 32));
    }
  };
  for (int j0 = 0; j0 < t.G; j0 += kGtChunk) {
    const int gn = min(kGtChunk, t.G - j0);
    __syncthreads();
    for (int j = tid; j < gn; j += kIouThreads) {
      s_gc[j] = make_double2(t.g_centers_img[(int64_t)(j0 + j) * 3], t.g_centers_img[(int64_t)(j0 + j) * 3 + 1]);
      s_cmax[j] = 0;
    }
    if (tid < 2) s_npos[tid] = 0;
    __syncthreads();
    // gate pass 1: count
    int cnt = 0;
    if (live)
      for (int j = 0; j < gn; ++j) cnt += gate_far(acx, acy, s_gc[j].x, s_gc[j].y) ? 0 : 1;
    // exclusive prefix sum of the counts over the workgroup (wave scans + wave totals)
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d, 64);
      if ((tid & 63) >= d) inc += o;
    }
    if ((tid & 63) == 63) s_off[tid >> 6] = inc;
    __syncthreads();
    int wave_base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kIouThreads / 64; ++w) {
      if (w < (tid >> 6)) wave_base += s_off[w];
      total += s_off[w];
    }
    const int my_off = wave_base + inc - cnt;
    if (total == 0) continue;
    if (total <= kPairCap) {
      // gate pass 2: fill the queue (ascending gt inside each anchor's slice)
      if (live) {
        int q = my_off;
        for (int j = 0; j < gn; ++j)
          if (!gate_far(acx, acy, s_gc[j].x, s_gc[j].y)) {
            s_pair_lane[q] = (unsigned short)tid;
            s_pair_gt[q] = (unsigned short)j;
            ++q;
          }
      }
      __syncthreads();
      const int64_t i0 = (int64_t)blockIdx.x * kIouThreads;
      if (tid < 64)  // wave 0 clips, one pair per lane
        for (int q = tid; q < total; q += 64)
          s_iou[q] = pair_iou(t, i0 + s_pair_lane[q], j0 + s_pair_gt[q], &bad, pl);
      __syncthreads();
      if (live)
        for (int q = my_off; q < my_off + cnt; ++q) row_max(j0 + s_pair_gt[q], s_iou[q]);
      // Column maxima and the list of positive pairs: every pair with IoU > 0 needs an
      // atomicMax on its column and a list slot.  Done per pair on global memory that is
      // tens of thousands of atomics on G + 1 addresses, which the L2 serialises (it was
      // 3/4 of this kernel); reduce in LDS and issue one global atomic per (workgroup,
      // column) and one per workgroup for the list.
      for (int q = tid; q < total; q += kIouThreads) {
        const double v = s_iou[q];
        if (v > 0.0) {
          atomicMax(&s_cmax[s_pair_gt[q]], (u64)__double_as_longlong(v));
          atomicAdd(&s_npos[0], 1u);
        }
      }
      __syncthreads();
      if (tid == 0 && s_npos[0]) s_base = atomicAdd(t.cand_count, s_npos[0]);
      for (int j = tid; j < gn; j += kIouThreads)
        if (s_cmax[j]) atomicMax(&t.col_max[j0 + j], s_cmax[j]);
      __syncthreads();
      if (s_npos[0]) {
        const unsigned base = s_base;
        for (int q = tid; q < total; q += kIouThreads) {
          const double v = s_iou[q];
          if (v > 0.0) {
            const unsigned pos = base + atomicAdd(&s_npos[1], 1u);
            const u64 bits = (u64)__double_as_longlong(v);
            if (pos < (unsigned)t.cand_cap)
              t.cand[pos] = make_int4((int)(i0 + s_pair_lane[q]), j0 + s_pair_gt[q],
                                      (int)(bits & 0xFFFFFFFFull), (int)(bits >> 32));
          }
        }
      }
    } else {
      // queue overflow (never on real scenes): the two waves take turns on the vertex lists
      for (int turn = 0; turn < kIouThreads / 64; ++turn) {
        if (live && (tid >> 6) == turn)
          for (int j = 0; j < gn; ++j) {
            if (gate_far(acx, acy, s_gc[j].x, s_gc[j].y)) continue;
            consume(j0 + j, pair_iou(t, i, j0 + j, &bad, pl));
          }
        __syncthreads();
      }
    }
  }
  if (bad) atomicExch(t.errflag, 1);
  if (!live) return;
  float *cls = t.cls_targets + i * t.num_classes;
  float *reg = t.reg_targets + i * 9;
  const bool pos = best > t.pos_thresh;  // box_utils.py:195 (strict >)
  const int cj = pos ? t.g_class[best_j] : -1;
  for (int c = 0; c < t.num_classes; ++c) cls[c] = (c == cj) ? 1.0f : 0.0f;
  float r[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (pos) make_target_dev(t, i, best_j, r);
#pragma unroll
  for (int d = 0; d < 9; ++d) reg[d] = r[d];
}

// T2: first anchor index that reaches each column maximum (np.argmax over the
// transposed matrix, box_utils.py:199-200): one lane per listed pair.  If the
// list overflowed (more pairs with IoU > 0 than anchors: never on real scenes),
// every lane re-scans its anchor instead -- the IoU is deterministic, so the
// recomputed bits are identical.
__global__ __launch_bounds__(kIouThreads) void k_targets_cols(TargetArgs t) {
  __shared__ double2 s_poly[16 * kIouThreads];
  const int64_t i = (int64_t)blockIdx.x * kIouThreads + threadIdx.x;
  const unsigned count = *t.cand_count;
  if (count <= (unsigned)t.cand_cap) {
    if (i < count) {
      const int4 c = t.cand[i];
      const u64 bits = ((u64)(unsigned)c.w << 32) | (unsigned)c.z;
      if (bits == t.col_max[c.y]) atomicMin(&t.col_arg[c.y], c.x);
    }
    return;
  }
  if (i >= t.A) return;
  const PolyLds pl{s_poly + threadIdx.x, kIouThreads};
  const AnchorId aid = anchor_id(t, i);
  const double acx = aid.cx, acy = aid.cy;
  bool bad = false;
  for (int j = 0; j < t.G; ++j) {
    const double gcx = t.g_centers_img[(int64_t)j * 3], gcy = t.g_centers_img[(int64_t)j * 3 + 1];
    if (gate_far(acx, acy, gcx, gcy)) continue;
    const double v = pair_iou(t, i, j, &bad, pl);
    if (v > 0.0 && (u64)__double_as_longlong(v) == t.col_max[j]) atomicMin(&t.col_arg[j], (int)i);
  }
}

// T3 (one workgroup): the highest-IoU anchor of every ground truth overrides
// the row written by T1 (box_utils.py:204-205, 212-213, 223-228).  A column
// whose argmax is anchor 0 -- all-zero columns included -- is dropped, exactly
// like the reference's np.nonzero filter.
constexpr int kForcedLds = 2048;  // ground truths whose forced anchor / class are staged in LDS

__global__ __launch_bounds__(kIouThreads) void k_targets_forced(TargetArgs t) {
  const int G = t.G;
  if (G <= kForcedLds && t.num_classes <= 64) {
    // One round trip to global memory for the column results, everything else in LDS.
    // (The general path below is a chain of dependent global round trips, ~1.5 us each.)
    __shared__ int s_i[kForcedLds];
    __shared__ unsigned char s_c[kForcedLds];
    for (int j = threadIdx.x; j < G; j += kIouThreads) {
      s_i[j] = (t.col_max[j] != 0ull) ? t.col_arg[j] : 0;
      s_c[j] = (unsigned char)t.g_class[j];
      t.col_max[j] = 0ull;  // re-arm the scratch words for the next call on this context
      t.col_arg[j] = INT_MAX;
    }
    if (threadIdx.x == 0) *t.cand_count = 0u;
    __syncthreads();
    for (int j = threadIdx.x; j < G; j += kIouThreads) {
      const int i = s_i[j];
      if (i == 0) continue;
      // class row of anchor i: ones at the classes of ALL ground truths forcing it (every
      // duplicate writes the same full row); regression row: the last ground truth wins
      u64 mask = 0;
      bool later = false;
      for (int j2 = 0; j2 < G; ++j2)
        if (s_i[j2] == i) {
          mask |= 1ull << (s_c[j2] & 63);
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
    return;
  }
  // phase A: clear the class rows of all forced anchors
  for (int j = threadIdx.x; j < G; j += kIouThreads) {
    const int i = (t.col_max[j] != 0ull) ? t.col_arg[j] : 0;
    if (i != 0) {
      float *cls = t.cls_targets + (int64_t)i * t.num_classes;
      for (int c = 0; c < t.num_classes; ++c) cls[c] = 0.0f;
    }
  }
  __threadfence_block();
  __syncthreads();
  // phase B: set the classes (duplicates of one anchor set several ones, as
  // numpy's fancy assignment does); regression row: the last ground truth wins
  for (int j = threadIdx.x; j < G; j += kIouThreads) {
    const int i = (t.col_max[j] != 0ull) ? t.col_arg[j] : 0;
    if (i == 0) continue;
    if ((unsigned)t.g_class[j] < (unsigned)t.num_classes)  // a class outside the row is ignored, never written
      t.cls_targets[(int64_t)i * t.num_classes + t.g_class[j]] = 1.0f;
    bool later = false;
    for (int j2 = j + 1; j2 < G; ++j2) {
      const int i2 = (t.col_max[j2] != 0ull) ? t.col_arg[j2] : 0;
      later = later || (i2 == i);
    }
    if (!later) {
      float r[9];
      make_target_dev(t, i, j, r);
      float *reg = t.reg_targets + (int64_t)i * 9;
      for (int d = 0; d < 9; ++d) reg[d] = r[d];
    }
  }
  // re-arm the scratch words for the next call on this context
  __syncthreads();
  for (int j = threadIdx.x; j < G; j += kIouThreads) {
    t.col_max[j] = 0ull;
    t.col_arg[j] = INT_MAX;
  }
  if (threadIdx.x == 0) *t.cand_count = 0u;
}

__global__ void k_targets_init(u64 *col_max, int *col_arg, int G, int *errflag,
                               unsigned *cand_count) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < G) {
    col_max[j] = 0ull;
    col_arg[j] = INT_MAX;
  }
  if (j == 0) {
    *errflag = 0;
    *cand_count = 0u;
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
  // scratch: [0,4096) flags/counters | col_max[Gcap] | col_arg[Gcap] | cand[A]
  const size_t gcap = (size_t)std::max<int64_t>(G, 1);
  const size_t off_cmax = 4096, off_carg = off_cmax + gcap * 8;
  const size_t off_cand = (off_carg + gcap * 4 + 255) / 256 * 256;
  const size_t need = off_cand + (size_t)A * 16;
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
  t.col_max = reinterpret_cast<u64 *>(ws + off_cmax);
  t.col_arg = reinterpret_cast<int *>(ws + off_carg);
  t.cand = reinterpret_cast<int4 *>(ws + off_cand);
  t.cand_cap = (int)A;
  t.cls_targets = cls_targets;
  t.reg_targets = reg_targets;
  // The scratch words are re-armed by k_targets_forced at the end of every call; only a
  // fresh / regrown / re-shaped workspace (or a call without ground truths) needs the init.
  const unsigned long long key = ((unsigned long long)A << 20) ^ (unsigned long long)gcap;
  if (grew || ctx->tgt_key != key || G == 0) {
    const unsigned gb = (unsigned)((gcap + 255) / 256);
    hipLaunchKernelGGL(k_targets_init, dim3(gb), dim3(256), 0, stream, t.col_max, t.col_arg,
                       (int)G, t.errflag, t.cand_count);
    ctx->tgt_key = key;
  }
  const unsigned ab = (unsigned)((A + kIouThreads - 1) / kIouThreads);
  hipLaunchKernelGGL(k_targets_rows, dim3(ab), dim3(kIouThreads), 0, stream, t);
  if (G > 0) {
    hipLaunchKernelGGL(k_targets_cols, dim3(ab), dim3(kIouThreads), 0, stream, t);
    hipLaunchKernelGGL(k_targets_forced, dim3(1), dim3(kIouThreads), 0, stream, t);
  }
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
