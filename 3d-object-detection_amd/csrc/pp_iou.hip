// pp_iou.hip -- rotated-box anchor/ground-truth IoU and the fused anchor-target
// assignment as hand-written HIP for gfx950.
//
// Replaces make_ious + iou (/root/reference data/pillars.cpp:400-427, 132-172)
// and, fused on the device-resident path, create_target + make_target
// (utils/box_utils.py:162-232, 70-109) so that the [A,G] IoU matrix is never
// materialised.  Kernels: k_targets<false> (the whole of create_target in one launch: gate,
// pair queue, clip with 8 lanes per pair, row / column reductions, target rows, and the forced
// rows by the last workgroup to finish), k_targets<true> (the dense matrix of the
// compatibility API: the same gate / queue / clip, the pairs' values written into a zeroed
// matrix), k_targets_init (arms the scratch words once).
//
// Boost.Geometry (bg::intersection / bg::area, pillars.cpp:160,164,165) is an
// absent, unpinned third-party dependency of the reference; its published
// behaviour for convex quads is restated as Sutherland-Hodgman clipping plus a
// shoelace area, executed in f64 with contraction disabled in exactly the
// operation order of oracle/pp_oracle.c, so HIP-vs-oracle parity is bit-exact.

#include "pp_common.h"

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstring>
#include <chrono>
#include <cstdlib>

namespace pp {

using u64 = unsigned long long;

__device__ __forceinline__ double shoelace_dev(const double *q, int n) {
  double s = 0.0;
  for (int k = 0; k < n; ++k) {
    const int j = (k + 1 == n) ? 0 : k + 1;
    s = s + (q[2 * k] * q[2 * j + 1] - q[2 * j] * q[2 * k + 1]);
  }
  return 0.5 * s;
}

// the +-10 cell centre gate of pillars.cpp:418-419
__device__ __forceinline__ bool gate_far(double acx, double acy, double gcx, double gcy) {
  return (fabs(acx - gcx) > 10.0) || (fabs(acy - gcy) > 10.0);
}

// ------------------------------------------------------------------------- //
// fused target assignment                                                    //
// ------------------------------------------------------------------------- //
struct ColEntry {
  u64 key;   // gt | anchor << 32
  u64 bits;  // IoU
};
static_assert(sizeof(ColEntry) == 16, "list entry layout");

struct TargetArgs {
  int64_t A;
  int G;
  const double *a_corners, *a_centers, *a_wlh, *a_yaw;  // anchor arrays, or all NULL with ...
  // ... the anchor grid of make_anchor_boxes (box_utils.py:111-159) evaluated on the fly:
  // anchor i = (y*fm_w + x)*per_cell + d; centre ((x+.5)/fm_scale, (y+.5)/fm_scale, z_d)
  int grid, fm_w, per_cell;
  double fm_scale;
  double inv_scale;     // 1 / fm_scale when fm_scale is a power of two (then c * inv_scale IS c / fm_scale, bit for
                        // bit, and an f64 division costs a wave ~15 instructions), else 0: divide
  const double *types;  // [per_cell][kTypeCols]: corner offsets x0,y0..x3,y3, w, l, h, yaw, z
  int a_center_cols, g_center_cols;  // doubles per row of a_centers / g_centers_img (3; the dense-matrix API: >= 2)
  const double *g_corners, *g_centers_img, *g_centers, *g_wlh, *g_yaw;
  const int *g_class;
  double *ious;  // matrix mode: [A][G] f64, zeroed by the host; the pairs past the gate are written here
  // matrix mode, sparse form (the host drop-in pp_make_ious_f64): instead of the dense matrix, the entries that are
  // not zero as {anchor, box, IoU} records -- 0.16 % of the pairs at BASELINE config 3 -- appended through one counter;
  // a record beyond triple_cap is counted and dropped (the host then takes the dense form)
  struct IouTriple *triples;
  unsigned *triple_count;
  unsigned triple_cap;
  double pos_thresh, canvas_height;
  int num_classes;
  // scratch
  u64 *col_max;  // [G] bit pattern of the column maximum IoU (0 = all zero): the tail's scratch when
  u64 *col_win;  // [G] {first anchor reaching it, its list entry}             G is beyond its LDS
  int *errflag;
  // per workgroup and ground truth it reaches with IoU > 0: {gt, first anchor of the workgroup's
  // column maximum}, the maximum's bits; at most G entries per workgroup
  ColEntry *cand;        // [workgroups * G]
  unsigned *cand_count;  // entries appended
  unsigned *ticket;      // groups of workgroups finished (second level)
  // First level: workgroup w bumps ticket1[(w >> ticket_shift) * kTicketPad]; the last of its group resets
  // that word and bumps `ticket`.  (One word for all workgroups: atomics on one address are served one
  // after the other, ~25 ns each -- 2110 workgroups of the reference's anchor set queued 50 us there.)
  unsigned *ticket1;
  int ticket_shift, ticket_groups;  // first-level tickets reserved per sample
  // The box-centric form (k_targets_gt, grid anchors): see there
  int fm_h, gt_splits;       // feature-map rows; workgroups per ground truth
  int cand_per_wg;           // candidate anchors per PAIR workgroup (<= kCandPerWg)
  unsigned cand_per_gt;      // list entries reserved per ground truth (anchor-centric form: its workgroups)
  struct PosKey *pos;        // [boxes][pos_per_gt] pairs above the threshold {anchor, box, IoU} ...
  struct PosRow *pos_rows;   // ... and, index for index, their regression rows + the box's class
  unsigned pos_per_gt;
  unsigned *pos_count;       // per sample
  float *cand_rows;          // [boxes][gt_splits][kRowPitch]: the regression row of each column-list slot's pair
  u64 *best;                 // [samples][A]: highest IoU above the threshold an anchor has seen (0 = none); and
  unsigned *bestj;           // [samples][A]: the first box reaching it (~0 = none) -- the two-pass tail's scratch (more
                             // than 1024 pairs above the threshold): armed, raised and re-armed by that tail alone
  // outputs
  float *cls_targets;  // [A][num_classes]
  float *reg_targets;  // [A][9]
};
// bits of the error word (pp_iou_check reports and clears them; both may be set by one launch)
constexpr int kErrWinding = 1;      // IoU < 0: a box with the wrong corner winding (pillars.cpp:166-169)
constexpr int kErrPosOverflow = 2;  // more pairs above the threshold than the positive list holds: entries dropped

struct IouTriple {
  unsigned anchor, box;
  double iou;
};
static_assert(sizeof(IouTriple) == 16, "triple layout");
// A pair above the threshold is two records in two arrays, index for index: the 16-byte key the tail's threads read
// side by side (lane e <-> key e: one wave-level load is 1024 consecutive bytes) and the 48-byte row.  (One 64-byte
// record per pair, read field by field with the lanes 64 bytes apart, made every wave-level load of the tail touch
// 64 lines and ask for every line a dozen times: 2 us between the last ticket and the first LDS phase.)
struct PosKey {
  unsigned anchor, gt;
  u64 bits;
};
struct PosRow {
  float row[9];  // the regression row of (anchor, box), worked out where the pair was clipped
  int cls;       // the box's class
  float pad[2];
};
static_assert(sizeof(PosKey) == 16 && sizeof(PosRow) == 48, "pair entry layout");
constexpr int kRowPitch = 12;  // floats between two rows of cand_rows (48 B = three 16-byte groups)
__device__ __forceinline__ ColEntry *cand_at(const TargetArgs &t, size_t e) { return t.cand + e; }

constexpr int kTypeCols = 13;

// A batch of samples in one launch (grid.y = sample; the reference prepares BATCH_SIZE samples per step,
// config.py:135, each through create_target, data/dataset.py:113-118): the anchors are shared, sample b's
// ground truths are rows [g_off[b], g_off[b+1]) of the concatenated g_* arrays, its outputs follow sample
// b-1's, and every piece of scratch the samples' tails depend on (list, counter, ticket, column words) is
// the sample's own -- a shared ticket would let one sample's tail read another one's unfinished list.
struct TargetBatch {
  int g_off[PP_MAX_BATCH + 1];
};
constexpr int kCounterStride = 32;  // unsigned words between two samples' counters (128 B: own cache lines)
constexpr int kTicketPad = 16;      // unsigned words between two first-level tickets (64 B)

struct AnchorId {
  int d;          // anchor type within the cell
  double cx, cy;  // centre
};

// the view of sample b: everything per-sample moved to its rows (all wave-uniform: SGPR arithmetic)
__device__ __forceinline__ void sample_view(TargetArgs &t, const TargetBatch &bt, int b) {
  const int o = bt.g_off[b];
  t.G = bt.g_off[b + 1] - o;
  t.g_corners += (int64_t)o * 8;
  t.g_centers_img += (int64_t)o * t.g_center_cols;
  t.g_centers += (int64_t)o * 3;
  t.g_wlh += (int64_t)o * 3;
  t.g_yaw += o;
  t.g_class += o;
  t.col_max += o;
  t.col_win += o;
  t.cand += (size_t)t.cand_per_gt * (size_t)o;  // sample b appends at most cand_per_gt * G_b entries
  t.pos += (size_t)t.pos_per_gt * (size_t)o;
  if (t.pos_rows) t.pos_rows += (size_t)t.pos_per_gt * (size_t)o;
  if (t.cand_rows) t.cand_rows += (size_t)t.cand_per_gt * (size_t)o * kRowPitch;
  t.pos_count += b * kCounterStride;
  if (t.best) {
    t.best += (int64_t)b * t.A;
    t.bestj += (int64_t)b * t.A;
  }
  t.cand_count += b * kCounterStride;
  t.ticket += b * kCounterStride;
  t.ticket1 += (size_t)b * (size_t)t.ticket_groups * kTicketPad;
  t.cls_targets += (int64_t)b * t.A * t.num_classes;
  t.reg_targets += (int64_t)b * t.A * 9;
}

// centre coordinate of feature-map cell x: (x + .5) / fm_scale (box_utils.py:137-138), same f64 result
__device__ __forceinline__ double cell_centre(const TargetArgs &t, unsigned x) {
  const double c = (double)x + 0.5;
  return t.inv_scale != 0.0 ? c * t.inv_scale : c / t.fm_scale;
}

__device__ __forceinline__ AnchorId anchor_id(const TargetArgs &t, int64_t i) {
  AnchorId a;
  if (t.grid) {  // A <= INT_MAX / 2 (host check): 32-bit divisions
    const unsigned iu = (unsigned)i, cell = iu / (unsigned)t.per_cell;
    a.d = (int)(iu - cell * (unsigned)t.per_cell);
    const unsigned y = cell / (unsigned)t.fm_w, x = cell - y * (unsigned)t.fm_w;
    a.cx = cell_centre(t, x);
    a.cy = cell_centre(t, y);
  } else {
    a.d = 0;
    a.cx = t.a_centers[i * t.a_center_cols];
    a.cy = t.a_centers[i * t.a_center_cols + 1];
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

// utils/box_utils.py:70-109 on plain values: one whole row on one lane -- the tail's forced rows (its
// only use: inlined there; as an out-of-line function its callee-saved registers went through scratch
// memory).  The workgroups' positives and the usual tail use target_quotient / _logratio / _angle, one kind per wave.
struct Row9 {
  float v[9];
};
struct BoxVals {
  double x, y, z, w, l, h, yaw;
};
__device__ __forceinline__ Row9 target_row(const BoxVals &a, const BoxVals &g, double canvas_height) {
  Row9 res;
  float *out = res.v;
  const double ax = a.x, ay = a.y, az = a.z, aw = a.w, al = a.l, ah = a.h, at = a.yaw;
  const double gx = g.x, gz = g.z, gw = g.w, gl = g.l, gh = g.h;
  double gy = g.y, gt = g.yaw;
  const double ad = sqrt(aw * aw + al * al);
  const double pi = 3.141592653589793;  // np.pi
  gy = (canvas_height - 1) - gy;         // box_utils.py:83
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
  return res;
}

__device__ __forceinline__ BoxVals anchor_vals(const TargetArgs &t, int64_t i) {
  BoxVals a;
  if (t.grid) {
    const AnchorId id = anchor_id(t, i);
    const double *ty = t.types + id.d * kTypeCols;
    a.x = id.cx, a.y = id.cy, a.z = ty[12];
    a.w = ty[8], a.l = ty[9], a.h = ty[10], a.yaw = ty[11];
  } else {
    a.x = t.a_centers[i * 3], a.y = t.a_centers[i * 3 + 1], a.z = t.a_centers[i * 3 + 2];
    a.w = t.a_wlh[i * 3], a.l = t.a_wlh[i * 3 + 1], a.h = t.a_wlh[i * 3 + 2];
    a.yaw = t.a_yaw[i];
  }
  return a;
}
__device__ __forceinline__ BoxVals gt_vals(const TargetArgs &t, int j) {
  BoxVals g;
  g.x = t.g_centers[j * 3], g.y = t.g_centers[j * 3 + 1], g.z = t.g_centers[j * 3 + 2];
  g.w = t.g_wlh[j * 3], g.l = t.g_wlh[j * 3 + 1], g.h = t.g_wlh[j * 3 + 2];
  g.yaw = t.g_yaw[j];
  return g;
}

// target_row's values sorted by what they cost, for callers that give each WAVE one kind (a wave
// executes every branch its lanes take: with the kinds mixed inside a wave every row pays for a square
// root, divisions, a logarithm AND a sine -- 2.6 us for the 40 forced rows of a tail).  Same
// expressions on the same operands as target_row: same bits.  Operands come as
// plain values (picking fields of a struct by a run-time index made the compiler spill it to scratch).
__device__ __forceinline__ float target_quotient(double g_c, double a_c, double a_w, double a_l, double a_h,
                                                 double canvas_height, int c) {
  const double ad = sqrt(a_w * a_w + a_l * a_l);
  const double gq = c == 1 ? (canvas_height - 1) - g_c : g_c;  // box_utils.py:83
  const double den = c < 2 ? ad : a_h;
  return (float)((gq - a_c) / den);  // dx, dy, dz
}
// the same with the anchor's diagonal handed in (ad = sqrt(a_w * a_w + a_l * a_l), the expression above, evaluated once
// per anchor TYPE instead of once per pair): same operands, same operations, same bits
__device__ __forceinline__ float target_quotient_ad(double g_c, double a_c, double ad, double a_h, double canvas_height,
                                                    int c) {
  const double gq = c == 1 ? (canvas_height - 1) - g_c : g_c;  // box_utils.py:83
  const double den = c < 2 ? ad : a_h;
  return (float)((gq - a_c) / den);  // dx, dy, dz
}
__device__ __forceinline__ float target_logratio(double g_s, double a_s) {
  return (float)log(g_s / a_s);  // dw, dl, dh
}
__device__ __forceinline__ void target_angle(double g_yaw, double a_yaw, float *dt, float *ort) {
  const double pi = 3.141592653589793;  // np.pi
  double gt = g_yaw;
  if (gt <= pi && gt >= pi / 2)  // box_utils.py:92-95
    gt -= pi;
  else if (gt >= -pi && gt <= -pi / 2)
    gt += pi;
  const double df = gt - a_yaw;
  *dt = (float)sin(df);
  *ort = ((df <= pi && df >= pi / 2) || (df >= -pi && df <= -pi / 2)) ? 1.0f : 0.0f;  // :99-102
}

__device__ __forceinline__ Row9 make_target_dev(const TargetArgs &t, int64_t i, int j) {
  return target_row(anchor_vals(t, i), gt_vals(t, j), t.canvas_height);
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
constexpr int kStageCols = 10;  // widest target row staged in LDS (the reference has 9 classes, config.py:97; wider rows take scalar stores)
constexpr int kForcedLds = 1024;  // ground truths whose column results the tail keeps in LDS

constexpr int kMaxNP = 1;        // pairs a group of 8 lanes clips side by side: 2 (and 3) measured no faster than
                                 // as many rounds of one -- a round is issue-bound -- and cost 30 VGPRs
constexpr int kLdsTypes = 8;     // anchor types per cell whose table is kept in LDS
constexpr int kTailBatch = 8;    // list entries per thread the tail keeps in registers

struct TgtLds {
  double2 gc[kGtChunk];                   // image-space centres of the chunk's ground truths
  double2 gk[kGtChunk][4];                // their corners
  double garea[kGtChunk];                 // their declared-orientation areas
  union {
    struct {                              // the clip phase's ...
      double2 poly[kMaxNP][kPairsPerRound][kGroup + 1];  // hand-over of a clip pass's output rings (+ a spare slot)
      double terms[kMaxNP][kPairsPerRound][kGroup];      // shoelace terms on their way to the ordered sum
      double iou[kPairCap];
    };
    float stage[kTgtThreads * kStageCols];  // ... and, after it, the class rows on their way out
  };
  float rstage[kTgtThreads * 9];          // the regression rows on their way out
  int gcls[kGtChunk];                     // classes of the chunk's ground truths
  double gv7[8];                          // box-centric form: x, y, z, w, l, h, yaw of the workgroup's box
  double rc_ad[kLdsTypes];                // ... per anchor type of the cell: sqrt(w^2 + l^2) (box_utils.py:79)
  float rc_lg[kLdsTypes][3];              // ... log(gw/aw), log(gl/al), log(gh/ah) (:88-90)
  float rc_dt[kLdsTypes], rc_ort[kLdsTypes];  // ... sin(gt - at), the orientation bit (:92-102)
  float colrow[12];                       // ... the regression row of its column maximum's pair
  u64 need;                               // ... which lanes' pairs need a row
  u64 cmax[kGtChunk], cseen[kGtChunk];    // column maximum of this workgroup / as of the last window
  int carg[kGtChunk];                     // first anchor reaching it
  unsigned short pair_lane[kPairCap], pair_gt[kPairCap];
  double2 acen[kTgtThreads];              // the workgroup's anchor centres ...
  unsigned short atype[kTgtThreads];      // ... and types (grid anchors)
  double types[kLdsTypes][kTypeCols];
  double bbox[kTgtWaves][4];
  int woff[kTgtWaves];
  int is_last, contrib;
};
constexpr int kTailGt = 128;  // ground truths whose box values the tail stages in LDS for the forced rows
struct TailLds {
  u64 colmax[kForcedLds];
  u64 colwin[kForcedLds];
  double gtv[kTailGt][7];                 // x, y, z, w, l, h, yaw of the sample's ground truths
  double types[kLdsTypes][kTypeCols];     // the anchor type table again (the workgroup's copy lies under colmax)
};
static_assert(sizeof(TailLds) <= sizeof(TgtLds), "the tail reuses the workgroup's LDS");
constexpr size_t kTgtLdsBytes = sizeof(TgtLds) > sizeof(TailLds) ? sizeof(TgtLds) : sizeof(TailLds);

__device__ __forceinline__ void iou_wave_sync() {
  // LDS operations of one wave execute in program order; this only stops the compiler from
  // moving them across a cross-lane hand-off
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// shoelace_dev over a ring of n vertices in LDS (ring[0..n), this group's slots), lane v holding
// vertex v in (x, y): the terms are formed in parallel, handed over through LDS (terms[0..8)) and
// summed in the serial order by every lane.  (Neighbour vertices and the terms come from LDS reads,
// not from ds_bpermute shuffles: five LDS instructions instead of twenty.)
template <int KMAX = kGroup>
__device__ __forceinline__ double group_shoelace(double x, double y, int n, int v, const double2 *ring,
                                                 double *terms) {
  const double2 j = ring[(v + 1 >= n) ? 0 : v + 1];
  terms[v] = x * j.y - j.x * y;
  iou_wave_sync();
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < KMAX; k += 2) {
    const double2 t2 = *reinterpret_cast<const double2 *>(terms + k);
    if (k < n) s = s + t2.x;
    if (k + 1 < n) s = s + t2.y;
  }
  iou_wave_sync();
  return 0.5 * s;
}

// iou_pair_dev with the ring spread over the 8 lanes of a group: lane v holds vertex v, the ring
// is mirrored in LDS (poly[0..n)) so that a lane reads its predecessor from there.  Every vertex
// goes through the same operations in the same order as in the serial loop (dp of vertex i is dc
// of vertex i-1: the same function of the same operands, evaluated again), and the output ring is
// laid out in the serial order (crossing point before the kept vertex, vertices ascending): same
// bits.  A group clips NP pairs side by side (NP = 1 is what ships, see kMaxNP).  (cx, cy): this
// lane's anchor corner of pair u (v < 4); gk: that pair's ground-truth corners in LDS; area_g: its
// area.  All 64 lanes call it together; lanes of idle groups / idle pairs pass zeros.
template <int NP>
__device__ __forceinline__ void clip_groups(double (&cx)[NP], double (&cy)[NP], const double2 *(&gk)[NP],
                                            const double (&area_g)[NP], double2 *(&poly)[NP], double *(&terms)[NP],
                                            int v, int gbase, double (&iou)[NP], bool (&wrong)[NP]) {
  double area_a[NP];
  int n[NP];
#pragma unroll
  for (int u = 0; u < NP; ++u) poly[u][v] = make_double2(cx[u], cy[u]);  // the anchor quad (v < 4; zeros beyond)
  iou_wave_sync();
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    area_a[u] = group_shoelace<4>(cx[u], cy[u], 4, v, poly[u], terms[u]);
    wrong[u] = (area_a[u] < 0.0 || area_g[u] < 0.0);
    n[u] = 4;
  }
  const unsigned below = (1u << v) - 1u;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ia = (4 - e) & 3, ib = (3 - e) & 3;
    // stage by stage over the pairs, so that the source order already interleaves their chains
    double ex[NP], ey[NP], ax[NP], ay[NP], dc[NP], dp[NP], px[NP], py[NP], tt[NP];
    unsigned cb[NP], kb[NP];
    bool cross[NP], keep[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const double2 a = gk[u][ia], b = gk[u][ib];
      const double2 p = poly[u][(v == 0) ? max(n[u] - 1, 0) : v - 1];  // the predecessor vertex
      ax[u] = a.x;
      ay[u] = a.y;
      ex[u] = b.x - a.x;
      ey[u] = b.y - a.y;
      px[u] = p.x;
      py[u] = p.y;
    }
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      dc[u] = ex[u] * (cy[u] - ay[u]) - ey[u] * (cx[u] - ax[u]);
      dp[u] = ex[u] * (py[u] - ay[u]) - ey[u] * (px[u] - ax[u]);
    }
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const bool act = v < n[u];
      const bool in_c = dc[u] >= 0.0, in_p = dp[u] >= 0.0;
      cross[u] = act && (in_c != in_p);
      keep[u] = act && in_c;
      cb[u] = (unsigned)(__ballot(cross[u]) >> gbase) & 0xFFu;
      kb[u] = (unsigned)(__ballot(keep[u]) >> gbase) & 0xFFu;
    }
    // straight-line code: every lane divides and stores, lanes with nothing to emit into the
    // spare slot.  (A ninth vertex cannot come from two convex quads; it would land there too.)
#pragma unroll
    for (int u = 0; u < NP; ++u) tt[u] = dp[u] / (dp[u] - dc[u]);
    iou_wave_sync();  // every lane has its predecessor: the ring may be overwritten
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int pos = __popc(cb[u] & below) + __popc(kb[u] & below);
      const int pc = cross[u] ? min(pos, kGroup) : kGroup;
      const int pk = keep[u] ? min(pos + (cross[u] ? 1 : 0), kGroup) : kGroup;
      poly[u][pc] = make_double2(px[u] + tt[u] * (cx[u] - px[u]), py[u] + tt[u] * (cy[u] - py[u]));
      poly[u][pk] = make_double2(cx[u], cy[u]);
      n[u] = min(__popc(cb[u]) + __popc(kb[u]), kGroup);
    }
    iou_wave_sync();
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const double2 c = poly[u][v];
      cx[u] = (v < n[u]) ? c.x : cx[u];
      cy[u] = (v < n[u]) ? c.y : cy[u];
    }
  }
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const double inter = group_shoelace(cx[u], cy[u], n[u], v, poly[u], terms[u]);
    double r = inter / (area_a[u] + area_g[u] - inter);
    if (!(inter > 0.0)) r = 0.0;
    if (n[u] < 3) r = 0.0;
    if (wrong[u]) r = -1.0;
    iou[u] = r;
  }
}

// one round of a window: pairs r0 + u*32 + group, u < NP
template <int NP>
__device__ __forceinline__ void clip_round(const TargetArgs &t, TgtLds &S, int r0, int wn, int64_t i0, int tid,
                                           int v, int gbase, bool lds_types, bool &bad) {
  double cx[NP], cy[NP], area_g[NP], iou[NP];
  const double2 *gk[NP];
  double2 *poly[NP];
  double *terms[NP];
  bool wrong[NP];
  int q[NP];
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    q[u] = r0 + u * kPairsPerRound + (tid >> 3);
    const bool on = q[u] < wn;
    const int pl = on ? S.pair_lane[q[u]] : 0, pg = on ? S.pair_gt[q[u]] : 0;
    cx[u] = cy[u] = 0.0;
    if (on && v < 4) {
      if (t.grid) {
        const double *ty = lds_types ? S.types[S.atype[pl]] : t.types + (int)S.atype[pl] * kTypeCols;
        cx[u] = ty[2 * v] + S.acen[pl].x;
        cy[u] = ty[2 * v + 1] + S.acen[pl].y;
      } else {
        cx[u] = t.a_corners[(i0 + pl) * 8 + 2 * v];
        cy[u] = t.a_corners[(i0 + pl) * 8 + 2 * v + 1];
      }
    }
    gk[u] = S.gk[pg];
    area_g[u] = on ? S.garea[pg] : 0.0;
    poly[u] = S.poly[u][tid >> 3];
    terms[u] = S.terms[u][tid >> 3];
  }
  clip_groups<NP>(cx, cy, gk, area_g, poly, terms, v, gbase, iou, wrong);
#pragma unroll
  for (int u = 0; u < NP; ++u)
    if (q[u] < wn && v == 0) {
      S.iou[q[u]] = iou[u];
      bad = bad || wrong[u];
    }
}

__device__ __forceinline__ u64 ld_agent(const u64 *p) {
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
// `total` zero floats to dst (16-byte aligned): the rows of a workgroup without a positive anchor
// (most of them) go out straight from registers, no LDS stage, no barrier
__device__ __forceinline__ void store_zero_rows(float *dst, int total, int tid) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, total * 4, 0x00020000);
  const int n4 = total >> 2;
  const v4u z = {0u, 0u, 0u, 0u};
  for (int k = tid; k < n4; k += kTgtThreads) __builtin_amdgcn_raw_buffer_store_b128(z, rs, k * 16, 0, kAuxSc1);
  for (int k = (n4 << 2) + tid; k < total; k += kTgtThreads) __builtin_amdgcn_raw_buffer_store_b32(0u, rs, k * 4, 0, kAuxSc1);
}
__device__ __forceinline__ void store_f32_sc1(float *p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The hand-off to a sample's last workgroup (k_targets, k_targets_gt).  Shipped form: every byte the tail reads
// or overwrites is stored write-through (sc1), every storing wave drains (s_waitcnt vmcnt(0)), a workgroup
// barrier, then ONE lane's agent-scope add on the ticket; the workgroup whose add came last learns so from
// the value the add returned, joins a barrier, and loads the bytes with sc1 loads only (DESIGN.md section 5
// maps this onto the guide's table and says where it departs: two-level tickets, several workgroups per CU).
// -DPP_STRICT_HANDOFF builds the architecturally guaranteed form beside it -- lane 0's agent-scope RELEASE
// fence (+ drain) in front of every ticket, no early ticket, an agent-scope ACQUIRE (+ drain + barrier) at
// the head of the tail -- for tests/test_gpu_handoff.py, which asserts that both libraries give the same
// bits over the fuzz cases and the alternating-input soak.  It is test infrastructure: ~2 us per launch.
#ifdef PP_STRICT_HANDOFF
constexpr bool kStrictHandoff = true;
#else
constexpr bool kStrictHandoff = false;
#endif
__device__ __forceinline__ void handoff_release_lane0() {  // behind the barrier that follows every wave's drain
  if (kStrictHandoff) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (inline asm: the compiler may drop the fence's own wait)
  }
}
__device__ __forceinline__ void handoff_acquire_tail() {  // every wave of the last workgroup
  if (kStrictHandoff) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

// The last workgroup: column argmax from the workgroups' entries, then the forced rows.  A
// column whose argmax is anchor 0 -- all-zero columns included -- is dropped, exactly like the
// reference's np.nonzero filter (box_utils.py:204-205).
// IN_LDS: the column words live in LDS (G <= kForcedLds), else in the armed global scratch
// (0 / all ones), which only this workgroup touches: atomics and sc1 loads meet in its XCD's L2.
#ifdef PP_IOU_STAMPS
__device__ unsigned long long g_iou_stamps[16 * 4096];
#define IOU_STAMP(k) do { const unsigned wg_ = blockIdx.x * gridDim.y + blockIdx.y; if (threadIdx.x == 0 && wg_ < 4096) g_iou_stamps[wg_ * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define IOU_STAMP(k) do {} while (0)
#endif

// n_fixed >= 0: the list has exactly that many slots, every one written by this launch (an unused one carries
// the bits 0): the box-centric form, no counter to wait for
template <bool IN_LDS>
__device__ void targets_tail(const TargetArgs &t, TailLds &T, int n_fixed = -1) {
  const int G = t.G, tid = threadIdx.x;
  const unsigned n = n_fixed >= 0 ? (unsigned)n_fixed
                                  : __hip_atomic_load(t.cand_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  auto cmax_at = [&](int j) -> u64 * { return IN_LDS ? &T.colmax[j] : &t.col_max[j]; };
  auto cwin_at = [&](int j) -> u64 * { return IN_LDS ? &T.colwin[j] : &t.col_win[j]; };
  auto cmax_ld = [&](int j) -> u64 { return IN_LDS ? T.colmax[j] : ld_agent(&t.col_max[j]); };
  auto cwin_ld = [&](int j) -> u64 { return IN_LDS ? T.colwin[j] : ld_agent(&t.col_win[j]); };
  auto sync = [&]() {
    if (!IN_LDS) __threadfence();
    __syncthreads();
  };
  if (IN_LDS)
    for (int j = tid; j < G; j += kTgtThreads) {
      T.colmax[j] = 0ull;
      T.colwin[j] = ~0ull;
    }
  __syncthreads();
  // The forced rows' operands come along with the list: the ground truths' box values and the type
  // table are loaded now (in flight with the entries) and wait in LDS -- read when the rows are due, they
  // were a second and third memory round trip at the very end of the launch.
  const bool gt_staged = IN_LDS && G <= kTailGt;
  const bool ty_staged = IN_LDS && t.grid && t.per_cell <= kLdsTypes;
  double pre_g[4] = {0.0, 0.0, 0.0, 0.0}, pre_t = 0.0;
  if (gt_staged) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // kTailGt * 7 <= 4 * kTgtThreads
      const int k = r * kTgtThreads + tid, j = k / 7, c = k - j * 7;
      if (k < G * 7) pre_g[r] = c < 3 ? t.g_centers[j * 3 + c] : c < 6 ? t.g_wlh[j * 3 + c - 3] : t.g_yaw[j];
    }
  }
  static_assert(kTailGt * 7 <= 4 * kTgtThreads, "staging loop");
  if (ty_staged && tid < t.per_cell * kTypeCols) pre_t = t.types[tid];
  // The first 2048 entries (all of them on real scenes) are fetched once, every load in flight
  // together.
  u64 e_key[kTailBatch], e_bits[kTailBatch];
#pragma unroll
  for (int k = 0; k < kTailBatch; ++k) {
    const unsigned e = (unsigned)(k * kTgtThreads + tid);
    e_key[k] = e_bits[k] = 0ull;  // an entry's maximum is never 0
    if (e < n) {
      e_key[k] = ld_agent(&cand_at(t, e)->key);
      e_bits[k] = ld_agent(&cand_at(t, e)->bits);
    }
  }
  IOU_STAMP(11);
  if (gt_staged) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = r * kTgtThreads + tid;
      if (k < G * 7) T.gtv[0][k] = pre_g[r];
    }
  }
  if (ty_staged && tid < t.per_cell * kTypeCols) T.types[0][tid] = pre_t;
#pragma unroll
  for (int k = 0; k < kTailBatch; ++k)
    if (e_bits[k]) atomicMax(cmax_at((int)(e_key[k] & 0xFFFFFFFFull)), e_bits[k]);
  for (unsigned e = kTailBatch * kTgtThreads + tid; e < n; e += kTgtThreads)
    atomicMax(cmax_at((int)(ld_agent(&cand_at(t, e)->key) & 0xFFFFFFFFull)), ld_agent(&cand_at(t, e)->bits));
  sync();
  // the first anchor that reaches the maximum, and the entry it came in
#pragma unroll
  for (int k = 0; k < kTailBatch; ++k) {
    const int j = (int)(e_key[k] & 0xFFFFFFFFull);
    if (e_bits[k] && e_bits[k] == cmax_ld(j))
      atomicMin(cwin_at(j), (e_key[k] & 0xFFFFFFFF00000000ull) | (unsigned)(k * kTgtThreads + tid));
  }
  for (unsigned e = kTailBatch * kTgtThreads + tid; e < n; e += kTgtThreads) {
    const u64 key = ld_agent(&cand_at(t, e)->key);
    const int j = (int)(key & 0xFFFFFFFFull);
    const u64 b = ld_agent(&cand_at(t, e)->bits);
    if (b != 0ull && b == cmax_ld(j)) atomicMin(cwin_at(j), (key & 0xFFFFFFFF00000000ull) | e);
  }
  sync();
  IOU_STAMP(12);
  // colwin[j] <- {anchor ground truth j forces, entry}; anchor 0: none
  for (int j = tid; j < G; j += kTgtThreads) {
    const u64 w = (cmax_ld(j) != 0ull) ? cwin_ld(j) : 0ull;
    if (IN_LDS) {
      // anchor, class (63: a class outside the row, ignored like numpy would raise -- never written)
      const unsigned gc = (unsigned)t.g_class[j];
      // ... and the list entry it came in (bits 6..31)
      T.colwin[j] = (w & 0xFFFFFFFF00000000ull) | ((w & 0x03FFFFFFull) << 6) | (gc < 63u ? gc : 63u);
    } else {
      __hip_atomic_store(&t.col_win[j], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  sync();
  IOU_STAMP(13);
  // the regression row of the last ground truth forcing an anchor
  auto write_row = [&](int i, int j) {
    const Row9 row = make_target_dev(t, i, j);
    float *reg = t.reg_targets + (int64_t)i * 9;
#pragma unroll
    for (int d = 0; d < 9; ++d) reg[d] = row.v[d];
  };
  if (IN_LDS && t.num_classes <= 63 && G <= 64) {
    // The usual case: one ground truth per lane (row j = lane), one KIND of work per wave.  A wave executes
    // every branch its lanes take, and the f64 library functions are chains of ~1 us each: with the kinds mixed
    // inside a wave every row paid for a square root, divisions, a logarithm AND a sine (one row on one lane was
    // a 2.6 us chain; eight lanes per row, one component each, 2 x 1.3 us).  Here the four chains run side by side:
    //   wave 0  the sine, the orientation bit         wave 1  two of the logarithms (independent: they interleave)
    //   wave 2  the third logarithm, the quotients    wave 3  which rows are due at all, and the class rows
    // and meet at one barrier.  Wave 3: the classes of ALL ground truths forcing row j's anchor (duplicates set
    // several ones, box_utils.py:212-213; every duplicate writes the same full class row) and whether a later
    // ground truth forces the same anchor (then ITS regression row wins, :223-228).
    const int ln = tid & 63, wv = tid >> 6;
    const u64 w = ln < G ? T.colwin[ln] : 0ull;
    const int i = (int)(w >> 32);
    auto gval = [&](int k) -> double {  // k: x, y, z, w, l, h, yaw
      if (gt_staged) return T.gtv[ln][k];
      return k < 3 ? t.g_centers[ln * 3 + k] : k < 6 ? t.g_wlh[ln * 3 + k - 3] : t.g_yaw[ln];
    };
    const AnchorId id = anchor_id(t, i);
    const int id_d = id.d;
    auto aval = [&](int k) -> double {  // k >= 2 for grid anchors (their centre is arithmetic: id)
      if (t.grid) {
        const int col = k == 2 ? 12 : k == 6 ? 11 : 5 + k;  // z | yaw | w, l, h
        return ty_staged ? T.types[id_d][col] : t.types[id_d * kTypeCols + col];
      }
      return k < 3 ? t.a_centers[(int64_t)i * 3 + k] : k < 6 ? t.a_wlh[(int64_t)i * 3 + k - 3] : t.a_yaw[i];
    };
    float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f, r3 = 0.0f;
    if (wv == 3) {
      // lanes forcing the same anchor find each other bit by bit: one ballot per bit of the anchor index
      // (a loop over the lanes with v_readlane was 70 ns per ground truth: VALU -> SGPR -> VALU round trips)
      const int cbit = (int)(w & 63ull);
      u64 same = G == 64 ? ~0ull : (1ull << G) - 1ull;
      const int nbits = 32 - __clz((int)t.A);  // i < A
      for (int b = 0; b < nbits; ++b) {
        const bool bit = ((unsigned)i >> b) & 1u;
        const u64 bal = __ballot(bit);
        same &= bit ? bal : ~bal;
      }
      const bool later = ((same >> ln) >> 1) != 0ull;
      u64 mask = 0ull, rest = i != 0 ? same : 0ull;  // (the rows without a forced anchor all "share" anchor 0)
      while (__ballot(rest != 0ull)) {  // as many turns as the largest group has members: one, as a rule
        const int src = rest ? __ffsll((long long)rest) - 1 : ln;
        const int c2 = __shfl(cbit, src);
        mask |= rest ? (1ull << c2) : 0ull;
        rest &= rest - 1ull;
      }
      if (ln < G) T.colmax[ln] = later ? 1ull : 0ull;  // the column maxima are not needed any more
      if (i != 0) {
        float *cls = t.cls_targets + (int64_t)i * t.num_classes;
        for (int c = 0; c < t.num_classes; ++c) cls[c] = ((mask >> c) & 1ull) ? 1.0f : 0.0f;
      }
    } else if (i != 0 && t.cand_rows) {
      // box-centric form: the PAIR workgroup that clipped (anchor i, box j) worked the row out: fetch it
      // (the list entry is in bits 6..31 of colwin; other XCDs wrote it: sc1 loads)
      const float *src = t.cand_rows + (size_t)((unsigned)(w & 0xFFFFFFFFull) >> 6) * kRowPitch;
      auto ldf = [&](int k) { return __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
      if (wv == 0) {
        r0 = ldf(7), r1 = ldf(8);
      } else if (wv == 1) {
        r0 = ldf(4), r1 = ldf(5);
      } else {
        r0 = ldf(6), r1 = ldf(1), r2 = ldf(2), r3 = ldf(3);
      }
    } else if (i != 0) {
      if (wv == 0) {
        target_angle(gval(6), aval(6), &r0, &r1);
      } else if (wv == 1) {
        r0 = target_logratio(gval(3), aval(3));
        r1 = target_logratio(gval(4), aval(4));
      } else {
        const double aw = aval(3), al = aval(4), ah = aval(5);
        r0 = target_logratio(gval(5), ah);
        r1 = target_quotient(gval(0), t.grid ? id.cx : aval(0), aw, al, ah, t.canvas_height, 0);
        r2 = target_quotient(gval(1), t.grid ? id.cy : aval(1), aw, al, ah, t.canvas_height, 1);
        r3 = target_quotient(gval(2), aval(2), aw, al, ah, t.canvas_height, 2);
      }
    }
    __syncthreads();
    if (wv < 3 && i != 0 && T.colmax[ln] == 0ull) {
      float *reg = t.reg_targets + (int64_t)i * 9;
      if (wv == 0) {
        reg[0] = 1.0f;
        reg[7] = r0;
        reg[8] = r1;
      } else if (wv == 1) {
        reg[4] = r0;
        reg[5] = r1;
      } else {
        reg[6] = r0;
        reg[1] = r1;
        reg[2] = r2;
        reg[3] = r3;
      }
    }
  } else if (IN_LDS && t.num_classes <= 63) {
    // class row of a forced anchor i: ones at the classes of ALL ground truths forcing it (every
    // duplicate writes the same full row); regression row: the last ground truth wins.  The scan
    // over the ground truths for duplicates is split over the four waves (quarter q of the
    // range each; a serial LDS scan was 90 ns per ground truth), merged in LDS.
    const int q = tid >> 6, ql = tid & 63;
    const int span = (G + kTgtWaves - 1) / kTgtWaves, qa = q * span, qb = min(G, qa + span);
    for (int jb = 0; jb < G; jb += 64) {
      const int j = jb + ql;
      const u64 w = (j < G) ? T.colwin[j] : 0ull;
      const int i = (int)(w >> 32);
      u64 mask = 0;
      bool later = false;
      if (i != 0) {
#pragma unroll 8
        for (int j2 = qa; j2 < qb; ++j2) {  // branch-free: the reads of one batch must not wait for each other
          const u64 w2 = T.colwin[j2];
          const bool hit = (int)(w2 >> 32) == i;
          mask |= hit ? (1ull << (w2 & 63)) : 0ull;
          later = later || (hit && j2 > j);
        }
      }
      // merge: colmax is free by now
      __syncthreads();
      if (q == 0 && j < G) T.colmax[j] = 0ull;
      __syncthreads();
      if (i != 0) atomicOr(&T.colmax[j], (mask & 0x7FFFFFFFFFFFFFFFull) | (later ? 0x8000000000000000ull : 0ull));
      __syncthreads();
      if (q == 0 && i != 0) {
        const u64 m = T.colmax[j];
        float *cls = t.cls_targets + (int64_t)i * t.num_classes;
        for (int c = 0; c < t.num_classes; ++c) cls[c] = ((m >> c) & 1ull) ? 1.0f : 0.0f;
      }
    }
    // Regression rows, one kind of value per wave (see target_quotient).  colmax[j] bit 63: a later
    // ground truth forces the same anchor and its row wins.
    __syncthreads();
    IOU_STAMP(15);
    // wave 0: the three quotients of every row (lane = row * 3 + component), wave 1: the three
    // logarithms, wave 2: the sine, the orientation bit and the leading 1
    const int wv = tid >> 6, ln = tid & 63;
    auto forced = [&](int j) -> int {  // the anchor whose regression row ground truth j writes, or 0
      if ((T.colmax[j] >> 63) != 0ull) return 0;
      return (int)(T.colwin[j] >> 32);
    };
    auto gval = [&](int j, int k) -> double {  // k: x, y, z, w, l, h, yaw
      if (gt_staged) return T.gtv[j][k];
      return k < 3 ? t.g_centers[j * 3 + k] : k < 6 ? t.g_wlh[j * 3 + k - 3] : t.g_yaw[j];
    };
    auto aval = [&](int i, int k) -> double {
      if (t.grid) {
        const AnchorId id = anchor_id(t, i);
        if (k < 2) return k == 0 ? id.cx : id.cy;
        const int col = k == 2 ? 12 : k == 6 ? 11 : 5 + k;  // z | yaw | w, l, h
        return ty_staged ? T.types[id.d][col] : t.types[id.d * kTypeCols + col];
      }
      return k < 3 ? t.a_centers[(int64_t)i * 3 + k] : k < 6 ? t.a_wlh[(int64_t)i * 3 + k - 3] : t.a_yaw[i];
    };
    if (wv < 2) {
      for (int it = ln; it < G * 3; it += 64) {
        const int j = it / 3, c = it - j * 3;
        const int i = forced(j);
        if (i == 0) continue;
        float *reg = t.reg_targets + (int64_t)i * 9;
        if (wv == 0)
          reg[1 + c] = target_quotient(gval(j, c), aval(i, c), aval(i, 3), aval(i, 4), aval(i, 5), t.canvas_height, c);
        else
          reg[4 + c] = target_logratio(gval(j, 3 + c), aval(i, 3 + c));
      }
    } else if (wv == 2) {
      for (int j = ln; j < G; j += 64) {
        const int i = forced(j);
        if (i == 0) continue;
        float *reg = t.reg_targets + (int64_t)i * 9;
        float dt, ort;
        target_angle(gval(j, 6), aval(i, 6), &dt, &ort);
        reg[0] = 1.0f;
        reg[7] = dt;
        reg[8] = ort;
      }
    }
  } else {
    // phase A: clear the class rows of all forced anchors
    for (int j = tid; j < G; j += kTgtThreads) {
      const int i = (int)(cwin_ld(j) >> 32);
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
      const u64 w = cwin_ld(j);
      const int i = (int)(w >> 32);
      if (i == 0) continue;
      if ((unsigned)t.g_class[j] < (unsigned)t.num_classes)  // a class outside the row is ignored, never written
        t.cls_targets[(int64_t)i * t.num_classes + t.g_class[j]] = 1.0f;
      bool later = false;
      for (int j2 = j + 1; j2 < G; ++j2) later = later || ((int)(cwin_ld(j2) >> 32) == i);
      if (!later) write_row(i, j);
    }
  }
  IOU_STAMP(14);
  // re-arm the scratch words for the next call on this context
  __syncthreads();
  if (!IN_LDS)
    for (int j = tid; j < G; j += kTgtThreads) {
      t.col_max[j] = 0ull;
      t.col_win[j] = ~0ull;
    }
  if (tid == 0) {
    *t.cand_count = 0u;
    *t.ticket = 0u;
    t.ticket[1] = 0u;  // (the box-centric form counts in 64 bits: groups done | pairs above the threshold)
  }
}

// MATRIX: make_ious (data/pillars.cpp:400-427) on the same gate / queue / clip machinery -- the
// pairs past the gate are written into the host-zeroed [A][G] matrix, nothing else is computed.
#ifdef PP_TGT_WAVES  // development knob (tools/lab): waves per SIMD the register allocation is held to
#define PP_TGT_OCC __attribute__((amdgpu_waves_per_eu(PP_TGT_WAVES, PP_TGT_WAVES)))
#else
#define PP_TGT_OCC
#endif
template <bool MATRIX>
__global__ __launch_bounds__(kTgtThreads) PP_TGT_OCC void k_targets(TargetArgs t, TargetBatch bt) {
  // Batch launches are (sample, tile of 256 anchors): workgroups are dispatched x first, so the samples
  // advance side by side -- with the samples one after the other the last one's workgroups all started
  // late, its heavy ones last, and the launch ended ~10 us after everything else had drained.
  // ... and centre-out over the tiles (dispatch position k -> tile mid, mid+1, mid-1, ...): anchors are laid out row
  // by row, so the tiles dispatched LAST are the canvas' first and last rows -- where boxes are rarest (objects
  // gather around the ego vehicle, the centre of the canvas; the synthetic boxes keep a margin) -- and a launch
  // whose last workgroups are the cheap ones drains sooner (C3 B=4: see profiles/r04/NOTES.md).
  const unsigned nwg = MATRIX ? gridDim.x : gridDim.y;
  unsigned tile = blockIdx.x;
  if constexpr (!MATRIX) {
    const unsigned k = blockIdx.y, mid = nwg >> 1, up = mid + ((k + 1u) >> 1);
    // k odd: mid + (k+1)/2, k even: mid - k/2; once one side is used up, the rest of the other side in order
    if (k & 1u)
      tile = up < nwg ? up : nwg - 1u - k;            // upper side exhausted: what is left below, downwards
    else
      tile = (k >> 1) <= mid ? mid - (k >> 1) : k;    // lower side exhausted: what is left above, upwards
  }
  if constexpr (!MATRIX) sample_view(t, bt, (int)blockIdx.x);
  __shared__ __align__(16) unsigned char smem[kTgtLdsBytes];
  TgtLds &S = *reinterpret_cast<TgtLds *>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int v = lane & (kGroup - 1), gbase = lane & ~(kGroup - 1);
  const int64_t i0 = (int64_t)tile * kTgtThreads;
  const int64_t i = i0 + tid;
  const bool live = i < t.A;
  IOU_STAMP(0);
  // The first chunk's ground truths and the type table are on their way while the arithmetic below
  // runs.  Every wave fetches the chunk's centres for itself (lane = ground truth): the "is any box
  // near this workgroup" decision below then needs no LDS and no barrier.
  double2 pre_c = make_double2(0.0, 0.0), pre_k = make_double2(0.0, 0.0);
  int pre_cls = 0;
  auto load_chunk = [&](int j0, int gn) {
    if (!MATRIX && tid < gn) pre_cls = t.g_class[j0 + tid];  // (a positive's class used to be a load of its own, late)
    if (lane < gn) {
      const double *gp = t.g_centers_img + (int64_t)(j0 + lane) * t.g_center_cols;
      pre_c = make_double2(gp[0], gp[1]);
    }
    if (tid < gn * 4) {  // kGtChunk * 4 == kTgtThreads: one corner per thread
      const double *gp = t.g_corners + ((int64_t)j0 * 4 + tid) * 2;
      pre_k = make_double2(gp[0], gp[1]);
    }
  };
  static_assert(kGtChunk * 4 <= kTgtThreads, "one corner per thread");
  const int gn0 = min(kGtChunk, t.G);
  load_chunk(0, gn0);
  const bool single = t.G <= kGtChunk;
  const bool lds_types = t.grid && t.per_cell <= kLdsTypes;
  double pre_ty = 0.0;
  static_assert(kLdsTypes * kTypeCols <= kTgtThreads, "one table entry per thread");
  if (lds_types && tid < t.per_cell * kTypeCols) pre_ty = t.types[tid];
  double acx = 0, acy = 0;
  if (!t.grid && live) {
    acx = t.a_centers[i * t.a_center_cols];
    acy = t.a_centers[i * t.a_center_cols + 1];
  }
  // Bounding box of the workgroup's anchor centres: a ground truth further than the gate's 10
  // (+1: rounding of the gate's subtraction, whatever the magnitudes) from it is far from every
  // anchor here, and most are -- the gate loop runs over the few that are left.
  double bx0, bx1, by0, by1;
  if (t.grid) {
    const unsigned pc = (unsigned)t.per_cell, fw = (unsigned)t.fm_w;
    const unsigned c0 = (unsigned)i0 / pc, c1 = (unsigned)(min(i0 + kTgtThreads, t.A) - 1) / pc;
    const unsigned y0 = c0 / fw, y1 = c1 / fw;
    const unsigned x0 = (y0 == y1) ? c0 - y0 * fw : 0u, x1 = (y0 == y1) ? c1 - y1 * fw : fw - 1u;
    bx0 = cell_centre(t, x0);
    bx1 = cell_centre(t, x1);
    by0 = cell_centre(t, y0);
    by1 = cell_centre(t, y1);
  } else {
    double mnx = live ? acx : INFINITY, mxx = live ? acx : -INFINITY;
    double mny = live ? acy : INFINITY, mxy = live ? acy : -INFINITY;
    const bool odd = live && !(acx == acx && acy == acy);  // a NaN centre passes every gate
    mnx = wave_minmax_f64<false>(mnx);
    mxx = wave_minmax_f64<true>(mxx);
    mny = wave_minmax_f64<false>(mny);
    mxy = wave_minmax_f64<true>(mxy);
    if (lane == 0) {
      const bool any_odd = __ballot(odd) != 0ull;
      S.bbox[wv][0] = any_odd ? -INFINITY : mnx;
      S.bbox[wv][1] = any_odd ? INFINITY : mxx;
      S.bbox[wv][2] = any_odd ? -INFINITY : mny;
      S.bbox[wv][3] = any_odd ? INFINITY : mxy;
    }
    __syncthreads();
    bx0 = S.bbox[0][0], bx1 = S.bbox[0][1], by0 = S.bbox[0][2], by1 = S.bbox[0][3];
#pragma unroll
    for (int w = 1; w < kTgtWaves; ++w) {
      bx0 = fmin(bx0, S.bbox[w][0]);
      bx1 = fmax(bx1, S.bbox[w][1]);
      by0 = fmin(by0, S.bbox[w][2]);
      by1 = fmax(by1, S.bbox[w][3]);
    }
  }
  const double fx0 = bx0 - 11.0, fx1 = bx1 + 11.0, fy0 = by0 - 11.0, fy1 = by1 + 11.0;
  // No box of the sample near this workgroup (a third of them at BASELINE config 3; every wave works
  // the same answer out from the same registers): nothing below concerns it -- zero rows, a ticket, done.
  const bool near0 = lane < gn0 && !(pre_c.x < fx0 || pre_c.x > fx1 || pre_c.y < fy0 || pre_c.y > fy1);
  const bool skip = single && __ballot(near0) == 0ull;
  if (MATRIX && skip) return;
  if (!skip) {
    if (live) {
      int d = 0;
      if (t.grid) {
        const AnchorId id = anchor_id(t, i);
        acx = id.cx;
        acy = id.cy;
        d = id.d;
      }
      S.acen[tid] = make_double2(acx, acy);
      S.atype[tid] = (unsigned short)d;
    }
    if (lds_types && tid < t.per_cell * kTypeCols) S.types[0][tid] = pre_ty;
  }
  double best = 0.0;  // np.max over a row that is all zeros is 0, argmax 0
  int best_j = 0, best_c = -1;
  bool bad = false;
  // this workgroup's columns of a chunk -> the list (at most one entry per ground truth)
  // wave 0: reserve list slots for the chunk's touched columns (one counter bump), write them later
  auto reserve_columns = [&](int gn, u64 &tb, unsigned &base) {
    const bool touched = lane < gn && S.cmax[lane] != 0ull;
    tb = __ballot(touched);
    base = 0u;
    if (tb && lane == 0) base = atomicAdd(t.cand_count, (unsigned)__popcll(tb));
  };
  auto write_columns = [&](int j0, u64 tb, unsigned base) {
    if (!tb) return;
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    if ((tb >> lane) & 1ull) {
      const unsigned pos = base + (unsigned)__popcll(tb & ((1ull << lane) - 1ull));
      ColEntry *ce = cand_at(t, pos);
      __hip_atomic_store(&ce->key, (u64)(unsigned)(j0 + lane) | ((u64)(unsigned)S.carg[lane] << 32), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&ce->bits, S.cmax[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto append_columns = [&](int j0, int gn) -> bool {
    if (wv != 0) return false;
    u64 tb;
    unsigned base;
    reserve_columns(gn, tb, base);
    write_columns(j0, tb, base);
    return tb != 0ull;
  };
  int last_j0 = 0, last_gn = 0;  // the last chunk's columns are appended behind the row stores
  bool touched_before = false;   // wave 0: an earlier chunk had a column with IoU > 0
  for (int j0 = 0; j0 < (skip ? 0 : t.G); j0 += kGtChunk) {
    const int gn = min(kGtChunk, t.G - j0);
    if (j0 > 0) load_chunk(j0, gn);
    __syncthreads();
    if (tid < gn) {  // wave 0's copy
      S.gc[tid] = pre_c;
      S.gcls[tid] = pre_cls;
      S.cmax[tid] = 0ull;
      S.cseen[tid] = 0ull;
      S.carg[tid] = INT_MAX;
    }
    if (tid < gn * 4) S.gk[tid >> 2][tid & 3] = pre_k;
    __syncthreads();
    IOU_STAMP(1);
    // the chunk's ground truths near this workgroup (every wave works out the same mask)
    bool near = false;
    if (lane < gn) {
      const double2 gcn = S.gc[lane];
      near = !(gcn.x < fx0 || gcn.x > fx1 || gcn.y < fy0 || gcn.y > fy1);
    }
    const u64 nb = __ballot(near);
    if (nb == 0ull) continue;
    if (near && wv == 0) {  // read after the next barrier
      double g8[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        g8[2 * k] = S.gk[lane][k].x;
        g8[2 * k + 1] = S.gk[lane][k].y;
      }
      S.garea[lane] = -shoelace_dev(g8, 4);
    }
    // gate: the survivors of this chunk as a mask
    u64 mask = 0;
    if (live)
      for (u64 m = nb; m; m &= m - 1) {
        const int j = __ffsll((long long)m) - 1;
        mask |= (u64)(gate_far(acx, acy, S.gc[j].x, S.gc[j].y) ? 0 : 1) << j;
      }
    const int cnt = __popcll(mask);
    // exclusive prefix sum of the counts over the workgroup (wave scans + wave totals)
    const int inc = (int)wave_scan_u32((unsigned)cnt);
    if (lane == 63) S.woff[wv] = inc;
    __syncthreads();
    int wave_base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kTgtWaves; ++w) {
      if (w < wv) wave_base += S.woff[w];
      total += S.woff[w];
    }
    const int my_off = wave_base + inc - cnt;
    IOU_STAMP(2);
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
      IOU_STAMP(3);
      // clip: 32 groups of 8 lanes, one pair each per round
      // (a wave whose eight groups have no pair in this round sits it out: a round is ~500 instructions,
      // and a third of the wave-rounds were empty)
      static_assert(kMaxNP == 1, "one pair per group and round");
      for (int r0 = 0; r0 < wn; r0 += kPairsPerRound)
        if (r0 + wv * (64 / kGroup) < wn) clip_round<1>(t, S, r0, wn, i0, tid, v, gbase, lds_types, bad);
      __syncthreads();
      IOU_STAMP(4);
      if constexpr (MATRIX) {
        if (t.triples) {
          for (int q = tid; q < wn; q += kTgtThreads) {
            const double val = S.iou[q];
            if (val != 0.0) {  // (a zero is what the host's fill already wrote)
              const unsigned at = atomicAdd(t.triple_count, 1u);
              if (at < t.triple_cap) t.triples[at] = IouTriple{(unsigned)(i0 + S.pair_lane[q]), (unsigned)(j0 + S.pair_gt[q]), val};
            }
          }
        } else {
          for (int q = tid; q < wn; q += kTgtThreads)
            t.ious[(i0 + S.pair_lane[q]) * t.G + j0 + S.pair_gt[q]] = S.iou[q];
        }
        __syncthreads();
        continue;
      }
      // rows: this anchor's slice of the window, ascending gt; strict >: first maximum wins
      if (cnt > 0) {
        const int qa = max(my_off, wb), qb = min(my_off + cnt, wb + wn);
        for (int q = qa; q < qb; ++q) {
          const double val = S.iou[q - wb];
          if (val > best) {
            best = val;
            best_j = j0 + S.pair_gt[q - wb];
            best_c = S.gcls[S.pair_gt[q - wb]];
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
    if constexpr (MATRIX) continue;
    if (j0 + kGtChunk < t.G) {
      touched_before = append_columns(j0, gn) || touched_before;
    } else {
      last_j0 = j0;
      last_gn = gn;
    }
  }
  IOU_STAMP(5);
  if (bad) atomicOr(t.errflag, kErrWinding);
  if constexpr (MATRIX) return;
  // rows: positives of both targets, zero rows otherwise
  const int nc = t.num_classes;
  const int nrows = (int)min((int64_t)kTgtThreads, t.A - i0);
  const bool pos = live && best > t.pos_thresh;  // box_utils.py:195 (strict >)
  const int cj = pos ? best_c : -1;
  float *cls_dst = t.cls_targets + i0 * nc, *reg_dst = t.reg_targets + i0 * 9;
  // Does the tail depend on this workgroup at all?  It reads the list entries and overwrites forced
  // rows, and a forced anchor has IoU > 0 with its ground truth: only a workgroup with a touched
  // column appends anything or owns a row the tail may write.
  bool contrib = false;
  u64 pb = 0ull, last_tb = 0ull;
  unsigned last_base = 0u;
  if (!skip) {
    if (wv == 0) {
      // the last chunk's list slots: the counter's round trip runs under the row phase
      reserve_columns(last_gn, last_tb, last_base);
      if (lane == 0) S.contrib = (touched_before || last_tb != 0ull) ? 1 : 0;
    }
    // the positives, lined up for the groups of 8 lanes that work their regression rows out
    pb = __ballot(pos);
    if (lane == 0) S.woff[wv] = __popcll(pb);
    __syncthreads();
    contrib = S.contrib != 0;
  }
  // Every other workgroup (half of them at BASELINE config 3) takes its ticket NOW -- no store of
  // its own to order, so no drain -- and the counter's round trip runs under its row stores.
  unsigned tk = 0u;
  unsigned *my_ticket1 = t.ticket1 + (size_t)(tile >> t.ticket_shift) * kTicketPad;
  if (!kStrictHandoff && !contrib && t.G != 0 && tid == 0) tk = atomicAdd(my_ticket1, 1u);
  int pbase = 0, npos = 0;
  if (!skip) {
#pragma unroll
    for (int w = 0; w < kTgtWaves; ++w) {
      if (w < wv) pbase += S.woff[w];
      npos += S.woff[w];
    }
  }
  const bool aligned = (((uintptr_t)t.cls_targets | (uintptr_t)t.reg_targets) & 15) == 0;
  if (npos == 0 && aligned) {
    // no positive anchor here (nine workgroups in ten): zero rows straight from registers
    store_zero_rows(cls_dst, nrows * nc, tid);
    store_zero_rows(reg_dst, nrows * 9, tid);
  } else {
    if (pos) {
      const int k = pbase + __popcll(pb & ((1ull << lane) - 1ull));
      S.pair_lane[k] = (unsigned short)tid;
      S.pair_gt[k] = (unsigned short)best_j;
    }
    const bool cls_staged = nc <= kStageCols && ((uintptr_t)t.cls_targets & 15) == 0;
    if (live) {
      if (cls_staged)
        for (int c = 0; c < nc; ++c) S.stage[tid * nc + c] = (c == cj) ? 1.0f : 0.0f;
#pragma unroll
      for (int d = 0; d < 9; ++d) S.rstage[tid * 9 + d] = 0.0f;
    }
    __syncthreads();
    if (cls_staged) {
      store_rows(cls_dst, S.stage, nrows * nc, tid);
    } else if (live) {
      for (int c = 0; c < nc; ++c) store_f32_sc1(&cls_dst[(int64_t)tid * nc + c], (c == cj) ? 1.0f : 0.0f);
    }
    // The positives' regression rows, one KIND of value per wave (a wave executes every branch its lanes take:
    // eight lanes per row, one value each, made every wave pay for a square root, divisions, a logarithm and
    // a sine per 32 rows): wave 0 the sine and the orientation bit, wave 1 two logarithms (independent chains:
    // they interleave), wave 2 the third and the quotients; lane = positive.
    if (wv < 3) {
      for (int k = lane; k < npos; k += 64) {
        const int pl = S.pair_lane[k], pj = S.pair_gt[k];
        double ax, ay, az, aw, al, ah, ayaw;
        if (lds_types) {  // centre and type row are in LDS already
          const double *ty = S.types[S.atype[pl]];
          ax = S.acen[pl].x, ay = S.acen[pl].y, az = ty[12];
          aw = ty[8], al = ty[9], ah = ty[10], ayaw = ty[11];
        } else {
          const BoxVals a = anchor_vals(t, i0 + pl);
          ax = a.x, ay = a.y, az = a.z, aw = a.w, al = a.l, ah = a.h, ayaw = a.yaw;
        }
        float *row = S.rstage + pl * 9;
        auto gval = [&](int k) -> double {  // k: x, y, z, w, l, h, yaw
          return k < 3 ? t.g_centers[pj * 3 + k] : k < 6 ? t.g_wlh[pj * 3 + k - 3] : t.g_yaw[pj];
        };
        if (wv == 0) {
          float dt, ort;
          target_angle(gval(6), ayaw, &dt, &ort);
          row[0] = 1.0f;
          row[7] = dt;
          row[8] = ort;
        } else if (wv == 1) {
          row[4] = target_logratio(gval(3), aw);
          row[5] = target_logratio(gval(4), al);
        } else {
          row[6] = target_logratio(gval(5), ah);
          row[1] = target_quotient(gval(0), ax, aw, al, ah, t.canvas_height, 0);
          row[2] = target_quotient(gval(1), ay, aw, al, ah, t.canvas_height, 1);
          row[3] = target_quotient(gval(2), az, aw, al, ah, t.canvas_height, 2);
        }
      }
    }
    __syncthreads();
    if (((uintptr_t)t.reg_targets & 15) == 0) {
      store_rows(reg_dst, S.rstage, nrows * 9, tid);
    } else if (live) {
#pragma unroll
      for (int d = 0; d < 9; ++d) store_f32_sc1(&reg_dst[(int64_t)tid * 9 + d], S.rstage[tid * 9 + d]);
    }
  }
  IOU_STAMP(6);
  if (t.G == 0) return;
  if (contrib || kStrictHandoff) {
    if (contrib && wv == 0) write_columns(last_j0, last_tb, last_base);
    // The last workgroup to get here finishes the job.  Every store above that the tail depends on
    // is write-through; drained per wave, then one agent-scope add per workgroup: the workgroup
    // whose add comes last reads the others' entries with sc1 loads and may overwrite their rows.
    IOU_STAMP(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    IOU_STAMP(8);
    if (tid == 0) {
      handoff_release_lane0();
      tk = atomicAdd(my_ticket1, 1u);
    }
  }
  if (tid == 0) {
    // the last workgroup of its group of 1 << ticket_shift carries the group's ticket on
    const unsigned grp = tile >> t.ticket_shift, ngrp = ((nwg - 1u) >> t.ticket_shift) + 1u;
    const unsigned gsize = min(1u << t.ticket_shift, nwg - (grp << t.ticket_shift));
    int last = 0;
    if (tk == gsize - 1u) {
      __hip_atomic_store(my_ticket1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next call
      last = (atomicAdd(t.ticket, 1u) == ngrp - 1u) ? 1 : 0;
    }
    S.is_last = last;
  }
  __syncthreads();
  IOU_STAMP(9);
  if (!S.is_last) return;
  handoff_acquire_tail();
  if (t.G <= kForcedLds)
    targets_tail<true>(t, *reinterpret_cast<TailLds *>(smem));
  else
    targets_tail<false>(t, *reinterpret_cast<TailLds *>(smem));
  IOU_STAMP(10);
}

// ------------------------------------------------------------------------- //
// k_targets_gt: create_target, box-centric (anchors on the fly)               //
// ------------------------------------------------------------------------- //
// k_targets walks the ANCHORS: 489 workgroups per sample at BASELINE config 3, every one paying ~2 us of
// staging before it knows whether a box is anywhere near, the pairs spread thinly over them (21 of 32 slots
// of a clip round filled), and the launch drains behind whichever workgroup happens to hold 100 pairs.  With
// the anchors a regular grid (make_anchor_boxes, box_utils.py:111-159) the pairs can be enumerated from the
// BOX side instead: the anchors within the +-10 centre gate (pillars.cpp:418-419) of box j are the cells
// x0..x1 x y0..y1 around its centre -- computed one cell generous, then put through the SAME gate test on
// the SAME centre values, so the set of pairs is exactly make_ious' -- about 200 per box at config 3.
//   ZERO role   kZeroWgs workgroups per sample stream zeros over both target arrays (99.9 % of the bytes),
//               a plain grid-stride fill;
//   PAIR role   gt_splits workgroups per box, kCandPerWg candidate anchors each: gate, queue, clip with 8
//               lanes per pair (clip_round: the code k_targets runs), then
//                 - the box's column maximum and the first anchor reaching it among this workgroup's
//                   pairs -> ONE list entry (the tail reduces a box's entries as it does k_targets'),
//                 - every pair above the threshold -> the sample's positive list, and an atomic maximum
//                   on best[anchor] (row maximum: np.max(ious, axis=1) can only exceed the threshold
//                   through such a pair);
//   tail        the sample's last workgroup (two-level ticket over ZERO and PAIR workgroups alike: the
//               zeros must be down before a row is written): a positive-list entry wins its anchor when it
//               holds best[anchor] and, among equals, the lowest box index (np.argmax: first maximum) --
//               class and regression rows of the winners -- then k_targets' tail: column argmax, forced rows.
// Results equal k_targets' bit for bit (same gate, same clip, same row expressions; tests run both forms
// against each other and the oracle).
#ifndef PP_CAND_PER_WG
#define PP_CAND_PER_WG 64
#endif
constexpr int kCandPerWg = PP_CAND_PER_WG;   // candidate anchors per PAIR workgroup (<= 64): one wave gates, four clip
static_assert(kCandPerWg >= 8 && kCandPerWg <= 64, "one lane of wave 0 per candidate");
constexpr int kZeroWgs = 64;     // ZERO workgroups per sample (x 4 samples = a hipMemset-shaped grid)

// `n` zero floats to dst (any 4-byte alignment, n * 4 < 2^31), this workgroup's share of a grid-stride fill by
// `nwg` workgroups: ONE buffer resource for the array (wave-uniform: a per-lane base would make every store a
// 64-trip waterfall loop), the lane's place in it as the 32-bit offset
__device__ __forceinline__ void zero_share(float *dst, int64_t n, int wg, int nwg, int tid) {
  const int head = (int)min((int64_t)((16 - ((uintptr_t)dst & 15)) & 15) / 4, n);  // floats before 16-byte alignment
  if (wg == 0 && tid < head) store_f32_sc1(dst + tid, 0.0f);
  const int n4 = (int)((n - head) >> 2);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(dst + head), 0, n4 * 16, 0x00020000);
  const v4u z = {0u, 0u, 0u, 0u};
  for (int k = wg * kTgtThreads + tid; k < n4; k += nwg * kTgtThreads) __builtin_amdgcn_raw_buffer_store_b128(z, rs, k * 16, 0, kAuxSc1);
  const int64_t done = head + 4 * (int64_t)n4;
  if (wg == 0 && tid < (int)(n - done)) store_f32_sc1(dst + done + tid, 0.0f);
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, o));
  return v;
}

// The winners of the positive list: rows of the anchors whose row maximum exceeds the threshold
// (box_utils.py:193-196, 211, 219-221).  Runs in the sample's last workgroup, before the forced rows.
// An entry wins its anchor when it holds the anchor's highest IoU and, among equals, the lowest box index
// (np.argmax: first maximum).  Up to kPosLds entries (all real scenes: ~130 at BASELINE config 3) are resolved
// in an LDS hash table keyed by the anchor; beyond that through the per-anchor words in global memory that the
// PAIR role keeps current (correct, slower: three passes of dependent loads).
constexpr int kPosLds = 1024;            // entries resolved in LDS
constexpr int kPosHash = 2 * kPosLds;    // table slots (a power of two)
struct PosLds {
  unsigned key[kPosHash];   // anchor + 1 (0 = empty)
  u64 bits[kPosHash];       // the anchor's highest IoU
  unsigned minj[kPosHash];  // the first box reaching it
};
static_assert(sizeof(PosLds) <= kTgtLdsBytes, "the positives' stage reuses the workgroup's LDS");

__device__ void positives_tail(const TargetArgs &t, unsigned char *smem) {
  PosLds &W = *reinterpret_cast<PosLds *>(smem);
  const int tid = threadIdx.x;
  const unsigned n_raw = __hip_atomic_load(t.pos_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned cap = (unsigned)t.G * t.pos_per_gt;
  const unsigned n = min(n_raw, cap);
  if (n_raw > cap && tid == 0) atomicOr(t.errflag, kErrPosOverflow);  // only with a box centre that is not finite on
                                                                       // ONE axis (its window is a whole band of the map)
  if (n == 0) return;
  constexpr int kPer = kPosLds / kTgtThreads;
  const bool in_lds = n <= (unsigned)kPosLds;
  // the entries (other XCDs wrote them: sc1 loads), all in flight together
  unsigned e_i[kPer], e_j[kPer];
  u64 e_b[kPer];
  bool e_win[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const unsigned e = (unsigned)(k * kTgtThreads + tid);
    e_i[k] = e_j[k] = 0u;
    e_b[k] = 0ull;
    e_win[k] = false;
    if (in_lds && e < n) {
      const u64 ij = ld_agent(reinterpret_cast<const u64 *>(&t.pos[e]));
      e_i[k] = (unsigned)ij;
      e_j[k] = (unsigned)(ij >> 32);
      e_b[k] = ld_agent(&t.pos[e].bits);
    }
  }
  if (in_lds) {
    for (int h = tid; h < kPosHash; h += kTgtThreads) {
      W.key[h] = 0u;
      W.bits[h] = 0ull;
      W.minj[h] = ~0u;
    }
    __syncthreads();
    int slot[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      slot[k] = -1;
      if ((unsigned)(k * kTgtThreads + tid) < n) {
        unsigned h = (e_i[k] * 2654435761u) >> (32 - 11);
        static_assert(kPosHash == 2048, "hash width");
        for (;;) {  // the table is at most half full
          const unsigned old = atomicCAS(&W.key[h], 0u, e_i[k] + 1u);
          if (old == 0u || old == e_i[k] + 1u) break;
          h = (h + 1u) & (kPosHash - 1);
        }
        slot[k] = (int)h;
        atomicMax(&W.bits[h], e_b[k]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; ++k)
      if (slot[k] >= 0 && e_b[k] == W.bits[slot[k]]) atomicMin(&W.minj[slot[k]], e_j[k]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; ++k) e_win[k] = slot[k] >= 0 && e_b[k] == W.bits[slot[k]] && e_j[k] == W.minj[slot[k]];
#pragma unroll
    for (int k = 0; k < kPer; ++k)
      if (e_win[k]) {
        // class row: a one at the box's class (box_utils.py:211); the row is zero already
        const int c = t.g_class[e_j[k]];
        if ((unsigned)c < (unsigned)t.num_classes) t.cls_targets[(int64_t)e_i[k] * t.num_classes + c] = 1.0f;
        // regression row (box_utils.py:219-221): the PAIR workgroup that clipped the pair worked it out
        const float *src = t.pos_rows[k * kTgtThreads + tid].row;
        float *reg = t.reg_targets + (int64_t)e_i[k] * 9;
        float rv[9];
#pragma unroll
        for (int d = 0; d < 9; ++d) rv[d] = __hip_atomic_load(src + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int d = 0; d < 9; ++d) reg[d] = rv[d];
      }
    __syncthreads();
  } else {
    // more pairs above the threshold than the table holds: per-anchor words in global memory (armed to "none", only
    // this workgroup touches them: atomics and agent-scope loads meet in its XCD's L2).  Round 5: this rare path raises
    // the anchors' maxima itself -- the PAIR role used to, with one more global atomic per pair above the threshold in
    // every launch, and every tail had to reset them.
    for (unsigned e = tid; e < n; e += kTgtThreads) {
      const unsigned i = (unsigned)ld_agent(reinterpret_cast<const u64 *>(&t.pos[e]));
      __hip_atomic_fetch_max(&t.best[i], ld_agent(&t.pos[e].bits), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
    __syncthreads();
    for (unsigned e0 = 0; e0 < n; e0 += kTgtThreads) {
      const unsigned e = e0 + tid;
      if (e < n) {
        const u64 ij = ld_agent(reinterpret_cast<const u64 *>(&t.pos[e]));
        const unsigned i = (unsigned)ij, j = (unsigned)(ij >> 32);
        if (ld_agent(&t.pos[e].bits) == ld_agent(&t.best[i])) atomicMin(&t.bestj[i], j);
      }
    }
    __threadfence();
    __syncthreads();
    for (unsigned e0 = 0; e0 < n; e0 += kTgtThreads) {
      const unsigned e = e0 + tid;
      if (e < n) {
        const u64 ij = ld_agent(reinterpret_cast<const u64 *>(&t.pos[e]));
        const unsigned i = (unsigned)ij, j = (unsigned)(ij >> 32);
        if (ld_agent(&t.pos[e].bits) == ld_agent(&t.best[i]) &&
            j == __hip_atomic_load(&t.bestj[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          const int c = t.g_class[j];
          if ((unsigned)c < (unsigned)t.num_classes) t.cls_targets[(int64_t)i * t.num_classes + c] = 1.0f;
          float *reg = t.reg_targets + (int64_t)i * 9;
#pragma unroll
          for (int d = 0; d < 9; ++d)
            reg[d] = __hip_atomic_load(t.pos_rows[e].row + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    __threadfence();
    __syncthreads();
    for (unsigned e = tid; e < n; e += kTgtThreads) {
      const unsigned i = (unsigned)ld_agent(reinterpret_cast<const u64 *>(&t.pos[e]));
      __hip_atomic_store(&t.best[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&t.bestj[i], ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tid == 0) *t.pos_count = 0u;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the winners' rows are down before a forced row overwrites one
  __syncthreads();
}

// The box-centric tail in its usual shape -- at most 64 boxes, 1024 list slots, 256 pairs above the threshold,
// 63 classes -- which the caller knows BEFORE anything is loaded: the number of pairs above the threshold arrives with
// the last ticket (the tickets count in 64 bits: workgroups done | the pairs they stored).
// Round 5, what the stamps said about round 4's form at one sample per launch (tools/lab/gt_stamps.py): last ticket ->
// entries in registers 2.0 us (18 dword loads per thread with the lanes 64 bytes apart: a wave-level load touched 64
// lines and every line was asked for a dozen times), LDS phases 0.8, the forced rows' fetch and the drain in front of
// it 1.2, end 0.4.  And what a first rebuild showed (every record's row kept in the registers of the lanes that
// loaded it, twelve unrolled 16-byte groups per lane: SLOWER, 7 us in its last phase): the tail is code that ONE
// workgroup executes ONCE per launch, on a CU that has never run it -- it runs at the speed its instructions arrive,
// so it has to be short.  Hence
//   * keys and rows live in separate arrays (PosKey / PosRow, ColEntry / cand_rows), thread e <-> record e: one
//     16-byte load per key (a wave-level load is 1024 consecutive bytes), three per row of a pair; every load of the
//     first round trip is issued before anything is waited for;
//   * the two LDS phases (row maximum per anchor through a hash table + the first box reaching it; column maximum per
//     box + the first anchor reaching it and its slot);
//   * then wave 0, lane = box, fetches the forced rows from the winning slots (the one dependent round trip left;
//     loading ALL slots' rows with the first round trip instead was measured: +0.6 us there, nothing gained here) and,
//     while they are on their way, lets the boxes that force the same anchor meet in the SAME hash table (class mask by
//     atomicOr, the last box by atomicMax: round 4's ballot-per-index-bit search was 1.3 us of once-executed code) --
//     which also tells the positives that this anchor's rows are the forced ones: a positive's row is simply not
//     written where a forced row goes, so no store has to be down before another one;
//   * winners' rows go out from registers; nothing is evaluated here.
constexpr int kFastPos = 256, kFastHash = 512, kFastSlots = 4;  // slots per thread
struct FastTailLds {
  unsigned key[kFastHash];     // anchor + 1 (0 = empty): anchors with a pair above the threshold, and forced anchors
  u64 bits[kFastHash];         // the anchor's highest IoU
  unsigned minj[kFastHash];    // the first box reaching it
  u64 fmask[kFastHash];        // classes of ALL boxes forcing this anchor (0: no box forces it)
  int flast[kFastHash];        // the last box forcing it
  u64 colmax[64], colwin[64];
};
static_assert(sizeof(FastTailLds) <= kTgtLdsBytes, "the fast tail reuses the workgroup's LDS");

__device__ __forceinline__ unsigned fast_hash_slot(FastTailLds &F, unsigned anchor) {  // find or insert
  unsigned h = (anchor * 2654435761u) >> (32 - 9);
  static_assert(kFastHash == 512, "hash width");
  for (;;) {  // at most 256 + 64 of the 512 slots are ever taken
    const unsigned old = atomicCAS(&F.key[h], 0u, anchor + 1u);
    if (old == 0u || old == anchor + 1u) return h;
    h = (h + 1u) & (kFastHash - 1);
  }
}

__device__ void tail_gt_fast(const TargetArgs &t, unsigned char *smem, int nslots, unsigned n) {
  FastTailLds &F = *reinterpret_cast<FastTailLds *>(smem);
  const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6, G = t.G;
  // ---- the first round trip: every load (a record beyond its array reads zeros)
  __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void *)t.pos, 0, (int)(n * 16u), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void *)t.pos_rows, 0, (int)(n * 48u), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void *)t.cand, 0, nslots * 16, 0x00020000);
  const v4u pk = __builtin_amdgcn_raw_buffer_load_b128(rs_k, tid * 16, 0, kAuxSc1);  // {anchor, box, IoU}
  v4u ck[kFastSlots];                                                                   // {box, anchor, IoU}
#pragma unroll
  for (int k = 0; k < kFastSlots; ++k) ck[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, (k * kTgtThreads + tid) * 16, 0, kAuxSc1);
  v4u pr[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) pr[c] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, tid * 48 + c * 16, 0, kAuxSc1);
  unsigned gc = (tid < G) ? (unsigned)t.g_class[tid] : 63u;  // wave 0, lane = box: its class
  gc = gc < 63u ? gc : 63u;                                  // (63: a class outside the row, never written)
  // ---- LDS phase 0
  for (int h = tid; h < kFastHash; h += kTgtThreads) {
    F.key[h] = 0u;
    F.bits[h] = 0ull;
    F.minj[h] = ~0u;
    F.fmask[h] = 0ull;
    F.flast[h] = -1;
  }
  if (tid < 64) {
    F.colmax[tid] = 0ull;
    F.colwin[tid] = ~0ull;
  }
  __syncthreads();
  IOU_STAMP(11);
  // ---- phase 1: maxima (rows: per anchor through the hash table; columns: per box)
  const u64 pbits = (u64)pk[2] | ((u64)pk[3] << 32);
  int slot = -1;
  if ((unsigned)tid < n) {
    slot = (int)fast_hash_slot(F, pk[0]);
    atomicMax(&F.bits[slot], pbits);
  }
#pragma unroll
  for (int k = 0; k < kFastSlots; ++k) {
    const u64 b = (u64)ck[k][2] | ((u64)ck[k][3] << 32);
    if (b != 0ull) atomicMax(&F.colmax[(int)ck[k][0]], b);
  }
  __syncthreads();
  IOU_STAMP(12);
  // ---- phase 2: first box reaching a row's maximum; first anchor reaching a column's, and its list slot
  if (slot >= 0 && pbits == F.bits[slot]) atomicMin(&F.minj[slot], pk[1]);
#pragma unroll
  for (int k = 0; k < kFastSlots; ++k) {
    const u64 b = (u64)ck[k][2] | ((u64)ck[k][3] << 32);
    const int j = (int)ck[k][0];
    if (b != 0ull && b == F.colmax[j]) atomicMin(&F.colwin[j], ((u64)ck[k][1] << 32) | (unsigned)(k * kTgtThreads + tid));
  }
  __syncthreads();
  IOU_STAMP(13);
  // ---- forced rows (box_utils.py:199-205, 212-213, 223-228), wave 0, lane = box: the row is fetched from the slot the
  // column's argmax came in, and while it is on its way the boxes that force the same anchor meet in the hash table
  // (the classes of all of them, the last of them: its regression row wins) -- which also tells the positives
  // below that this anchor's rows are the forced ones
  int i = 0, fslot = 0;
  v4u fr[3] = {v4u{0u, 0u, 0u, 0u}, v4u{0u, 0u, 0u, 0u}, v4u{0u, 0u, 0u, 0u}};
  if (wv == 0) {
    const u64 w = (ln < G && F.colmax[ln] != 0ull) ? F.colwin[ln] : 0ull;
    i = (int)(w >> 32);  // the anchor box ln forces; 0: none (an argmax of 0 is dropped, :204-205)
    if (i != 0) {
      __amdgpu_buffer_rsrc_t rs_f = __builtin_amdgcn_make_buffer_rsrc((void *)t.cand_rows, 0, nslots * kRowPitch * 4, 0x00020000);
      const int off = (int)(unsigned)(w & 0xFFFFFFFFull) * (kRowPitch * 4);
#pragma unroll
      for (int c = 0; c < 3; ++c) fr[c] = __builtin_amdgcn_raw_buffer_load_b128(rs_f, off + c * 16, 0, kAuxSc1);
      fslot = (int)fast_hash_slot(F, (unsigned)i);
      atomicOr(&F.fmask[fslot], 1ull << gc);
      atomicMax(&F.flast[fslot], ln);
    }
  }
  __syncthreads();
  IOU_STAMP(14);
  // ---- the positives' rows (box_utils.py:211, 219-221), from registers -- unless a box forces the anchor
  if (slot >= 0) {
    if (pbits == F.bits[slot] && pk[1] == F.minj[slot] && F.fmask[slot] == 0ull) {
      const unsigned cl = pr[2][1];
      if (cl < (unsigned)t.num_classes) t.cls_targets[(int64_t)pk[0] * t.num_classes + cl] = 1.0f;
      float *reg = t.reg_targets + (int64_t)pk[0] * 9;
#pragma unroll
      for (int d = 0; d < 9; ++d) reg[d] = __uint_as_float(pr[d >> 2][d & 3]);
    }
  }
  if (i != 0) {  // (wave 0)
    const u64 mask = F.fmask[fslot];
    float *cls = t.cls_targets + (int64_t)i * t.num_classes;
    for (int c = 0; c < t.num_classes; ++c) cls[c] = ((mask >> c) & 1ull) ? 1.0f : 0.0f;  // every duplicate: the same row
    if (F.flast[fslot] == ln) {
      float *reg = t.reg_targets + (int64_t)i * 9;
#pragma unroll
      for (int d = 0; d < 9; ++d) reg[d] = __uint_as_float(fr[d >> 2][d & 3]);
    }
  }
  if (tid == 0) {  // re-armed for the next call on this context
    *t.pos_count = 0u;
    *t.cand_count = 0u;
    *reinterpret_cast<u64 *>(t.ticket) = 0ull;
  }
}

__global__ __launch_bounds__(kTgtThreads) PP_TGT_OCC void k_targets_gt(TargetArgs t, TargetBatch bt) {
  sample_view(t, bt, (int)blockIdx.x);
  __shared__ __align__(16) unsigned char smem[kTgtLdsBytes];
  TgtLds &S = *reinterpret_cast<TgtLds *>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int v = lane & (kGroup - 1), gbase = lane & ~(kGroup - 1);
  const int G = t.G, nsp = t.gt_splits;
  const unsigned u = blockIdx.y, n_units = (unsigned)kZeroWgs + (unsigned)G * (unsigned)nsp;
  if (u >= n_units) return;
  unsigned npos_wg = 0u;
  IOU_STAMP(0);
  if (u < (unsigned)kZeroWgs) {
    zero_share(t.cls_targets, t.A * t.num_classes, (int)u, kZeroWgs, tid);
    zero_share(t.reg_targets, t.A * 9, (int)u, kZeroWgs, tid);
    if (G == 0) return;  // no box: nothing else runs for this sample, no tail
  } else {
    // Dispatch order (workgroups start in block-id order): the windows' MIDDLE bands first, for all boxes, then outwards.
    // The candidates are enumerated row-major over the box's window, so the middle splits hold the pairs above the
    // threshold -- the workgroups with the longest chain (list atomics, entries) -- and the sample's tail waits for
    // the last of them.  The slot of the column list stays j * nsp + split.
    const int rank = ((int)u - kZeroWgs) / G, j = ((int)u - kZeroWgs) - rank * G;
    const int mid = (nsp - 1) >> 1, dst = (rank + 1) >> 1;
    const int split = (rank & 1) ? mid + dst : mid - dst;
    const int pu = j * nsp + split;
    const bool lds_types = t.per_cell <= kLdsTypes;
    // The prologue's global loads, ALL issued before any is waited for.  (A launch starts on cold caches: a load is
    // ~1 us, and a wave that branches by lane role with a load inside each branch pays that once per branch -- the
    // first version of this prologue spent 2.2 us there.)  Wave 1: the box (lane = value), wave 2 / 3: the anchor
    // type table, and the operands of the logarithms (wave 2) / the sine (wave 3); every lane straight from global
    // memory, no lane waits for another one's LDS store.
    double ld0 = 0.0, ld1 = 0.0, rc_g = 0.0, rc_a = 1.0;
    int ldi = 0;
    if (wv == 1) {
      const double *p0 = nullptr;
      if (lane < 8) p0 = t.g_corners + (int64_t)j * 8 + lane;          // corners x0, y0 .. x3, y3
      else if (lane < 11) p0 = t.g_centers + j * 3 + (lane - 8);       // the box's values for the rows: x, y, z,
      else if (lane < 14) p0 = t.g_wlh + j * 3 + (lane - 11);          // w, l, h,
      else if (lane == 14) p0 = t.g_yaw + j;                            // yaw
      else if (lane >= 32 && lane < 32 + t.per_cell) p0 = t.types + (lane - 32) * kTypeCols + 8;  // a type's w ...
      if (p0) ld0 = *p0;
      if (lane >= 32 && lane < 32 + t.per_cell) ld1 = t.types[(lane - 32) * kTypeCols + 9];       // ... and l
      if (lane == 15) ldi = t.g_class[j];
    } else if (wv >= 2) {
      const int k2 = tid - 2 * 64;  // 0..127
      if (k2 < t.per_cell * kTypeCols) ld0 = t.types[k2];
      if (wv == 2 && lane < t.per_cell * 3) {
        const int d = lane / 3, k = lane - d * 3;
        rc_g = t.g_wlh[j * 3 + k], rc_a = t.types[d * kTypeCols + 8 + k];
      } else if (wv == 3 && lane < t.per_cell) {
        rc_g = t.g_yaw[j], rc_a = t.types[lane * kTypeCols + 11];
      }
    }
    // The box's image-space centre: every thread holds it (the candidate window, the gate).  Everything else the
    // workgroup needs of the box is loaded by waves 1-3 WHILE wave 0 gates the first window (round 5; the box's
    // loads and the gate used to be two phases with a barrier and a serial area in between: 2.2 us of the
    // workgroup's 6.4), each lane straight from global memory -- no lane waits for another one's LDS store.
    const double gcx = t.g_centers_img[(int64_t)j * t.g_center_cols], gcy = t.g_centers_img[(int64_t)j * t.g_center_cols + 1];
    // candidate cells: cell_centre(x) = (x + .5) / fm_scale within 10 of the box centre <=> x in [a, a + 20 * fm_scale]
    // with a = (gcx - 10) * fm_scale - .5; the window is that range in exact arithmetic plus ONE cell either side
    // (the gate below decides on the centre values themselves); a centre that is not finite passes the gate
    // with every anchor (NaN compares false): the whole map
    int x0 = 0, x1 = t.fm_w - 1, y0 = 0, y1 = t.fm_h - 1;
    if (gcx - gcx == 0.0) {
      x0 = max(x0, (int)fmax(ceil((gcx - 10.0) * t.fm_scale - 0.5) - 1.0, -1.0e9));
      x1 = min(x1, (int)fmin(floor((gcx + 10.0) * t.fm_scale - 0.5) + 1.0, 1.0e9));
    }
    if (gcy - gcy == 0.0) {
      y0 = max(y0, (int)fmax(ceil((gcy - 10.0) * t.fm_scale - 0.5) - 1.0, -1.0e9));
      y1 = min(y1, (int)fmin(floor((gcy + 10.0) * t.fm_scale - 0.5) + 1.0, 1.0e9));
    }
    const int ncx = max(x1 - x0 + 1, 0), ncy = max(y1 - y0 + 1, 0);
    const unsigned ncand = (unsigned)ncx * (unsigned)ncy * (unsigned)t.per_cell;  // <= A < 2^24
    u64 col_bits = 0ull;          // wave 0: this workgroup's column maximum for box j ...
    // (npos_wg, below: the pairs above the threshold this workgroup stored; rides on its ticket)
    unsigned col_anchor = ~0u;    // ... and the first anchor reaching it
    bool bad = false;
    const unsigned cpw = (unsigned)t.cand_per_wg;
    IOU_STAMP(1);
    if (wv == 1) {  // the box into LDS slot 0 (the loads were issued at the top)
      if (lane < 8) reinterpret_cast<double *>(S.gk[0])[lane] = ld0;
      else if (lane < 15) S.gv7[lane - 8] = ld0;
      else if (lane == 15) S.gcls[0] = ldi;
      else if (lane >= 32 && lane < 32 + t.per_cell) S.rc_ad[lane - 32] = sqrt(ld0 * ld0 + ld1 * ld1);  // box_utils.py:79
      iou_wave_sync();
      if (lane == 0) {
        double g8[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          g8[2 * k] = S.gk[0][k].x;
          g8[2 * k + 1] = S.gk[0][k].y;
        }
        S.garea[0] = -shoelace_dev(g8, 4);
      }
    } else if (wv >= 2) {
      const int k2 = tid - 2 * 64;
      if (k2 < t.per_cell * kTypeCols) S.types[0][k2] = ld0;
      // What the regression rows need per anchor TYPE: the logarithms and the sine of a row depend on (box, type)
      // only (box_utils.py:88-102) -- f64 library chains of ~1 us, evaluated once per type here, side by side with
      // wave 0's gate, instead of once per pair after the clip; a pair's row then costs three subtractions and
      // divisions.  (Behind the gate's barrier, side by side with the clip, was measured too: 0.5 us slower -- waves 2
      // and 3 clip as well.)
      if (wv == 2 && lane < t.per_cell * 3) {
        S.rc_lg[lane / 3][lane % 3] = target_logratio(rc_g, rc_a);
      } else if (wv == 3 && lane < t.per_cell) {
        float dt, ort;
        target_angle(rc_g, rc_a, &dt, &ort);
        S.rc_dt[lane] = dt;
        S.rc_ort[lane] = ort;
      }
    }
    bool first = true;
    for (unsigned q0 = (unsigned)split * cpw; q0 < ncand || first; q0 += (unsigned)nsp * cpw) {
      if (!first) __syncthreads();  // the previous trip's LDS is read
      int wn = 0;
      if (wv == 0) {
        const unsigned q = q0 + (unsigned)lane;
        bool pass = false;
        if ((unsigned)lane < cpw && q < ncand) {
          const unsigned cell = q / (unsigned)t.per_cell, d = q - cell * (unsigned)t.per_cell;
          const unsigned yy = cell / (unsigned)ncx, xx = cell - yy * (unsigned)ncx;
          const unsigned x = (unsigned)x0 + xx, y = (unsigned)y0 + yy;
          const double acx = cell_centre(t, x), acy = cell_centre(t, y);
          S.acen[lane] = make_double2(acx, acy);
          S.atype[lane] = (unsigned short)d;
          S.carg[lane] = (int)((y * (unsigned)t.fm_w + x) * (unsigned)t.per_cell + d);  // the anchor's index
          pass = !gate_far(acx, acy, gcx, gcy);
        }
        const u64 pm = __ballot(pass);
        if (pass) {
          const int k = __popcll(pm & ((1ull << lane) - 1ull));
          S.pair_lane[k] = (unsigned short)lane;
          S.pair_gt[k] = 0;
        }
        wn = __popcll(pm);
        if (lane == 0) S.woff[0] = wn;
      }
      __syncthreads();
      wn = S.woff[0];
      IOU_STAMP(2);
      first = false;
      for (int r0 = 0; r0 < wn; r0 += kPairsPerRound)
        if (r0 + wv * (64 / kGroup) < wn) clip_round<1>(t, S, r0, wn, 0, tid, v, gbase, lds_types, bad);
      __syncthreads();
      IOU_STAMP(4);
#ifdef PP_IOU_CLIP_TWICE  // tools/lab: the clip once more (same result) -- what the SECOND pass through the same code takes
      for (int r0 = 0; r0 < wn; r0 += kPairsPerRound)
        if (r0 + wv * (64 / kGroup) < wn) clip_round<1>(t, S, r0, wn, 0, tid, v, gbase, lds_types, bad);
      __syncthreads();
      IOU_STAMP(5);
#endif
      // wave 0, lane = pair: the column (maximum, then the lowest anchor index among the pairs that reach it)
      // and the pairs above the threshold
      double val = 0.0;
      unsigned ai = ~0u, pos_base = 0u;
      bool pos = false, newcol = false;
      int col_lane = -1;
      u64 pb = 0ull;
      if (wv == 0) {
        val = lane < wn ? S.iou[lane] : 0.0;
        ai = lane < wn ? (unsigned)S.carg[S.pair_lane[lane]] : ~0u;
        const double wmax = wave_minmax_f64<true>(val > 0.0 ? val : 0.0);
        if (wmax > 0.0) {
          const u64 wb = (u64)__double_as_longlong(wmax);
          const unsigned firsta = wave_min_u32(val == wmax ? ai : ~0u);
          if (wb > col_bits || (wb == col_bits && firsta < col_anchor)) {
            newcol = true;
            col_lane = __ffsll((long long)__ballot(val == wmax && ai == firsta)) - 1;
            col_anchor = firsta;
            col_bits = wb;
          }
        }
        pos = val > t.pos_thresh;
        pb = __ballot(pos);
        const u64 need = pb | (newcol ? 1ull << col_lane : 0ull);
        if (lane == 0) {
          S.need = need;
          // the pairs' places in the sample's list: asked for NOW, the answer is needed after the rows
          if (pb) pos_base = atomicAdd(t.pos_count, (unsigned)__popcll(pb));
        }
        npos_wg += (unsigned)__popcll(pb);
      }
      __syncthreads();
      // The regression rows of those pairs (box_utils.py:70-109), lane = pair.  With the per-type values in LDS:
      // wave c < 3 the quotient of coordinate c (one division chain per wave, all three side by side) and its share
      // of the constants.  (More than kLdsTypes anchor types per cell never get here: the host sends them through the
      // anchor-centric kernel -- the per-pair library chains inside this loop cost every launch 46 VGPRs.)
      const u64 need = S.need;
      if (need && wv < 3 && ((need >> lane) & 1ull)) {
        const int pl = S.pair_lane[lane];
        float *row = S.rstage + lane * 9;
        {
          const int d = S.atype[pl];
          const double *ty = S.types[d];
          const double a_c = wv == 0 ? S.acen[pl].x : wv == 1 ? S.acen[pl].y : ty[12];
          row[1 + wv] = target_quotient_ad(S.gv7[wv], a_c, S.rc_ad[d], ty[10], t.canvas_height, wv);
          if (wv == 0) {
            row[0] = 1.0f;
            row[7] = S.rc_dt[d];
            row[8] = S.rc_ort[d];
          } else if (wv == 1) {
            row[4] = S.rc_lg[d][0];
            row[5] = S.rc_lg[d][1];
          } else {
            row[6] = S.rc_lg[d][2];
          }
        }
      }
      if (need) __syncthreads();  // (workgroup-uniform)
      if (wv == 0) {
        if (newcol && lane < 9) S.colrow[lane] = S.rstage[col_lane * 9 + lane];
        if (pb) {
          const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)pos_base);
          if (pos) {
            const unsigned at = base + (unsigned)__popcll(pb & ((1ull << lane) - 1ull));
            const u64 bits = (u64)__double_as_longlong(val);
            if (at < (unsigned)G * t.pos_per_gt) {
              // the pair's key {anchor, box, IoU} and its row {row[0..8], class}: four 16-byte write-through stores
              __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)t.pos, 0, 0x7FFFFFFF, 0x00020000);
              __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *)t.pos_rows, 0, 0x7FFFFFFF, 0x00020000);
              const float *r9 = S.rstage + lane * 9;
              const v4u c0 = {ai, (unsigned)j, (unsigned)bits, (unsigned)(bits >> 32)};
              const v4u c1 = {__float_as_uint(r9[0]), __float_as_uint(r9[1]), __float_as_uint(r9[2]), __float_as_uint(r9[3])};
              const v4u c2 = {__float_as_uint(r9[4]), __float_as_uint(r9[5]), __float_as_uint(r9[6]), __float_as_uint(r9[7])};
              const v4u c3 = {__float_as_uint(r9[8]), (unsigned)S.gcls[0], 0u, 0u};  // the box's class rides along
              __builtin_amdgcn_raw_buffer_store_b128(c0, rk, (int)at * 16, 0, kAuxSc1);
              __builtin_amdgcn_raw_buffer_store_b128(c1, rr, (int)at * 48, 0, kAuxSc1);
              __builtin_amdgcn_raw_buffer_store_b128(c2, rr, (int)at * 48 + 16, 0, kAuxSc1);
              __builtin_amdgcn_raw_buffer_store_b128(c3, rr, (int)at * 48 + 32, 0, kAuxSc1);
            }
          }
        }
      }
    }
    if (bad) atomicOr(t.errflag, kErrWinding);
    if (wv == 0 && lane < 4) {
      // this workgroup's slot of the column list {box, anchor, IoU} and the pair's row, written whether or not a pair
      // overlapped (bits 0 = none): no counter, nothing stale, and the tail knows the list's length without a load.
      // Lane 0 the entry, lanes 1-3 one 16-byte group of the row each: ONE write-through store per lane.
      if (lane == 0) {
        const v4u ch = {(unsigned)j, col_bits ? col_anchor : 0u, (unsigned)col_bits, (unsigned)(col_bits >> 32)};
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)t.cand, 0, 0x7FFFFFFF, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(ch, rs, pu * 16, 0, kAuxSc1);
      } else if (col_bits != 0ull) {
        const float *cr = S.colrow + 4 * (lane - 1);
        const v4u ch = {__float_as_uint(cr[0]), __float_as_uint(cr[1]), __float_as_uint(cr[2]), __float_as_uint(cr[3])};
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)t.cand_rows, 0, 0x7FFFFFFF, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(ch, rs, pu * (kRowPitch * 4) + (lane - 1) * 16, 0, kAuxSc1);
      }
    }
  }
  // every store and atomic above is down before the ticket; the sample's last workgroup finishes the job
  IOU_STAMP(6);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  IOU_STAMP(8);
  if (tid == 0) {
    handoff_release_lane0();
    // two-level ticket, 64 bits a word: low half = workgroups (groups) done, high half = the pairs above the threshold
    // they stored -- the sample's last workgroup gets the length of the pair list with its ticket, not from a load
    u64 *my_ticket1 = reinterpret_cast<u64 *>(t.ticket1 + (size_t)(u >> t.ticket_shift) * kTicketPad);
    const unsigned grp = u >> t.ticket_shift, ngrp = ((n_units - 1u) >> t.ticket_shift) + 1u;
    const unsigned gsize = min(1u << t.ticket_shift, n_units - (grp << t.ticket_shift));
    const u64 mine = 1ull | ((u64)min(npos_wg, 65535u) << 32);  // (<= 65535 workgroups x 65535: no carry out of the word)
    int last = 0;
    unsigned ntot = ~0u;
    const u64 old1 = atomicAdd(my_ticket1, mine);
    if ((unsigned)old1 == gsize - 1u) {
      __hip_atomic_store(my_ticket1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const u64 grp_pos = (old1 + mine) & 0xFFFFFFFF00000000ull;
      const u64 old2 = atomicAdd(reinterpret_cast<u64 *>(t.ticket), 1ull | grp_pos);
      if ((unsigned)old2 == ngrp - 1u) {
        last = 1;
        ntot = (unsigned)((old2 + grp_pos) >> 32);
      }
    }
    S.is_last = last;
    S.contrib = (int)ntot;
  }
  __syncthreads();
  IOU_STAMP(9);
  if (!S.is_last) return;
  handoff_acquire_tail();
  const unsigned n_pairs = (unsigned)S.contrib;  // pairs above the threshold in the sample's list
  if (t.G <= 64 && t.G * nsp <= kFastSlots * kTgtThreads && t.num_classes <= 63 && n_pairs <= (unsigned)kFastPos &&
      n_pairs <= (unsigned)t.G * t.pos_per_gt) {
    tail_gt_fast(t, smem, t.G * nsp, n_pairs);
    IOU_STAMP(10);
    return;
  }
  positives_tail(t, smem);
  IOU_STAMP(15);
  if (t.G <= kForcedLds)
    targets_tail<true>(t, *reinterpret_cast<TailLds *>(smem), t.G * nsp);
  else
    targets_tail<false>(t, *reinterpret_cast<TailLds *>(smem), t.G * nsp);
  IOU_STAMP(10);
}


__global__ void k_targets_init(u64 *col_max, u64 *col_win, int G, int *errflag,
                               unsigned *cand_count, unsigned *ticket, unsigned *ticket1, int n_ticket1) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n_ticket1) ticket1[j] = 0u;
  if (j < G) {
    col_max[j] = 0ull;
    col_win[j] = ~0ull;
  }
  if (j < PP_MAX_BATCH) {  // every sample's counter and ticket
    cand_count[j * kCounterStride] = 0u;
    ticket[j * kCounterStride] = 0u;
    ticket[j * kCounterStride + 1] = 0u;  // (k_targets_gt counts in 64 bits)
  }
  if (j == 0) *errflag = 0;
}

__global__ void k_targets_gt_init(u64 *best, unsigned *bestj, int64_t n, unsigned *pos_count) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) {
    best[k] = 0ull;
    bestj[k] = ~0u;
  }
  if (k < PP_MAX_BATCH) pos_count[k * kCounterStride] = 0u;
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

// make_ious' launch: the dense matrix (ious_dev), or -- triples != NULL -- the entries that are not zero as records
// behind *triple_count (zeroed here)
static int launch_make_ious(pp_ctx_t *ctx, hipStream_t stream, const double *a_corners_dev, const double *a_centers_dev,
                            int64_t a_center_cols, int64_t A, const double *g_corners_dev, const double *g_centers_dev,
                            int64_t g_center_cols, int64_t G, double *ious_dev, IouTriple *triples,
                            unsigned *triple_count, unsigned triple_cap) {
  DeviceGuard2 guard(ctx->device);
  int rc = ctx->iou_ws.ensure(4096);
  if (rc) return rc;
  int *errflag = static_cast<int *>(ctx->iou_ws.ptr);
  PP_HIP_TRY(hipMemsetAsync(errflag, 0, 4, stream));
  if (triples) {
    PP_HIP_TRY(hipMemsetAsync(triple_count, 0, 4, stream));
  } else {
    // every entry is written (pillars.cpp:421,424): zeros by this fill, the 0.16 % of the pairs that
    // pass the centre gate by the kernel behind it on the same stream
    PP_HIP_TRY(hipMemsetAsync(ious_dev, 0, (size_t)A * (size_t)G * 8, stream));
  }
  TargetArgs t{};
  t.A = A;
  t.G = (int)G;
  t.a_corners = a_corners_dev;
  t.a_centers = a_centers_dev;
  t.a_center_cols = (int)a_center_cols;
  t.g_center_cols = (int)g_center_cols;
  t.g_corners = g_corners_dev;
  t.g_centers_img = g_centers_dev;
  t.ious = ious_dev;
  t.triples = triples;
  t.triple_count = triple_count;
  t.triple_cap = triple_cap;
  t.errflag = errflag;
  const unsigned nwg = (unsigned)((A + kTgtThreads - 1) / kTgtThreads);
  TargetBatch bt{};
  bt.g_off[1] = (int)G;
  hipLaunchKernelGGL(k_targets<true>, dim3(nwg), dim3(kTgtThreads), 0, stream, t, bt);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}

extern "C" int pp_make_ious_dev(pp_ctx_t *ctx, void *stream_, const double *a_corners_dev,
                                const double *a_centers_dev, int64_t a_center_cols, int64_t A,
                                const double *g_corners_dev, const double *g_centers_dev,
                                int64_t g_center_cols, int64_t G, double *ious_dev) {
  if (!ctx) {
    set_error("ctx is NULL");
    return PP_ERR_VALUE;
  }
  if (A < 0 || G < 0 || G > INT_MAX / 16 || A > (1ll << 38) || a_center_cols < 2 || a_center_cols > 64 ||
      g_center_cols < 2 || g_center_cols > 64) {
    set_error("pp_make_ious_dev: bad sizes (A=%lld G=%lld)", (long long)A, (long long)G);
    return PP_ERR_VALUE;
  }
  if (A == 0 || G == 0) return PP_OK;
  if (!a_corners_dev || !a_centers_dev || !g_corners_dev || !g_centers_dev || !ious_dev) {
    set_error("pp_make_ious_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  return launch_make_ious(ctx, static_cast<hipStream_t>(stream_), a_corners_dev, a_centers_dev, a_center_cols, A,
                          g_corners_dev, g_centers_dev, g_center_cols, G, ious_dev, nullptr, nullptr, 0);
}

// make_ious' host drop-in: the record count and the error word into device-visible host memory behind the kernel (one
// wait instead of three copies with a wait each); the error word is cleared once it has been handed over, like
// pp_iou_check does
__global__ void k_iou_tail(const unsigned *count, int *flag, unsigned *host_out) {
  if (threadIdx.x == 0) {
    const int f = *flag;
    host_out[0] = *count;
    host_out[1] = (unsigned)f;
    if (f) *flag = 0;
  }
}

static int iou_flag_result(pp_ctx_t *ctx, int flag);

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
  }
  return iou_flag_result(ctx, flag);
}

static int iou_flag_result(pp_ctx_t *ctx, int flag) {
  if (flag) {
    if (flag & kErrPosOverflow) {  // (takes precedence: the scratch has to be re-armed whatever else happened)
      // the positive list overflowed (sized from the candidate count, so only reachable through boxes whose centre is
      // not finite): entries were dropped; re-arm every scratch word before the next call
      std::memset(ctx->tgt_key, 0, sizeof ctx->tgt_key);
      set_error("target assignment: more pairs above the threshold than candidate anchors (a box centre that is not finite?)");
      return PP_ERR_VALUE;
    }
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
  static const bool trace = getenv("PP_DROPIN_TRACE") != nullptr;  // development knob: where a call's time goes
  auto t_prev = std::chrono::steady_clock::now();
  double t_us[5] = {0, 0, 0, 0, 0};
  auto lap = [&](int k) {
    if (!trace) return;
    const auto now = std::chrono::steady_clock::now();
    t_us[k] += std::chrono::duration<double, std::micro>(now - t_prev).count();
    t_prev = now;
  };
  // pinned staging: the ground truths [G][8] + [G][2] (per call); the anchors [A][8] + [A][2] in a mirror of their own
  const size_t a_bytes = (size_t)A * 10 * 8, g_bytes = (size_t)G * 10 * 8;
  int rc = ctx->pin_in.ensure(g_bytes);
  if (rc) return rc;
  rc = ctx->stage_in.ensure(g_bytes);
  if (rc) return rc;
  bool fresh = ctx->anchors_A != A || !ctx->anchors_pin.ptr || !ctx->anchors_dev.ptr;
  if (fresh) ctx->anchors_A = -1;  // (an allocation failure below leaves nothing half-valid)
  rc = ctx->anchors_pin.ensure(a_bytes);
  if (rc) return rc;
  rc = ctx->anchors_dev.ensure(a_bytes);
  if (rc) return rc;
  // The signature demands every entry of the caller's [A,G] matrix written (pillars.cpp:421,424) -- 40 MB at BASELINE
  // config 3, of which ~8 000 entries are not zero.  Sparse form: the device returns those entries as 16-byte records
  // (128 kB over PCIe instead of 40 MB), the pool's threads zero the caller's matrix WHILE the anchors travel and the
  // kernel runs, then the records are written over the zeros.  More records than the list holds (every anchor near
  // every box): the dense form below, as before.
  const size_t cells = (size_t)A * (size_t)G;
  const unsigned cap = (unsigned)std::min<size_t>(std::max<size_t>(cells / 16, 1u << 16), 8u << 20);  // <= 128 MB
  const bool sparse = A < (1ll << 32) && cells >= 4096;
  HostPool *pool = host_pool(ctx);
  double *h_ac = static_cast<double *>(ctx->anchors_pin.ptr), *h_an = h_ac + A * 8;
  double *h_gc = static_cast<double *>(ctx->pin_in.ptr), *h_gn = h_gc + G * 8;
  auto rd = [](const void *base, int64_t off) {
    double v;
    std::memcpy(&v, static_cast<const char *>(base) + off, 8);
    return v;
  };
  const bool ac_dense = ac[2] == 8 && ac[1] == 16 && ac[0] == 64;
  std::atomic<int> changed{fresh ? 1 : 0};
  pool->run([&](int part, int parts) {  // the anchors' rows, split across the threads; compared (as bits) while copied
    const int64_t i0 = A * part / parts, i1 = A * (part + 1) / parts;
    int diff = 0;
    auto put = [&](double *slot, double v) {
      if (std::memcmp(slot, &v, 8) != 0) {
        *slot = v;
        diff = 1;
      }
    };
    for (int64_t i = i0; i < i1; ++i) {
      if (ac_dense) {
        const char *src = static_cast<const char *>(a_corners) + i * 64;
        if (std::memcmp(h_ac + i * 8, src, 64) != 0) {
          std::memcpy(h_ac + i * 8, src, 64);
          diff = 1;
        }
      } else {
        for (int k = 0; k < 4; ++k)
          for (int c = 0; c < 2; ++c) put(&h_ac[i * 8 + k * 2 + c], rd(a_corners, i * ac[0] + k * ac[1] + c * ac[2]));
      }
      put(&h_an[i * 2], rd(a_centers, i * an[0]));
      put(&h_an[i * 2 + 1], rd(a_centers, i * an[0] + an[1]));
    }
    if (diff) changed.store(1, std::memory_order_relaxed);
  });
  for (int64_t j = 0; j < G; ++j) {
    for (int k = 0; k < 4; ++k)
      for (int c = 0; c < 2; ++c) h_gc[j * 8 + k * 2 + c] = rd(g_corners, j * gc[0] + k * gc[1] + c * gc[2]);
    h_gn[j * 2] = rd(g_centers, j * gn[0]);
    h_gn[j * 2 + 1] = rd(g_centers, j * gn[0] + gn[1]);
  }
  lap(0);
  double *d_a = static_cast<double *>(ctx->anchors_dev.ptr), *d_g = static_cast<double *>(ctx->stage_in.ptr);
  if (changed.load()) {
    ctx->anchors_A = -1;  // (the mirror is ahead of the device until the copy is enqueued: a failure here must not leave
                          // the next call comparing against a mirror the device never received)
    PP_HIP_TRY(hipMemcpyAsync(d_a, h_ac, a_bytes, hipMemcpyHostToDevice, stream));
    ctx->anchors_A = A;  // (the copy is ordered before every launch below on this stream, and the call ends synchronised)
  }
  PP_HIP_TRY(hipMemcpyAsync(d_g, h_gc, g_bytes, hipMemcpyHostToDevice, stream));
  char *dst = static_cast<char *>(ious);
  const bool out_dense = io[1] == 8 && io[0] == G * 8;
  if (sparse) {
    rc = ctx->stage_out2.ensure(256 + (size_t)cap * 16);
    if (rc) return rc;
    rc = ctx->pin_meta.ensure(256);
    if (rc) return rc;
    char *so = static_cast<char *>(ctx->stage_out2.ptr);
    unsigned *count_dev = reinterpret_cast<unsigned *>(so);
    IouTriple *triples_dev = reinterpret_cast<IouTriple *>(so + 256);
    unsigned *count_host = static_cast<unsigned *>(ctx->pin_meta.ptr);
    // PP_DROPIN_DIRECT (any bit; pp_create_pillars_f64 shares the knob): the kernel appends its records straight into
    // device-visible pinned memory -- each is one 16-byte store of one lane, ~8 000 of them -- and a one-thread kernel
    // behind it leaves the count and the error word there too: the call waits for the device ONCE (before: the count's
    // copy, the records' copy and pp_iou_check's copy, a wait each).  Lists beyond 16 MB keep the copies.
    static const bool k_direct = [] { const char *e = getenv("PP_DROPIN_DIRECT"); return e ? atoi(e) != 0 : true; }();
    void *tr_map = nullptr, *cnt_map = nullptr;
    bool direct = k_direct && (size_t)cap * 16 <= (16u << 20) && ctx->pin_out.ensure((size_t)cap * 16) == PP_OK &&
                  hipHostGetDevicePointer(&tr_map, ctx->pin_out.ptr, 0) == hipSuccess && tr_map &&
                  hipHostGetDevicePointer(&cnt_map, count_host, 0) == hipSuccess && cnt_map;
    (void)hipGetLastError();  // (a refused mapping is not an error of this call: the copies are used)
    rc = launch_make_ious(ctx, stream, d_a, d_a + A * 8, 2, A, d_g, d_g + G * 8, 2, G, nullptr,
                          direct ? static_cast<IouTriple *>(tr_map) : triples_dev, count_dev, cap);
    if (rc) return rc;
    if (direct) {
      hipLaunchKernelGGL(k_iou_tail, dim3(1), dim3(64), 0, stream, count_dev, static_cast<int *>(ctx->iou_ws.ptr),
                         static_cast<unsigned *>(cnt_map));
      PP_HIP_TRY(hipGetLastError());
    } else {
      PP_HIP_TRY(hipMemcpyAsync(count_host, count_dev, 4, hipMemcpyDeviceToHost, stream));
    }
    // ... meanwhile: zeros into the caller's matrix (rows split across the workers)
    const std::function<void(int, int)> zero_rows = [&](int part, int parts) {
      const int64_t i0 = A * part / parts, i1 = A * (part + 1) / parts;
      if (out_dense) {
        if (i1 > i0) std::memset(dst + i0 * G * 8, 0, (size_t)(i1 - i0) * G * 8);
      } else {
        const double zero = 0.0;
        for (int64_t i = i0; i < i1; ++i)
          for (int64_t j = 0; j < G; ++j) std::memcpy(dst + i * io[0] + j * io[1], &zero, 8);
      }
    };
    pool->start(zero_rows);
    const hipError_t e_sync = hipStreamSynchronize(stream);
    pool->wait();
    PP_HIP_TRY(e_sync);
    lap(1);
    const unsigned count = count_host[0];
    if (direct && count_host[1]) return iou_flag_result(ctx, (int)count_host[1]);  // (a wrong winding: the matrix is zeros)
    if (count <= cap) {
      if (count) {
        if (!direct) {
          rc = ctx->pin_out.ensure((size_t)count * 16);
          if (rc) return rc;
          PP_HIP_TRY(hipMemcpyAsync(ctx->pin_out.ptr, triples_dev, (size_t)count * 16, hipMemcpyDeviceToHost, stream));
          PP_HIP_TRY(hipStreamSynchronize(stream));
        }
        const IouTriple *tr = static_cast<const IouTriple *>(ctx->pin_out.ptr);
        for (unsigned k = 0; k < count; ++k)
          std::memcpy(dst + (int64_t)tr[k].anchor * io[0] + (int64_t)tr[k].box * io[1], &tr[k].iou, 8);
      }
      lap(2);
      if (trace)
        fprintf(stderr, "pp_make_ious_f64: gather + compare %.0f us (anchors %s) | H2D + kernel (host zero fill alongside, %d threads) %.0f | %u records back + written %.0f\n",
                t_us[0], changed.load() ? "uploaded" : "resident", pool->size(), t_us[1], count, t_us[2]);
      return direct ? PP_OK : pp_iou_check(ctx, stream);
    }
    // the list overflowed: the anchors are on the device already, take the dense form
  }
  rc = ctx->stage_out2.ensure(cells * 8);
  if (rc) return rc;
  rc = ctx->pin_out.ensure(cells * 8);
  if (rc) return rc;
  rc = pp_make_ious_dev(ctx, stream, d_a, d_a + A * 8, 2, A, d_g, d_g + G * 8, 2, G,
                        static_cast<double *>(ctx->stage_out2.ptr));
  if (rc) return rc;
  PP_HIP_TRY(hipMemcpyAsync(ctx->pin_out.ptr, ctx->stage_out2.ptr, cells * 8, hipMemcpyDeviceToHost, stream));
  PP_HIP_TRY(hipStreamSynchronize(stream));
  // every entry is written (pillars.cpp:421,424)
  const double *src = static_cast<const double *>(ctx->pin_out.ptr);
  pool->run([&](int part, int parts) {
    const int64_t i0 = A * part / parts, i1 = A * (part + 1) / parts;
    if (out_dense) {
      if (i1 > i0) std::memcpy(dst + i0 * G * 8, src + i0 * G, (size_t)(i1 - i0) * G * 8);
    } else {
      for (int64_t i = i0; i < i1; ++i)
        for (int64_t j = 0; j < G; ++j) std::memcpy(dst + i * io[0] + j * io[1], &src[i * G + j], 8);
    }
  });
  return pp_iou_check(ctx, stream);
}

struct AnchorSource {
  const double *corners = nullptr, *centers = nullptr, *wlh = nullptr, *yaw = nullptr;
  int grid = 0, fm_w = 0, fm_h = 0, per_cell = 0;
  double fm_scale = 1.0;
  const double *types = nullptr;
};

static int assign_targets_impl(pp_ctx_t *ctx, void *stream_, int batch, const int32_t *g_counts,
                               int64_t A, const AnchorSource &an, const double *g_corners,
                               const double *g_centers_img, const double *g_centers, const double *g_wlh,
                               const double *g_yaw, const int32_t *g_class, const pp_target_params_t *prm,
                               float *cls_targets, float *reg_targets) {
  if (!ctx || !prm || !g_counts) {
    set_error("pp_assign_targets*_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  if (batch < 1 || batch > PP_MAX_BATCH) {
    set_error("pp_assign_targets*_dev: batch %d outside 1..%d", batch, PP_MAX_BATCH);
    return PP_ERR_VALUE;
  }
  TargetBatch bt{};
  int64_t g_total = 0;
  for (int b = 0; b < batch; ++b) {
    if (g_counts[b] < 0 || g_counts[b] > 65535) {
      set_error("pp_assign_targets*_dev: sample %d has %d ground truths (0..65535)", b, (int)g_counts[b]);
      return PP_ERR_VALUE;
    }
    g_total += g_counts[b];
    bt.g_off[b + 1] = (int)g_total;
  }
  if (A < 1 || A > 65535ll * kTgtThreads || prm->num_classes < 1 || prm->num_classes > 1024 ||
      A * std::max(prm->num_classes, 9) * 4 > INT_MAX) {  // grid.y = tiles of 256; 32-bit offsets into one sample's rows
    set_error("pp_assign_targets*_dev: bad sizes (A=%lld classes=%d)", (long long)A, prm->num_classes);
    return PP_ERR_VALUE;
  }
  if (!cls_targets || !reg_targets ||
      (g_total > 0 && (!g_corners || !g_centers_img || !g_centers || !g_wlh || !g_yaw || !g_class))) {
    set_error("pp_assign_targets*_dev: NULL array");
    return PP_ERR_VALUE;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DeviceGuard2 guard(ctx->device);
  // Which form: anchors on the fly -> the box-centric kernel (k_targets_gt); anchor arrays -> k_targets.
  static const int form_env = [] {  // development knob: PP_TARGETS_FORM=anchors|boxes
    const char *e = getenv("PP_TARGETS_FORM");
    return !e ? 0 : (e[0] == 'a' ? 1 : 2);
  }();
  bool boxes_form = an.grid && form_env != 1 && an.per_cell <= kLdsTypes;  // (the box-centric kernel keeps the type table in LDS)
  // scratch: [0,8192) error flag + every sample's {list counter, ticket, pair counter} | col_max[Gcap] |
  // col_win[Gcap] | first-level tickets | cand[cand_per_gt * Gcap] | pos[pos_per_gt * Gcap] | best, bestj
  // [batch * A]; sample b owns the rows [g_off[b], g_off[b+1]) of each (Gcap = all samples' G)
  const size_t gcap = (size_t)std::max<int64_t>(g_total, 1);
  const size_t nwg = (size_t)((A + kTgtThreads - 1) / kTgtThreads);
  int g_max = 0;
  for (int b = 0; b < batch; ++b) g_max = std::max<int>(g_max, g_counts[b]);
  // box-centric: candidate cells per axis <= floor(20 * fm_scale) + 3 (the +-10 gate, one cell generous either side)
  // ... kCandPerWg of them per PAIR workgroup, fewer when the launch is small: 32 candidates leave ~22 pairs, ONE
  // clip round, where 64 leave ~45 and take two -- worth it as long as the workgroups still fit the chip at once
  // (one sample: 15.1 -> 14.2 us; four samples: 22.6 us with 64 against 25.3 with 32)
  size_t splits = 1, cand_per_wg = kCandPerWg;
  if (boxes_form) {
    const double per_axis = std::min(std::floor(20.0 * an.fm_scale) + 3.0, 32768.0);
    const double cand = per_axis * per_axis * an.per_cell;
    static const int cpw_env = [] {  // development knob: PP_TARGETS_CAND=<candidates per workgroup, 8..64>
      const char *e = getenv("PP_TARGETS_CAND");
      return e ? std::min(std::max(atoi(e), 8), kCandPerWg) : 0;
    }();
    for (size_t c : {(size_t)32, (size_t)40, (size_t)48, (size_t)64}) {
      cand_per_wg = c;
      if ((double)g_total * std::ceil(cand / (double)c) + (double)batch * kZeroWgs <= 900.0) break;
    }
    if (cpw_env) cand_per_wg = (size_t)cpw_env;
    splits = (size_t)std::min(std::max(std::ceil(cand / (double)cand_per_wg), 1.0), 64.0);
    // grid.y holds the zero-fill workgroups and every box's PAIR workgroups: a sample with more boxes than that
    // (eleven thousand at C3's six workgroups per box) goes through the anchor-centric kernel, which walks any
    // number of boxes in chunks
    if ((size_t)kZeroWgs + (size_t)g_max * splits > 65535) {
      boxes_form = false;
      splits = 1;
      cand_per_wg = kCandPerWg;
    }
  }
  // positives per box: at most one entry per candidate.  The PAIR loop strides over ALL of a box's candidates, also
  // when `splits` was clamped to 64 workgroups (fm_scale ~1 with six anchors per cell: 3174 candidates against
  // 64 x 32), so the list is sized from the candidate count itself, not from splits x cand_per_wg
  size_t pos_per_gt = 0;
  if (boxes_form) {
    const double per_axis = std::min(std::floor(20.0 * an.fm_scale) + 3.0, 32768.0);
    const double cand = std::min(per_axis * per_axis * an.per_cell, (double)A);
    pos_per_gt = std::max(splits * cand_per_wg, (size_t)std::ceil(cand));
    // the PAIR role addresses a sample's pair entries through one buffer resource with 32-bit byte offsets
    if ((double)g_max * (double)pos_per_gt * sizeof(PosRow) >= 2147483648.0) {
      boxes_form = false;
      splits = 1;
      cand_per_wg = kCandPerWg;
      pos_per_gt = 0;
    }
  }
  const size_t units = boxes_form ? (size_t)kZeroWgs + (size_t)g_max * splits : nwg;  // workgroups per sample
  if (units > 65535) {
    set_error("pp_assign_targets*_dev: %zu workgroups per sample (limit 65535)", units);
    return PP_ERR_VALUE;
  }
  const size_t cand_per_gt = boxes_form ? splits : nwg;
  const size_t off_cmax = 8192, off_cwin = off_cmax + gcap * 8;
  // first-level tickets: groups of ~sqrt(workgroups) (a power of two), one 64-byte line per group and sample
  int ticket_shift = 0;
  while (((size_t)1 << (2 * ticket_shift)) < units) ++ticket_shift;
  const size_t ngrp = ((units - 1) >> ticket_shift) + 1;
  const size_t n_ticket1 = (size_t)PP_MAX_BATCH * ngrp * kTicketPad;
  const size_t off_tk1 = (off_cwin + gcap * 8 + 255) / 256 * 256;
  const size_t off_cand = (off_tk1 + n_ticket1 * 4 + 255) / 256 * 256;
  const size_t off_pos = (off_cand + cand_per_gt * gcap * sizeof(ColEntry) + 255) / 256 * 256;
  const size_t off_prow = (off_pos + pos_per_gt * gcap * sizeof(PosKey) + 255) / 256 * 256;
  const size_t off_crow = (off_prow + pos_per_gt * gcap * sizeof(PosRow) + 255) / 256 * 256;
  const size_t off_best = (off_crow + (boxes_form ? cand_per_gt * gcap * kRowPitch * 4 : 0) + 255) / 256 * 256;
  const size_t n_best = boxes_form ? (size_t)batch * (size_t)A : 0;
  const size_t off_bestj = off_best + n_best * 8;
  const size_t need = off_bestj + n_best * 4;
  static_assert(256 + PP_MAX_BATCH * kCounterStride * 4 <= 8192, "counter block");
  bool grew = false;
  int rc = ctx->iou_ws.ensure(need, &grew);
  if (rc) return rc;
  char *ws = static_cast<char *>(ctx->iou_ws.ptr);
  TargetArgs t;
  std::memset(&t, 0, sizeof t);
  t.A = A;
  t.G = 0;  // per sample: sample_view
  t.a_corners = an.corners;
  t.a_centers = an.centers;
  t.a_wlh = an.wlh;
  t.a_yaw = an.yaw;
  t.a_center_cols = 3;
  t.g_center_cols = 3;
  t.ious = nullptr;
  t.grid = an.grid;
  t.fm_w = an.fm_w;
  t.fm_h = an.fm_h;
  t.per_cell = an.per_cell;
  t.fm_scale = an.fm_scale;
  {
    int e2 = 0;
    t.inv_scale = (an.grid && std::frexp(an.fm_scale, &e2) == 0.5) ? 1.0 / an.fm_scale : 0.0;
  }
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
  t.cand_count = reinterpret_cast<unsigned *>(ws + 256);
  t.ticket = reinterpret_cast<unsigned *>(ws + 256 + 64);
  t.pos_count = reinterpret_cast<unsigned *>(ws + 256 + 96);
  t.col_max = reinterpret_cast<u64 *>(ws + off_cmax);
  t.col_win = reinterpret_cast<u64 *>(ws + off_cwin);
  t.cand = reinterpret_cast<ColEntry *>(ws + off_cand);
  t.cand_per_gt = (unsigned)cand_per_gt;
  t.ticket1 = reinterpret_cast<unsigned *>(ws + off_tk1);
  t.ticket_shift = ticket_shift;
  t.ticket_groups = (int)ngrp;
  t.gt_splits = (int)splits;
  t.cand_per_wg = (int)cand_per_wg;
  t.pos = reinterpret_cast<PosKey *>(ws + off_pos);
  t.pos_rows = boxes_form ? reinterpret_cast<PosRow *>(ws + off_prow) : nullptr;
  t.pos_per_gt = (unsigned)pos_per_gt;
  t.cand_rows = boxes_form ? reinterpret_cast<float *>(ws + off_crow) : nullptr;
  t.best = boxes_form ? reinterpret_cast<u64 *>(ws + off_best) : nullptr;
  t.bestj = boxes_form ? reinterpret_cast<unsigned *>(ws + off_bestj) : nullptr;
  t.cls_targets = cls_targets;
  t.reg_targets = reg_targets;
  // The scratch words are re-armed by every sample's tail at the end of every call; only a
  // fresh / regrown / re-shaped workspace needs the init.
  // ... identified by every value that moves a scratch word (no packed bit fields: two assigners with different grids
  // on one context must never look alike)
  const unsigned long long key[9] = {(unsigned long long)A, (unsigned long long)gcap, (unsigned long long)batch,
                                     (unsigned long long)units, boxes_form ? 2ull : 1ull, (unsigned long long)splits,
                                     (unsigned long long)cand_per_wg, (unsigned long long)off_best,
                                     (unsigned long long)ngrp};
  if (grew || std::memcmp(ctx->tgt_key, key, sizeof key) != 0) {
    const unsigned gb = (unsigned)((std::max<size_t>(std::max<size_t>(gcap, PP_MAX_BATCH), n_ticket1) + 255) / 256);
    hipLaunchKernelGGL(k_targets_init, dim3(gb), dim3(256), 0, stream, t.col_max, t.col_win,
                       (int)gcap, t.errflag, t.cand_count, t.ticket, t.ticket1, (int)n_ticket1);
    if (boxes_form)
      hipLaunchKernelGGL(k_targets_gt_init, dim3((unsigned)((std::max<size_t>(n_best, PP_MAX_BATCH) + 255) / 256)),
                         dim3(256), 0, stream, t.best, t.bestj, (int64_t)n_best, t.pos_count);
    std::memcpy(ctx->tgt_key, key, sizeof key);
  }
  if (boxes_form)
    hipLaunchKernelGGL(k_targets_gt, dim3((unsigned)batch, (unsigned)units), dim3(kTgtThreads), 0, stream, t, bt);
  else
    hipLaunchKernelGGL(k_targets<false>, dim3((unsigned)batch, (unsigned)nwg), dim3(kTgtThreads), 0, stream, t, bt);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) {
    std::memset(ctx->tgt_key, 0, sizeof ctx->tgt_key);  // counters in an unknown state: re-arm on the next call
    set_error("k_targets launch failed: %s", hipGetErrorString(e));
    return PP_ERR_HIP;
  }
  return PP_OK;
}

static int check_anchor_arrays(const double *a_corners, const double *a_centers, const double *a_wlh,
                               const double *a_yaw, AnchorSource *an) {
  if (!a_corners || !a_centers || !a_wlh || !a_yaw) {
    set_error("pp_assign_targets*_dev: NULL anchor array");
    return PP_ERR_VALUE;
  }
  an->corners = a_corners;
  an->centers = a_centers;
  an->wlh = a_wlh;
  an->yaw = a_yaw;
  return PP_OK;
}

static int check_anchor_grid(int fm_height, int fm_width, double fm_scale, int per_cell,
                             const double *anchor_types_dev, AnchorSource *an) {
  if (fm_height < 1 || fm_width < 1 || per_cell < 1 || per_cell > 1024 || !(fm_scale > 0.0) ||
      !anchor_types_dev || (int64_t)fm_height * fm_width * per_cell > 65535ll * kTgtThreads) {
    set_error("pp_assign_targets_grid*_dev: bad anchor grid (%dx%d, %d per cell, scale %g)",
              fm_height, fm_width, per_cell, fm_scale);
    return PP_ERR_VALUE;
  }
  an->grid = 1;
  an->fm_w = fm_width;
  an->fm_h = fm_height;
  an->per_cell = per_cell;
  an->fm_scale = fm_scale;
  an->types = anchor_types_dev;
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
  AnchorSource an;
  if (int rc = check_anchor_arrays(a_corners, a_centers, a_wlh, a_yaw, &an)) return rc;
  if (G < 0 || G > 65535) {
    set_error("pp_assign_targets_dev: bad sizes (G=%lld)", (long long)G);
    return PP_ERR_VALUE;
  }
  const int32_t g1 = (int32_t)G;
  return assign_targets_impl(ctx, stream_, 1, &g1, A, an, g_corners, g_centers_img, g_centers, g_wlh,
                             g_yaw, g_class, prm, cls_targets, reg_targets);
}

extern "C" int pp_assign_targets_batch_dev(pp_ctx_t *ctx, void *stream_, int32_t batch,
                                           const int32_t *g_counts, int64_t A, const double *a_corners,
                                           const double *a_centers, const double *a_wlh,
                                           const double *a_yaw, const double *g_corners,
                                           const double *g_centers_img, const double *g_centers,
                                           const double *g_wlh, const double *g_yaw,
                                           const int32_t *g_class, const pp_target_params_t *prm,
                                           float *cls_targets, float *reg_targets) {
  AnchorSource an;
  if (int rc = check_anchor_arrays(a_corners, a_centers, a_wlh, a_yaw, &an)) return rc;
  return assign_targets_impl(ctx, stream_, batch, g_counts, A, an, g_corners, g_centers_img, g_centers,
                             g_wlh, g_yaw, g_class, prm, cls_targets, reg_targets);
}

extern "C" int pp_assign_targets_grid_dev(pp_ctx_t *ctx, void *stream_, int fm_height, int fm_width,
                                          double fm_scale, int per_cell,
                                          const double *anchor_types_dev, int64_t G,
                                          const double *g_corners, const double *g_centers_img,
                                          const double *g_centers, const double *g_wlh,
                                          const double *g_yaw, const int32_t *g_class,
                                          const pp_target_params_t *prm, float *cls_targets,
                                          float *reg_targets) {
  AnchorSource an;
  if (int rc = check_anchor_grid(fm_height, fm_width, fm_scale, per_cell, anchor_types_dev, &an)) return rc;
  if (G < 0 || G > 65535) {
    set_error("pp_assign_targets_grid_dev: bad sizes (G=%lld)", (long long)G);
    return PP_ERR_VALUE;
  }
  const int32_t g1 = (int32_t)G;
  return assign_targets_impl(ctx, stream_, 1, &g1, (int64_t)fm_height * fm_width * per_cell, an, g_corners,
                             g_centers_img, g_centers, g_wlh, g_yaw, g_class, prm, cls_targets,
                             reg_targets);
}

extern "C" int pp_assign_targets_grid_batch_dev(pp_ctx_t *ctx, void *stream_, int32_t batch,
                                                const int32_t *g_counts, int fm_height, int fm_width,
                                                double fm_scale, int per_cell,
                                                const double *anchor_types_dev, const double *g_corners,
                                                const double *g_centers_img, const double *g_centers,
                                                const double *g_wlh, const double *g_yaw,
                                                const int32_t *g_class, const pp_target_params_t *prm,
                                                float *cls_targets, float *reg_targets) {
  AnchorSource an;
  if (int rc = check_anchor_grid(fm_height, fm_width, fm_scale, per_cell, anchor_types_dev, &an)) return rc;
  return assign_targets_impl(ctx, stream_, batch, g_counts, (int64_t)fm_height * fm_width * per_cell, an,
                             g_corners, g_centers_img, g_centers, g_wlh, g_yaw, g_class, prm,
                             cls_targets, reg_targets);
}

#ifdef PP_IOU_STAMPS
// development builds only (tools/lab): the phase stamps of the last k_targets launch, 16 per workgroup
extern "C" int pp_debug_iou_stamps(unsigned long long *out, int n_words) {
  if (hipDeviceSynchronize() != hipSuccess) return PP_ERR_HIP;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pp::g_iou_stamps), (size_t)n_words * 8) == hipSuccess ? PP_OK : PP_ERR_HIP;
}
#endif
