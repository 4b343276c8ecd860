// pp_voxelize.hip -- the pillar voxelizer as hand-written HIP for gfx950 (CDNA4).
//
// Replaces create_pillars (/root/reference data/pillars.cpp:236-398) and, on the
// device-resident path, the caller glue around it (data/dataset.py:88-106):
// np.zeros + create_pillars + transpose to [9,P,N] + f64->f32 + indices->int64.
//
// Pipeline (one launch each, grid.y = sweep of the batch):
//   k_bin_count  point -> cell slot (f64 true division + floor, half-open range
//                test: pillars.cpp:271-280); ONE returning atomic per point gives
//                the per-cell population and the point's arrival rank
//   k_scan       single-pass decoupled-look-back scan over the cell grid:
//                pillar index of every non-empty cell (P-index compaction) and
//                CSR offset of its bucket, written in place of the counts
//   k_fill       atomic-free CSR fill: point record + point index go to
//                bucket start + arrival rank
//   k_emit       one wave per 4 consecutive pillars: coalesced bucket read,
//                LDS-staged, input order restored, sequential running mean
//                (pillars.cpp:311-328), N-cap, 9 features (pillars.cpp:30-31,
//                48-56,381-383), dense [9,P,N] f32 store incl. the zero padding,
//                [P,3] int64 indices
//
// The path is HBM-bound: 97% of the bytes are the dense store of
// k_emit.  Every 128-byte line of the output is written once, whole: lines
// without live points are zero-filled before the bucket data arrives, lines
// with live points are written (data + zero tail) after it.
// All arithmetic that decides a value is f64 with contraction disabled
// (-ffp-contract=off), matching the reference's x86-64 build.

#include "pp_common.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>

namespace pp {

// ------------------------------------------------------------------------- //
// constants                                                                  //
// ------------------------------------------------------------------------- //
constexpr int kWave = 64;
constexpr int kBinThreads = 256;
constexpr int kScanThreads = 1024;
constexpr int kScanTile = kScanThreads * 4;  // cells per scan workgroup
constexpr int kEmitWaves = 4;                // waves per emit workgroup
constexpr int kEmitThreads = kEmitWaves * kWave;
constexpr int KW = 4;      // pillars per emit wave
#ifndef PP_CAPW
#define PP_CAPW 128
#endif
constexpr int CAPW = PP_CAPW;  // pooled bucket capacity (points, 4-padded per pillar) per emit wave
constexpr int kPre = CAPW / 64;  // bucket entries prefetched per lane
constexpr unsigned kSpinLimit = 1u << 26;

using u64 = unsigned long long;

struct NPoints {
  int n[PP_MAX_BATCH];
};

// status word of the look-back scan: [63:62] flag, [61:31] points, [30:0] pillars
constexpr u64 kFlagAgg = 1ull << 62;
constexpr u64 kFlagPre = 2ull << 62;
__device__ __forceinline__ u64 st_pack(u64 flag, u64 v) {
  // v = points << 32 | pillars
  return flag | ((v >> 32) << 31) | (v & 0x7FFFFFFFull);
}
__device__ __forceinline__ u64 st_payload(u64 s) {
  return (((s >> 31) & 0x7FFFFFFFull) << 32) | (s & 0x7FFFFFFFull);
}

__device__ __forceinline__ void wave_sync() {
  // LDS operations of one wave execute in program order; this only stops the
  // compiler from moving LDS accesses across a cross-lane hand-off.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int cell_to_slot(int cell, const GridGeom &g) {
  if (g.order == PP_ORDER_ROW_MAJOR) return cell;
  return (int)(((u64)cell * g.mult) % (u64)g.ncells);
}
__device__ __forceinline__ int slot_to_cell(int slot, const GridGeom &g) {
  if (g.order == PP_ORDER_ROW_MAJOR) return slot;
  return (int)(((u64)slot * g.mult_inv) % (u64)g.ncells);
}

// ------------------------------------------------------------------------- //
// k_bin_count                                                                 //
// ------------------------------------------------------------------------- //
template <typename T>
__device__ __forceinline__ void load_point(const T *pts, int64_t row, int64_t s0,
                                           int64_t s1, bool contig, T &x, T &y,
                                           T &z, T &r);

template <>
__device__ __forceinline__ void load_point<float>(const float *pts, int64_t row,
                                                  int64_t s0, int64_t s1,
                                                  bool contig, float &x, float &y,
                                                  float &z, float &r) {
  if (contig) {
    float4 v = reinterpret_cast<const float4 *>(pts)[row];  // 16 B/lane, coalesced
    x = v.x; y = v.y; z = v.z; r = v.w;
  } else {
    const float *p = pts + row * s0;
    x = p[0]; y = p[s1]; z = p[2 * s1]; r = p[3 * s1];
  }
}
template <>
__device__ __forceinline__ void load_point<double>(const double *pts, int64_t row,
                                                   int64_t s0, int64_t s1,
                                                   bool contig, double &x, double &y,
                                                   double &z, double &r) {
  if (contig) {
    const double4 v = reinterpret_cast<const double4 *>(pts)[row];
    x = v.x; y = v.y; z = v.z; r = v.w;
  } else {
    const double *p = pts + row * s0;
    x = p[0]; y = p[s1]; z = p[2 * s1]; r = p[3 * s1];
  }
}

// pillars.cpp:271-280.  Returns the cell id (row-major over ascending
// (canvas_y, canvas_x)) or -1 when the point is outside the half-open box.
// NaN coordinates fail the positive comparisons and are dropped.
__device__ __forceinline__ int point_cell(double x, double y, double z,
                                          const GridGeom &g) {
  if (!(x >= g.x_min && x < g.x_max && y >= g.y_min && y < g.y_max &&
        z >= g.z_min && z < g.z_max))
    return -1;
  const double fx = floor((x - g.x_min) / g.x_step);
  const double fy = floor((y - g.y_min) / g.y_step);
  const int ix = (int)fx, iy = (int)fy;
  return ((g.ny - 1) - iy) * g.nx + ix;
}

template <typename T>
__global__ __launch_bounds__(kBinThreads) void k_bin_count(
    const T *__restrict__ pts, int64_t sweep_stride, int64_t s0, int64_t s1,
    int contig, NPoints np, GridGeom g, int2 *__restrict__ cell_rank, int ncap,
    int *__restrict__ cursor) {
  const int b = blockIdx.y;
  const int n = np.n[b];
  const int i = blockIdx.x * kBinThreads + threadIdx.x;
  if (i >= n) return;
  T x, y, z, r;
  load_point<T>(pts + (int64_t)b * sweep_stride * 4, i, s0, s1, contig != 0, x, y, z, r);
  const int cell = point_cell((double)x, (double)y, (double)z, g);
  int slot = -1, rank = 0;
  if (cell >= 0) {
    slot = cell_to_slot(cell, g);
    // the only atomic of the pipeline: population count + arrival rank in one
    rank = atomicAdd(&cursor[(int64_t)b * g.ncells_pad + slot], 1);
  }
  cell_rank[(int64_t)b * ncap + i] = make_int2(slot, rank);
}

// ------------------------------------------------------------------------- //
// k_scan: decoupled look-back over the cell grid                             //
// ------------------------------------------------------------------------- //
__device__ __forceinline__ u64 shfl_up64(u64 v, int d) {
  int lo = __shfl_up((int)(v & 0xFFFFFFFFull), d, kWave);
  int hi = __shfl_up((int)(v >> 32), d, kWave);
  return ((u64)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ u64 wave_sum64(u64 v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    int lo = __shfl_xor((int)(v & 0xFFFFFFFFull), d, kWave);
    int hi = __shfl_xor((int)(v >> 32), d, kWave);
    v += ((u64)(unsigned)hi << 32) | (unsigned)lo;
  }
  return v;
}

__global__ __launch_bounds__(kScanThreads) void k_scan(
    int *__restrict__ cursor, int ncells_pad, int P, int4 *__restrict__ pillar_meta,
    u64 *status, unsigned *ticket, int2 *__restrict__ totals, int *errflag) {
  __shared__ unsigned s_ticket;
  __shared__ u64 s_wave[kScanThreads / kWave];
  __shared__ u64 s_excl;
  const int b = blockIdx.y;
  const int nwg = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // Dynamic tile id: a workgroup that holds ticket t knows tickets < t have
  // started, so the look-back below never waits on an unscheduled workgroup.
  if (tid == 0) s_ticket = atomicAdd(&ticket[b], 1u);
  __syncthreads();
  const int t = (int)s_ticket;
  u64 *st = status + (int64_t)b * nwg;

  int4 *cur = reinterpret_cast<int4 *>(cursor + (int64_t)b * ncells_pad) +
              (int64_t)t * kScanThreads + tid;
  const int4 c = *cur;
  const u64 mine = ((u64)((unsigned)c.x + (unsigned)c.y + (unsigned)c.z + (unsigned)c.w) << 32) |
                   (u64)((c.x > 0) + (c.y > 0) + (c.z > 0) + (c.w > 0));
  // inclusive scan inside the wave
  u64 inc = mine;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    u64 o = shfl_up64(inc, d);
    if (lane >= d) inc += o;
  }
  if (lane == kWave - 1) s_wave[wv] = inc;
  __syncthreads();
  u64 wave_off = 0, agg = 0;
#pragma unroll
  for (int k = 0; k < kScanThreads / kWave; ++k) {
    if (k < wv) wave_off += s_wave[k];
    agg += s_wave[k];
  }
  // publish + look back (wave 0)
  if (wv == 0) {
    u64 excl = 0;
    if (t == 0) {
      if (lane == 0)
        __hip_atomic_store(&st[0], st_pack(kFlagPre, agg), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0)
        __hip_atomic_store(&st[t], st_pack(kFlagAgg, agg), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      int base = t - 1;
      bool failed = false;
      while (true) {
        const int j = base - lane;
        u64 s = kFlagPre;  // virtual predecessor of tile 0: prefix 0
        if (j >= 0) {
          unsigned spins = 0;
          do {
            s = __hip_atomic_load(&st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } while ((s >> 62) == 0 && ++spins < kSpinLimit);
          if ((s >> 62) == 0) failed = true;
        }
        const u64 maskP = __ballot((s >> 62) == 2);
        const int firstP = maskP ? (__ffsll((long long)maskP) - 1) : kWave;
        const u64 val = (lane <= firstP) ? st_payload(s) : 0ull;
        excl += wave_sum64(val);
        if (maskP || __any(failed)) break;
        base -= kWave;
      }
      if (__any(failed) && lane == 0) atomicExch(errflag, 1);
      if (lane == 0)
        __hip_atomic_store(&st[t], st_pack(kFlagPre, excl + agg), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_excl = excl;
  }
  __syncthreads();
  const u64 base = s_excl + wave_off + (inc - mine);
  int p = (int)(base & 0xFFFFFFFFull);
  int s = (int)(base >> 32);
  const int slot0 = (t * kScanThreads + tid) * 4;
  const int vals[4] = {c.x, c.y, c.z, c.w};
  int outv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    outv[e] = s;
    if (vals[e] > 0) {
      if (p < P) pillar_meta[(int64_t)b * P + p] = make_int4(slot0 + e, s, vals[e], 0);
      ++p;
      s += vals[e];
    }
  }
  *cur = make_int4(outv[0], outv[1], outv[2], outv[3]);  // cursor = bucket start
  if (t == nwg - 1 && tid == 0) {
    const u64 tot = s_excl + agg;
    totals[b] = make_int2((int)(tot & 0xFFFFFFFFull), (int)(tot >> 32));
  }
}

// ------------------------------------------------------------------------- //
// k_fill: atomic-free CSR fill; also re-arms the scan's ticket/status words   //
// ------------------------------------------------------------------------- //
template <typename T> struct Rec4;
template <> struct Rec4<float> { using type = float4; };
template <> struct Rec4<double> { using type = double4; };

template <typename T>
__global__ __launch_bounds__(kBinThreads) void k_fill(
    const T *__restrict__ pts, int64_t sweep_stride, const int2 *__restrict__ cell_rank,
    int ncap, NPoints np, const int *__restrict__ cursor, int ncells_pad,
    typename Rec4<T>::type *__restrict__ sorted_pts, int *__restrict__ sorted_idx,
    u64 *status, int nwg_scan, unsigned *ticket) {
  const int b = blockIdx.y;
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < nwg_scan; i += kBinThreads)
      status[(int64_t)b * nwg_scan + i] = 0ull;
    if (threadIdx.x == 0) ticket[b] = 0u;
  }
  const int i = blockIdx.x * kBinThreads + threadIdx.x;
  if (i >= np.n[b]) return;
  const int2 cr = cell_rank[(int64_t)b * ncap + i];
  if (cr.x < 0) return;
  const typename Rec4<T>::type rec =
      reinterpret_cast<const typename Rec4<T>::type *>(pts + (int64_t)b * sweep_stride * 4)[i];
  const int pos = cursor[(int64_t)b * ncells_pad + cr.x] + cr.y;  // bucket start + arrival rank
  sorted_pts[(int64_t)b * ncap + pos] = rec;
  sorted_idx[(int64_t)b * ncap + pos] = i;
}

// ------------------------------------------------------------------------- //
// k_emit                                                                      //
// ------------------------------------------------------------------------- //
template <typename TIn>
struct alignas(32) WaveLds {
  int idx[CAPW];  // point indices as stored by k_fill (arrival order), 4-aligned buckets, INT_MAX pads
  TIn px[CAPW], py[CAPW], pz[CAPW], pr[CAPW];  // points, input order per pillar
  union {
    double4 cq[CAPW];  // chain operands of one point: {n/(n+1), x/(n+1), y/(n+1), z/(n+1)}
    float feat[PP_NUM_FEATURES][CAPW];  // f32 features (dense / fused-net modes), aliases cq
  } u;
  double mean[KW][3];
  double cx[KW], cy[KW];  // canvas_x / canvas_y (pillars.cpp:278-280), once per pillar
  int cnt[KW], live[KW], slot[KW], start[KW];
};

struct EmitArgs {
  GridGeom g;
  NPoints np;
  int P, N, ncap;
  const int4 *pillar_meta;  // [B][P] {slot, start, count, -}
  const int2 *totals;       // [B]    {cells, points}
  const int *sorted_idx;    // [B][ncap] point index, CSR (bucket) order
  const void *sorted_pts;   // [B][ncap] point record (x,y,z,r), CSR order
  const int2 *cell_rank;    // [B][ncap] {slot, arrival rank} per input point
  int *cursor;              // [B][ncells_pad], zeroed here for the next call
  const void *pts;          // [B][sweep_stride][4] (contiguous rows)
  int64_t sweep_stride;
  // dense mode
  float *out;          // [B][9][P][N]
  long long *idx_out;  // [B][P][3]
  // compact mode
  double *feat_out;  // [B][ncap][9]
  // fused feature-net mode
  const float *pfn_w;  // [64][12]: w[0..8], bias, bn scale, bn shift per output channel
  float *pfn_out;      // [B][64][P], or NULL when only the canvas is wanted
  // ... scattered straight into the BEV canvas (PPScatter, model/model.py:53-62)
  float *canvas;       // NULL, [B][H][W][64] (channels last) or [B][64][H][W]
  int canvas_h, canvas_w, canvas_nhwc;
};

enum { kModeDenseVec4 = 0, kModeDenseScalar = 1, kModeCompact = 2, kModePfn = 3 };

constexpr int kPfnChannels = 64;  // one lane per output channel (model/model.py:28: 9 -> 64)

// Fused PPFeatureNet (inference): y[c,p] = max_n BN_c(ReLU(b_c + sum_d W[c,d] x[d,p,n]))
// over ALL N slots of the pillar, zero-padded ones included (model/model.py:31-40:
// conv1x1, ReLU, THEN BatchNorm, max over N).  BN in eval mode is the affine map
// s*r + t, so the max over n moves inside: s >= 0 ? s*max(r) + t : s*min(r) + t.
struct PfnAcc {
  float w[9], bias, scale, shift;
  float rmax[KW], rmin[KW];
};

__device__ __forceinline__ float pfn_relu_conv(const PfnAcc &A, const float x[9]) {
  float r = A.bias;
#pragma unroll
  for (int d = 0; d < 9; ++d) r = fmaf(A.w[d], x[d], r);
  return fmaxf(r, 0.0f);
}

__device__ __forceinline__ void pillar_canvas(int slot, const GridGeom &g,
                                              double &canvas_x, double &canvas_y) {
  const int cell = slot_to_cell(slot, g);
  const int ix = cell % g.nx;
  const int ry = cell / g.nx;
  canvas_x = (double)ix;
  const double fy = (double)((g.ny - 1) - ry);
  canvas_y = (g.canvas_height - 1) - fy;  // pillars.cpp:280
}

// nine features of one point, pillars.cpp:48-56 (order), :30-31 (xp,yp), :381-383
__device__ __forceinline__ void point_features(double x, double y, double z, double r,
                                               double canvas_x, double canvas_y,
                                               const double *mean, double f[9]) {
  f[0] = x;
  f[1] = y;
  f[2] = z;
  f[3] = r;
  f[4] = canvas_x - x;
  f[5] = canvas_y - y;
  f[6] = mean[0] - x;
  f[7] = mean[1] - y;
  f[8] = mean[2] - z;
}

// A pillar whose bucket does not fit the wave's LDS pool: re-scan the sweep's
// cell ids in input order and compact the matches with ballot + popcount, so
// the order is restored without sorting.  O(n_points/64) per such pillar.
template <typename TIn, int MODE>
__device__ void emit_big_pillar(WaveLds<TIn> &L, const EmitArgs &a, int b, int k, int p,
                                int lane, PfnAcc *acc = nullptr, float *rmax = nullptr,
                                float *rmin = nullptr) {
  const int slot = L.slot[k];
  const int nb = a.np.n[b];
  const TIn *pts = reinterpret_cast<const TIn *>(a.pts) + (int64_t)b * a.sweep_stride * 4;
  const int2 *cell_rank = a.cell_rank + (int64_t)b * a.ncap;
  const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  double m0 = 0, m1 = 0, m2 = 0;
  int seen = 0;
  for (int base = 0; base < nb; base += kWave) {
    const int i = base + lane;
    const bool match = (i < nb) && (cell_rank[i].x == slot);
    const u64 mask = __ballot(match);
    if (!mask) continue;
    const int rank = __popcll(mask & lt_mask);
    if (match) {
      const TIn x = pts[(int64_t)i * 4 + 0], y = pts[(int64_t)i * 4 + 1],
                z = pts[(int64_t)i * 4 + 2];
      const double n = (double)(seen + rank), den = n + 1;
      L.px[rank] = x;
      L.py[rank] = y;
      L.pz[rank] = z;
      L.u.cq[rank] = make_double4(n / den, (double)x / den, (double)y / den, (double)z / den);
    }
    wave_sync();
    const int c = __popcll(mask);
    for (int t = 0; t < c; ++t) {  // uniform: every lane carries the chain
      if (seen + t == 0) {
        m0 = (double)L.px[0];
        m1 = (double)L.py[0];
        m2 = (double)L.pz[0];
      } else {
        const double4 o = L.u.cq[t];
        m0 = m0 * o.x + o.y;
        m1 = m1 * o.x + o.z;
        m2 = m2 * o.x + o.w;
      }
    }
    seen += c;
    wave_sync();
  }
  const double mean[3] = {m0, m1, m2};
  const double cx = L.cx[k], cy = L.cy[k];
  const int N = a.N;
  const int live = L.live[k];
  seen = 0;
  for (int base = 0; base < nb && seen < N; base += kWave) {
    const int i = base + lane;
    const bool match = (i < nb) && (cell_rank[i].x == slot);
    const u64 mask = __ballot(match);
    if (!mask) continue;
    const int n = seen + __popcll(mask & lt_mask);
    if (match && n < N) {
      const TIn x = pts[(int64_t)i * 4 + 0], y = pts[(int64_t)i * 4 + 1],
                z = pts[(int64_t)i * 4 + 2], r = pts[(int64_t)i * 4 + 3];
      double f[9];
      point_features((double)x, (double)y, (double)z, (double)r, cx, cy, mean, f);
      if (MODE == kModeCompact) {
        double *o = a.feat_out + ((int64_t)b * a.ncap + L.start[k] + n) * 9;
#pragma unroll
        for (int d = 0; d < 9; ++d) o[d] = f[d];
      } else if (MODE == kModePfn) {
        const int t = __popcll(mask & lt_mask);  // position inside this chunk
#pragma unroll
        for (int d = 0; d < 9; ++d) L.u.feat[d][t] = (float)f[d];
      } else {
        float *o = a.out + (int64_t)b * 9 * a.P * N;
#pragma unroll
        for (int d = 0; d < 9; ++d) o[((int64_t)d * a.P + p) * N + n] = (float)f[d];
      }
    }
    if (MODE == kModePfn) {
      // every lane is a channel: fold this chunk's live points into its max/min
      wave_sync();
      const int c = min(__popcll(mask), N - seen);
      for (int t = 0; t < c; ++t) {
        float xv[9];
#pragma unroll
        for (int d = 0; d < 9; ++d) xv[d] = L.u.feat[d][t];
        const float rr = pfn_relu_conv(*acc, xv);
        *rmax = fmaxf(*rmax, rr);
        *rmin = fminf(*rmin, rr);
      }
      wave_sync();
    }
    seen += __popcll(mask);
  }
  if (MODE == kModeDenseVec4) {
    // zero the tail of the 16-byte group that straddles `live`
    const int up = (live + 3) & ~3;
    const int e = live + (lane / 9), d = lane % 9;
    if (lane < 27 && e < up) {
      float *o = a.out + (int64_t)b * 9 * a.P * N;
      o[((int64_t)d * a.P + p) * N + e] = 0.0f;
    }
  }
}

// Pooled pillars [kbeg,kend) of this wave: entries j (pooled bucket position)
// were prefetched by the caller into registers (idx_r[it] / rec_r[it] for
// j = lane + 64*it) straight from the CSR arrays.  Leaves the f32 features of
// the live points in L.u.feat (dense vec4 mode) or stores them (other modes).
template <typename TIn, int MODE>
__device__ __forceinline__ void emit_group(WaveLds<TIn> &L, const EmitArgs &a, int b, int p0,
                                           int kbeg, int kend, int lane, const int idx_r[kPre],
                                           const typename Rec4<TIn>::type rec_r[kPre],
                                           int segbeg[KW],
                                           int segpad[KW], int cntk[KW], int T) {
  const int N = a.N;
  // bucket entries go to 4-aligned bucket starts; the up-to-3 pad entries compare as
  // "not smaller" in the rank search below
  if (lane < KW) {
    int sp = 0, sc = 0;
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      sp = (lane == kk) ? segpad[kk] : sp;
      sc = (lane == kk) ? cntk[kk] : sc;
    }
    for (int e = sc; e < ((sc + 3) & ~3); ++e) L.idx[sp + e] = INT_MAX;
  }
#pragma unroll
  for (int it = 0; it < kPre; ++it) {
    const int j = lane + it * kWave;
    if (j < T) {
      int k = 0;
#pragma unroll
      for (int kk = 1; kk < KW; ++kk) k = (j >= segbeg[kk] && cntk[kk] > 0) ? kk : k;
      int sb = 0, sp = 0;
#pragma unroll
      for (int kk = 0; kk < KW; ++kk) {
        sb = (k == kk) ? segbeg[kk] : sb;
        sp = (k == kk) ? segpad[kk] : sp;
      }
      L.idx[sp + (j - sb)] = idx_r[it];
    }
  }
  wave_sync();
  // restore input order: rank of every entry inside its bucket (counts are small)
#pragma unroll
  for (int it = 0; it < kPre; ++it) {
    const int j = lane + it * kWave;
    if (j < T) {
      int k = 0;
#pragma unroll
      for (int kk = 1; kk < KW; ++kk) k = (j >= segbeg[kk] && cntk[kk] > 0) ? kk : k;
      int sb = 0, sc = 0, sp = 0;
#pragma unroll
      for (int kk = 0; kk < KW; ++kk) {
        sb = (k == kk) ? segbeg[kk] : sb;
        sc = (k == kk) ? cntk[kk] : sc;
        sp = (k == kk) ? segpad[kk] : sp;
      }
      const int my = idx_r[it];
      const typename Rec4<TIn>::type rec = rec_r[it];
      int r = 0;
      for (int g4 = 0; g4 < ((sc + 3) >> 2); ++g4) {  // 16-byte LDS reads, four compares each
        const int4 v = *reinterpret_cast<const int4 *>(&L.idx[sp + 4 * g4]);
        r += (v.x < my) + (v.y < my) + (v.z < my) + (v.w < my);
      }
      const int pos = sp + r;  // sorted arrays use 4-aligned bucket starts too
      L.px[pos] = rec.x;
      L.py[pos] = rec.y;
      L.pz[pos] = rec.z;
      L.pr[pos] = rec.w;
      const double n = (double)r, den = n + 1;
      // pillars.cpp:322-326: n/(n+1) and v/(n+1), true f64 divisions, lane-parallel
      L.u.cq[pos] = make_double4(n / den, (double)rec.x / den, (double)rec.y / den,
                                 (double)rec.z / den);
    }
  }
  wave_sync();
  // sequential running mean per pillar, pillars.cpp:311-328 (one lane each)
  if (lane < KW) {
    int sb = 0, sc = 0;
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      sb = (lane == kk) ? segpad[kk] : sb;
      sc = (lane == kk) ? cntk[kk] : sc;
    }
    if (sc > 0) {
      double m0 = (double)L.px[sb], m1 = (double)L.py[sb], m2 = (double)L.pz[sb];
      // the chain is serial in m, but its operands are not: fetch four points' operands
      // ahead of the four dependent mul+add steps that consume them
      int n = 1;
      for (; n + 3 < sc; n += 4) {
        const double4 o0 = L.u.cq[sb + n], o1 = L.u.cq[sb + n + 1], o2 = L.u.cq[sb + n + 2],
                      o3 = L.u.cq[sb + n + 3];
        m0 = m0 * o0.x + o0.y;
        m1 = m1 * o0.x + o0.z;
        m2 = m2 * o0.x + o0.w;
        m0 = m0 * o1.x + o1.y;
        m1 = m1 * o1.x + o1.z;
        m2 = m2 * o1.x + o1.w;
        m0 = m0 * o2.x + o2.y;
        m1 = m1 * o2.x + o2.z;
        m2 = m2 * o2.x + o2.w;
        m0 = m0 * o3.x + o3.y;
        m1 = m1 * o3.x + o3.z;
        m2 = m2 * o3.x + o3.w;
      }
      for (; n < sc; ++n) {
        const double4 o = L.u.cq[sb + n];
        m0 = m0 * o.x + o.y;
        m1 = m1 * o.x + o.z;
        m2 = m2 * o.x + o.w;
      }
      L.mean[lane][0] = m0;
      L.mean[lane][1] = m1;
      L.mean[lane][2] = m2;
    }
  }
  wave_sync();
  // features of the first min(count, N) points of every pillar.  The f32
  // staging array aliases the (now dead) chain operands.
  float *outb = a.out + (int64_t)b * 9 * a.P * N;
  const int start0 = L.start[kbeg];
#pragma unroll
  for (int it = 0; it < kPre; ++it) {
    const int j = lane + it * kWave;
    if (j >= T) continue;
    int k = 0;
#pragma unroll
    for (int kk = 1; kk < KW; ++kk) k = (j >= segbeg[kk] && cntk[kk] > 0) ? kk : k;
    int sb = 0, sp = 0;
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      sb = (k == kk) ? segbeg[kk] : sb;
      sp = (k == kk) ? segpad[kk] : sp;
    }
    const int n = j - sb;
    const int q = sp + n;
    if (n < N) {
      double f[9];
      point_features((double)L.px[q], (double)L.py[q], (double)L.pz[q], (double)L.pr[q],
                     L.cx[k], L.cy[k], L.mean[k], f);
      if (MODE == kModeCompact) {
        double *o = a.feat_out + ((int64_t)b * a.ncap + start0 + j) * 9;
#pragma unroll
        for (int d = 0; d < 9; ++d) o[d] = f[d];
      } else if (MODE == kModeDenseScalar) {
#pragma unroll
        for (int d = 0; d < 9; ++d)
          outb[((int64_t)d * a.P + (p0 + k)) * N + n] = (float)f[d];
      } else {
#pragma unroll
        for (int d = 0; d < 9; ++d) L.u.feat[d][q] = (float)f[d];
      }
    }
  }
  wave_sync();
}

// Dense vec4 mode.  The wave owns, per feature d, one contiguous slab of
// kw_eff*N floats (rowf4 16-byte groups); group `rem` of the slab of feature d
// is group g4 = (d*P + p0)*N4 + rem of the sweep's output, line = g4 >> 3 (128 B).
// A line is "late" when it holds a live point of a POOLED pillar: late lines are
// written whole -- data + zeros -- once the features are in LDS; every other
// line is zero-filled early, before the bucket data has arrived.  The decision
// depends on `rem` only (not on d) when P*N4 is a multiple of 8 groups, which
// lets one decision drive nine back-to-back stores; otherwise the unit of
// deferral falls back from the line to the 16-byte group.  For a non-pooled
// (big) pillar only its head groups are skipped; emit_big_pillar writes those.
struct SlabGeom {
  unsigned rowf4, N4, magic_n4;
  unsigned base0;      // first 16-byte group of the wave's slab in feature plane 0
  unsigned pn4_bytes;  // byte distance between two feature planes
  int nh[KW];          // head groups (ceil(live/4)) per pillar
  unsigned pooled;     // bit k: pillar k's features are (will be) in L.u.feat
  bool aligned;        // one line mask serves all nine planes
  u64 late_lines;      // bit l: line (base0>>3)+l holds a head group of a pooled pillar
};

__device__ __forceinline__ unsigned fastdiv(unsigned n, unsigned d, unsigned magic) {
  return d == 1u ? n : __umulhi(n, magic);  // magic = floor((2^32-1)/d)+1, exact for n*d < 2^32
}

__device__ __forceinline__ u64 slab_late_lines(const SlabGeom &sg, unsigned pooled) {
  u64 m = 0;
  const unsigned line0 = sg.base0 >> 3;
#pragma unroll
  for (int kk = 0; kk < KW; ++kk) {
    if (((pooled >> kk) & 1u) && sg.nh[kk] > 0) {
      const unsigned lo = ((sg.base0 + (unsigned)kk * sg.N4) >> 3) - line0;
      const unsigned hi = ((sg.base0 + (unsigned)kk * sg.N4 + (unsigned)sg.nh[kk] - 1u) >> 3) - line0;
      const unsigned w = hi - lo + 1u;
      m |= (w >= 64u ? ~0ull : ((1ull << w) - 1ull)) << lo;
    }
  }
  return m;
}

typedef int v4i_t __attribute__((ext_vector_type(4)));

// Stores go through a buffer descriptor (wave-uniform base of the sweep's output,
// per-lane 32-bit byte offset, the feature plane as an SGPR offset): nine
// back-to-back buffer_store_dwordx4 per decision with no 64-bit address math.
enum { kPassEarly = 0, kPassLate = 1, kPassAll = 2 };

template <int PASS, typename TIn>
__device__ __forceinline__ void store_slab(const WaveLds<TIn> &L, const SlabGeom &sg,
                                           __amdgpu_buffer_rsrc_t rs, int lane,
                                           const int segbeg[KW]) {
  const v4i_t z4 = {0, 0, 0, 0};
  const unsigned line0 = sg.base0 >> 3;
  for (unsigned rem = lane; rem < sg.rowf4; rem += kWave) {
    const unsigned k = fastdiv(rem, sg.N4, sg.magic_n4);
    const unsigned n4 = rem - k * sg.N4;
    int nhk = 0, sb = 0;
    bool pooled_k = false;
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      nhk = (k == (unsigned)kk) ? sg.nh[kk] : nhk;
      sb = (k == (unsigned)kk) ? segbeg[kk] : sb;
      pooled_k = (k == (unsigned)kk) ? (((sg.pooled >> kk) & 1u) != 0) : pooled_k;
    }
    const bool head = (int)n4 < nhk;
    const bool late = sg.aligned ? (((sg.late_lines >> (((sg.base0 + rem) >> 3) - line0)) & 1ull) != 0)
                                 : (head && pooled_k);
    const unsigned voff = (sg.base0 + rem) * 16u;
    bool write_zero, write_head;
    if (PASS == kPassEarly) {
      write_zero = !late && !head;
      write_head = false;
    } else if (PASS == kPassLate) {
      write_zero = late && !head;
      write_head = late && head && pooled_k;
    } else {  // single pass: the features of every pooled pillar are already in LDS
      write_zero = !head;
      write_head = head && pooled_k;
    }
    if (write_head || write_zero) {
      v4i_t v[PP_NUM_FEATURES];
#pragma unroll
      for (int d = 0; d < PP_NUM_FEATURES; ++d) v[d] = z4;
      if (write_head) {
        const int lv = L.live[k] - 4 * (int)n4;  // live entries in this 16-byte group (>= 1)
        const int j0 = sb + 4 * (int)n4;         // 4-aligned: buckets start on 16-byte LDS boundaries
#pragma unroll
        for (int d = 0; d < PP_NUM_FEATURES; ++d) {
          const v4i_t t = *reinterpret_cast<const v4i_t *>(&L.u.feat[d][j0]);
          v[d].x = t.x;
          v[d].y = lv > 1 ? t.y : 0;
          v[d].z = lv > 2 ? t.z : 0;
          v[d].w = lv > 3 ? t.w : 0;
        }
      }
#pragma unroll
      for (int d = 0; d < PP_NUM_FEATURES; ++d)
        __builtin_amdgcn_raw_buffer_store_b128(v[d], rs, voff, d * sg.pn4_bytes, 0);
    }
  }
}

template <typename TIn, int MODE>
#ifndef PP_EMIT_MINWAVES
#define PP_EMIT_MINWAVES 4
#endif
__global__ __launch_bounds__(kEmitThreads, PP_EMIT_MINWAVES) void k_emit(EmitArgs a) {
  __shared__ WaveLds<TIn> lds[kEmitWaves];
  using Rec = typename Rec4<TIn>::type;
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int N = a.N, P = a.P;
  // (0) hand the count/cursor array back zeroed for the next call
  {
    int4 *c4 = reinterpret_cast<int4 *>(a.cursor + (int64_t)b * a.g.ncells_pad);
    const int n4 = a.g.ncells_pad >> 2;
    for (int i = blockIdx.x * kEmitThreads + tid; i < n4; i += gridDim.x * kEmitThreads)
      c4[i] = make_int4(0, 0, 0, 0);
  }
  WaveLds<TIn> &L = lds[w];
  // wave-uniform by construction; readfirstlane lets the compiler keep everything
  // derived from it (slab geometry, line masks, buffer offsets) in SGPRs
  const int p0 = __builtin_amdgcn_readfirstlane((blockIdx.x * kEmitWaves + w) * KW);
  if (p0 >= P) return;
  const int kw_eff = min(KW, P - p0);
  // (1) pillar descriptors (one lane per pillar)
  const int2 tot = a.totals[b];
  int4 m = make_int4(-1, 0, 0, 0);
  if (lane < KW && p0 + lane < P) m = a.pillar_meta[(int64_t)b * P + p0 + lane];
  SlabGeom sg;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(a.out + (int64_t)b * 9 * P * N), 0,
      MODE == kModeDenseVec4 ? (int)(36u * (unsigned)P * (unsigned)N) : 0, 0x00020000);
  if (MODE == kModeDenseVec4) {
    sg.N4 = (unsigned)(N >> 2);
    sg.rowf4 = (unsigned)kw_eff * sg.N4;
    sg.magic_n4 = 0xFFFFFFFFu / sg.N4 + 1u;
    sg.base0 = (unsigned)p0 * sg.N4;
    sg.pn4_bytes = (unsigned)P * sg.N4 * 16u;
    sg.aligned = ((((int64_t)P * sg.N4) & 7) == 0) && (KW * sg.N4 / 8 + 2 <= 64);
    sg.late_lines = 0;
  }
  const int npil = min(tot.x, P);
  if (lane < KW) {
    if (p0 + lane >= npil) m = make_int4(-1, 0, 0, 0);  // rows beyond the occupied cells hold stale descriptors
    L.slot[lane] = m.x;
    L.start[lane] = m.y;
    L.cnt[lane] = m.z;
    L.live[lane] = min(m.z, N);
    double cx = 0, cy = 0;
    if (m.z > 0) pillar_canvas(m.x, a.g, cx, cy);
    L.cx[lane] = cx;
    L.cy[lane] = cy;
  }
  wave_sync();
  int cnts[KW], segbeg[KW], segpad[KW];
  int T = 0, Tpad = 0;
#pragma unroll
  for (int k = 0; k < KW; ++k) {
    cnts[k] = __builtin_amdgcn_readfirstlane(L.cnt[k]);
    segbeg[k] = T;       // position in the CSR range
    segpad[k] = Tpad;    // position in the sorted LDS arrays: 4-aligned bucket starts
    T += cnts[k];
    Tpad += (cnts[k] + 3) & ~3;
  }
  const bool pooled = (Tpad <= CAPW);
  // (2) the wave's pooled bucket in one coalesced read: consecutive pillars own
  //     consecutive CSR ranges
  int idx_r[kPre];
  Rec rec_r[kPre];
#pragma unroll
  for (int it = 0; it < kPre; ++it) {
    idx_r[it] = 0;
    rec_r[it].x = rec_r[it].y = rec_r[it].z = rec_r[it].w = 0;
  }
  if (pooled && T > 0) {
    const int start0 = __builtin_amdgcn_readfirstlane(L.start[0]);
    const int *sidx = a.sorted_idx + (int64_t)b * a.ncap + start0;
    const Rec *srec = reinterpret_cast<const Rec *>(a.sorted_pts) + (int64_t)b * a.ncap + start0;
#pragma unroll
    for (int it = 0; it < kPre; ++it)
      if (lane + it * kWave < T) {
        idx_r[it] = sidx[lane + it * kWave];
        rec_r[it] = srec[lane + it * kWave];
      }
  }
  // (3) scatter indices; dense modes: the zero padding that needs no point data
  float *outb = nullptr;
  if (MODE != kModeCompact) {
    if (lane < kw_eff) {
      long long *io = a.idx_out + ((int64_t)b * P + p0 + lane) * 3;
      long long i0 = 0, i1 = 0, i2 = 0;
      if (L.cnt[lane] > 0) {
        i0 = 1;                        // pillars.cpp:390
        i1 = (long long)L.cx[lane];    // pillars.cpp:391 + dataset.py:106 (.long())
        i2 = (long long)L.cy[lane];    // pillars.cpp:392
      }
      io[0] = i0;
      io[1] = i1;
      io[2] = i2;
    }
  }
  if (MODE == kModeDenseVec4 || MODE == kModeDenseScalar) {
    outb = a.out + (int64_t)b * 9 * P * N;
    if (MODE == kModeDenseVec4) {
      sg.pooled = 0;
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        sg.nh[k] = (min(cnts[k], N) + 3) >> 2;
        // pooled together, or alone in its own pass when the pool overflowed
        if (cnts[k] > 0 && (pooled || cnts[k] <= CAPW)) sg.pooled |= 1u << k;
      }
      // Everything fits the LDS pool (the normal case): ONE store pass after the point
      // phase writes every line whole.  (Measured: a separate zero pass before the
      // bucket data arrives plus a late pass for the live lines costs 10 us of 34 at
      // 4 sweeps per launch -- the second pass's issue slots, not its bytes.)
      if (!pooled || T == 0) {
        sg.late_lines = slab_late_lines(sg, sg.pooled);
        store_slab<kPassEarly, TIn>(L, sg, rs, lane, segbeg);
      }
    } else {
      const int rowf = kw_eff * N;
      const int total = 9 * rowf;
      for (int f = lane; f < total; f += kWave) {
        const int d = f / rowf, rem = f - d * rowf;
        const int k = rem / N, n = rem - k * N;
        if (n >= L.live[k]) outb[((int64_t)d * P + p0) * N + rem] = 0.0f;
      }
    }
  }
  // fused feature net: this lane's output channel
  PfnAcc acc;
  if constexpr (MODE == kModePfn) {
    const float *wp = a.pfn_w + lane * 12;
#pragma unroll
    for (int d = 0; d < 9; ++d) acc.w[d] = wp[d];
    acc.bias = wp[9];
    acc.scale = wp[10];
    acc.shift = wp[11];
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      acc.rmax[k] = -INFINITY;
      acc.rmin[k] = INFINITY;
    }
  }
  // folds the live points [sb, sb+live) of pillar k (features in L.u.feat) into acc
  auto pfn_fold = [&](int k, int sb, int live) {
    float mx = -INFINITY, mn = INFINITY;
    for (int j = sb; j < sb + live; ++j) {
      float xv[9];
#pragma unroll
      for (int d = 0; d < 9; ++d) xv[d] = L.u.feat[d][j];
      const float rr = pfn_relu_conv(acc, xv);
      mx = fmaxf(mx, rr);
      mn = fminf(mn, rr);
    }
#pragma unroll
    for (int kk = 0; kk < KW; ++kk)
      if (kk == k) {
        acc.rmax[kk] = fmaxf(acc.rmax[kk], mx);
        acc.rmin[kk] = fminf(acc.rmin[kk], mn);
      }
  };
  auto pfn_finish = [&]() {
    // zero-padded slots take part in the max (model/model.py:36-39)
    const float rpad = fmaxf(acc.bias, 0.0f);
    float yv[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      float mx = acc.rmax[k], mn = acc.rmin[k];
      if (min(cnts[k], N) < N) {
        mx = fmaxf(mx, rpad);
        mn = fminf(mn, rpad);
      }
      yv[k] = fmaf(acc.scale >= 0.0f ? mx : mn, acc.scale, acc.shift);
    }
    if (a.pfn_out) {
      float *o = a.pfn_out + ((int64_t)b * kPfnChannels + lane) * P + p0;
      if (kw_eff == KW && (P & 3) == 0) {
        *reinterpret_cast<float4 *>(o) = make_float4(yv[0], yv[1], yv[2], yv[3]);
      } else {
#pragma unroll
        for (int k = 0; k < KW; ++k)
          if (k < kw_eff) o[k] = yv[k];
      }
    }
    if (a.canvas) {
      // out[b, :, row, col] = x[b, :, p] for the flagged pillars only (model/model.py:56-61);
      // channels last: the 64 lanes write one 256-byte pixel
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        if (k >= kw_eff || cnts[k] == 0) continue;
        double cx, cy;
        pillar_canvas(L.slot[k], a.g, cx, cy);
        const int64_t col = (int64_t)cx, row = (int64_t)cy;
        if (row < 0 || row >= a.canvas_h || col < 0 || col >= a.canvas_w) continue;
        if (a.canvas_nhwc)
          a.canvas[((((int64_t)b * a.canvas_h + row) * a.canvas_w) + col) * kPfnChannels + lane] = yv[k];
        else
          a.canvas[((((int64_t)b * kPfnChannels + lane) * a.canvas_h) + row) * a.canvas_w + col] = yv[k];
      }
    }
  };
  // (4) the points
  if (T == 0) {
    if constexpr (MODE == kModePfn) pfn_finish();
    return;
  }
  if (pooled) {
    emit_group<TIn, MODE>(L, a, b, p0, 0, KW, lane, idx_r, rec_r, segbeg, segpad, cnts, T);
    if (MODE == kModeDenseVec4) store_slab<kPassAll, TIn>(L, sg, rs, lane, segpad);
    if constexpr (MODE == kModePfn) {
#pragma unroll
      for (int k = 0; k < KW; ++k) pfn_fold(k, segpad[k], min(cnts[k], N));
    }
  } else {
    // the pool overflowed: greedy runs of consecutive pillars that fit the pool,
    // a pillar beyond the pool on its own through the ballot re-scan
    int k = 0;
#pragma unroll 1
    while (k < KW) {
      if (cnts[k] == 0) {
        ++k;
        continue;
      }
      if (cnts[k] > CAPW) {
        if constexpr (MODE == kModePfn) {
          float mx = -INFINITY, mn = INFINITY;
          emit_big_pillar<TIn, MODE>(L, a, b, k, p0 + k, lane, &acc, &mx, &mn);
#pragma unroll
          for (int kk = 0; kk < KW; ++kk)
            if (kk == k) {
              acc.rmax[kk] = fmaxf(acc.rmax[kk], mx);
              acc.rmin[kk] = fminf(acc.rmin[kk], mn);
            }
        } else {
          emit_big_pillar<TIn, MODE>(L, a, b, k, p0 + k, lane);
        }
        wave_sync();
        ++k;
        continue;
      }
      const int kb = k;
      int gpad = 0, graw = 0;
      int sbg[KW], spg[KW], ckg[KW];
#pragma unroll
      for (int kk = 0; kk < KW; ++kk) {
        sbg[kk] = 0;
        spg[kk] = 0;
        ckg[kk] = 0;
      }
      while (k < KW && cnts[k] <= CAPW && gpad + ((cnts[k] + 3) & ~3) <= CAPW) {
#pragma unroll
        for (int kk = 0; kk < KW; ++kk)
          if (kk == k) {
            sbg[kk] = graw;
            spg[kk] = gpad;
            ckg[kk] = cnts[kk];
          }
        graw += cnts[k];
        gpad += (cnts[k] + 3) & ~3;
        ++k;
      }
      unsigned gmask = 0;
#pragma unroll
      for (int kk = 0; kk < KW; ++kk)
        if (ckg[kk] > 0) gmask |= 1u << kk;
      const int st = __builtin_amdgcn_readfirstlane(L.start[kb]);  // kb is occupied
      const int *sidx = a.sorted_idx + (int64_t)b * a.ncap + st;
      const Rec *srec = reinterpret_cast<const Rec *>(a.sorted_pts) + (int64_t)b * a.ncap + st;
#pragma unroll
      for (int it = 0; it < kPre; ++it)
        if (lane + it * kWave < graw) {
          idx_r[it] = sidx[lane + it * kWave];
          rec_r[it] = srec[lane + it * kWave];
        }
      emit_group<TIn, MODE>(L, a, b, p0, kb, k, lane, idx_r, rec_r, sbg, spg, ckg, graw);
      if constexpr (MODE == kModePfn) {
#pragma unroll
        for (int kk = 0; kk < KW; ++kk)
          if (ckg[kk] > 0) pfn_fold(kk, spg[kk], min(ckg[kk], N));
        wave_sync();
      }
      if (MODE == kModeDenseVec4) {
        // late pass restricted to this run's lines: a line shared with a pillar of
        // another run is completed by that run's own pass (both write identical
        // zeros outside their own head groups)
        SlabGeom s1 = sg;
        s1.pooled = gmask;
        s1.late_lines = slab_late_lines(sg, gmask);
        store_slab<kPassLate, TIn>(L, s1, rs, lane, spg);
        wave_sync();
      }
    }
  }
  if constexpr (MODE == kModePfn) pfn_finish();
}

// ------------------------------------------------------------------------- //
// host side                                                                  //
// ------------------------------------------------------------------------- //
static unsigned long long gcd_u64(unsigned long long a, unsigned long long b) {
  while (b) {
    unsigned long long t = a % b;
    a = b;
    b = t;
  }
  return a;
}

static unsigned long long modinv(unsigned long long a, unsigned long long m) {
  // extended Euclid on signed 128-bit-safe ranges (m < 2^31)
  long long t = 0, nt = 1, r = (long long)m, nr = (long long)(a % m);
  while (nr != 0) {
    long long q = r / nr;
    long long tmp = t - q * nt;
    t = nt;
    nt = tmp;
    tmp = r - q * nr;
    r = nr;
    nr = tmp;
  }
  if (t < 0) t += (long long)m;
  return (unsigned long long)t;
}

int make_grid(const pp_voxel_params_t *prm, GridGeom *g) {
  if (!prm) {
    set_error("voxel params are NULL");
    return PP_ERR_VALUE;
  }
  if (!(prm->x_step > 0.0) || !(prm->y_step > 0.0) || !(prm->x_max > prm->x_min) ||
      !(prm->y_max > prm->y_min)) {
    set_error("invalid grid: steps must be > 0 and max > min");
    return PP_ERR_VALUE;
  }
  if (prm->order != PP_ORDER_ROW_MAJOR && prm->order != PP_ORDER_SCRAMBLED) {
    set_error("unknown pillar order %d", prm->order);
    return PP_ERR_VALUE;
  }
  // same bound as the oracle: floor((x-x_min)/x_step) <= floor((x_max-x_min)/x_step)
  const double qx = std::floor((prm->x_max - prm->x_min) / prm->x_step);
  const double qy = std::floor((prm->y_max - prm->y_min) / prm->y_step);
  if (!(qx < 32768.0) || !(qy < 32768.0)) {
    set_error("cell grid too large (%g x %g cells; limit 32768 per axis)", qx + 1, qy + 1);
    return PP_ERR_VALUE;
  }
  g->x_step = prm->x_step;
  g->y_step = prm->y_step;
  g->x_min = prm->x_min;
  g->y_min = prm->y_min;
  g->z_min = prm->z_min;
  g->x_max = prm->x_max;
  g->y_max = prm->y_max;
  g->z_max = prm->z_max;
  g->canvas_height = prm->canvas_height;
  g->nx = (int)qx + 1;
  g->ny = (int)qy + 1;
  const long long nc = (long long)g->nx * g->ny;
  if (nc >= (1ll << 30)) {
    set_error("cell grid too large (%lld cells)", nc);
    return PP_ERR_VALUE;
  }
  g->ncells = (int)nc;
  g->ncells_pad = (int)((nc + kScanTile - 1) / kScanTile * kScanTile);
  g->order = prm->order;
  g->mult = 1;
  g->mult_inv = 1;
  if (prm->order == PP_ORDER_SCRAMBLED && nc > 2) {
    unsigned long long m = (unsigned long long)std::floor((double)nc * 0.6180339887498949);
    if (m < 1) m = 1;
    while (gcd_u64(m, (unsigned long long)nc) != 1) ++m;
    m %= (unsigned long long)nc;
    g->mult = m;
    g->mult_inv = modinv(m, (unsigned long long)nc);
  }
  return PP_OK;
}

namespace {

struct VoxLayout {
  size_t cursor, cell_rank, sorted_idx, sorted_pts, meta, status, ticket, totals, errflag, bytes;
  int nwg_scan;
  int ncap;
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

VoxLayout vox_layout(int B, int64_t max_points, const GridGeom &g, int P, int rec_bytes) {
  VoxLayout l;
  l.ncap = (int)align_up((size_t)std::max<int64_t>(max_points, 1), 64);
  l.nwg_scan = g.ncells_pad / kScanTile;
  size_t off = 0;
  l.cursor = off;
  off = align_up(off + (size_t)B * g.ncells_pad * 4, 256);
  l.cell_rank = off;
  off = align_up(off + (size_t)B * l.ncap * 8, 256);
  l.sorted_idx = off;
  off = align_up(off + (size_t)B * l.ncap * 4, 256);
  l.sorted_pts = off;
  off = align_up(off + (size_t)B * l.ncap * rec_bytes, 256);
  l.meta = off;
  off = align_up(off + (size_t)B * P * 16, 256);
  l.status = off;
  off = align_up(off + (size_t)B * l.nwg_scan * 8, 256);
  l.ticket = off;
  off = align_up(off + (size_t)B * 4, 256);
  l.totals = off;
  off = align_up(off + (size_t)B * 8, 256);
  l.errflag = off;
  off = align_up(off + 4, 256);
  l.bytes = off;
  return l;
}

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess) {
      ok = true;
      if (prev != dev) (void)hipSetDevice(dev);
    } else {
      (void)hipGetLastError();
    }
  }
  ~DeviceGuard() {
    if (ok) (void)hipSetDevice(prev);
  }
};

// Makes the workspace fit (B, max_points, grid, P) and guarantees the "clean"
// invariant (cursor, status, ticket all zero) whenever the layout changed.
int prepare_ws(pp_ctx *ctx, hipStream_t stream, int B, int64_t max_points,
               const GridGeom &g, int P, int rec_bytes, VoxLayout *out) {
  VoxLayout l = vox_layout(B, max_points, g, P, rec_bytes);
  const unsigned long long key[6] = {(unsigned long long)B, (unsigned long long)l.ncap,
                                     (unsigned long long)g.ncells_pad,
                                     (unsigned long long)P, (unsigned long long)l.bytes,
                                     (unsigned long long)rec_bytes};
  bool grew = false;
  int rc = ctx->vox_ws.ensure(l.bytes, &grew);
  if (rc) return rc;
  if (grew || std::memcmp(key, ctx->vox_layout_key, sizeof key) != 0) {
    PP_HIP_TRY(hipMemsetAsync(ctx->vox_ws.ptr, 0, ctx->vox_ws.bytes, stream));
    std::memcpy(ctx->vox_layout_key, key, sizeof key);
  }
  *out = l;
  return PP_OK;
}

template <typename TIn>
int launch_pipeline(pp_ctx *ctx, hipStream_t stream, const TIn *pts, int64_t sweep_stride,
                    int64_t s0, int64_t s1, int contig, const NPoints &np, int B,
                    int maxn, const GridGeom &g, int P, int N, const VoxLayout &l,
                    int mode, float *out, long long *idx_out, double *feat_out,
                    bool timed, const float *pfn_w = nullptr, float *pfn_out = nullptr,
                    float *canvas = nullptr, int canvas_h = 0, int canvas_w = 0,
                    int canvas_nhwc = 0) {
  char *ws = static_cast<char *>(ctx->vox_ws.ptr);
  int *cursor = reinterpret_cast<int *>(ws + l.cursor);
  int2 *cell_rank = reinterpret_cast<int2 *>(ws + l.cell_rank);
  int *sorted_idx = reinterpret_cast<int *>(ws + l.sorted_idx);
  typename Rec4<TIn>::type *sorted_pts =
      reinterpret_cast<typename Rec4<TIn>::type *>(ws + l.sorted_pts);
  int4 *meta = reinterpret_cast<int4 *>(ws + l.meta);
  u64 *status = reinterpret_cast<u64 *>(ws + l.status);
  unsigned *ticket = reinterpret_cast<unsigned *>(ws + l.ticket);
  int2 *totals = reinterpret_cast<int2 *>(ws + l.totals);
  int *errflag = reinterpret_cast<int *>(ws + l.errflag);

  const dim3 grid_pts((unsigned)std::max(1, (maxn + kBinThreads - 1) / kBinThreads), (unsigned)B);
  hipLaunchKernelGGL((k_bin_count<TIn>), grid_pts, dim3(kBinThreads), 0, stream, pts,
                     sweep_stride, s0, s1, contig, np, g, cell_rank, l.ncap, cursor);
  hipLaunchKernelGGL(k_scan, dim3((unsigned)l.nwg_scan, (unsigned)B), dim3(kScanThreads), 0,
                     stream, cursor, g.ncells_pad, P, meta, status, ticket, totals, errflag);
  hipLaunchKernelGGL((k_fill<TIn>), grid_pts, dim3(kBinThreads), 0, stream, pts, sweep_stride,
                     cell_rank, l.ncap, np, cursor, g.ncells_pad, sorted_pts, sorted_idx, status,
                     l.nwg_scan, ticket);
  EmitArgs a;
  a.g = g;
  a.np = np;
  a.P = P;
  a.N = N;
  a.ncap = l.ncap;
  a.pillar_meta = meta;
  a.totals = totals;
  a.sorted_idx = sorted_idx;
  a.sorted_pts = sorted_pts;
  a.cell_rank = cell_rank;
  a.cursor = cursor;
  a.pts = pts;
  a.sweep_stride = sweep_stride;
  a.out = out;
  a.idx_out = idx_out;
  a.feat_out = feat_out;
  a.pfn_w = pfn_w;
  a.pfn_out = pfn_out;
  a.canvas = canvas;
  a.canvas_h = canvas_h;
  a.canvas_w = canvas_w;
  a.canvas_nhwc = canvas_nhwc;
  const dim3 grid_emit((unsigned)((P + KW * kEmitWaves - 1) / (KW * kEmitWaves)), (unsigned)B);
  // When the timing ring is armed the emit launch carries its own start/stop events
  // (hipExtLaunchKernelGGL binds them to the dispatch packet, so the pair brackets the
  // kernel alone, like a profiler's kernel trace, not the gaps around it).
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (timed && ctx->ev_slots > 0) {
    ev0 = ctx->ev_start[ctx->ev_next];
    ev1 = ctx->ev_stop[ctx->ev_next];
    ctx->ev_next = (ctx->ev_next + 1) % ctx->ev_slots;
    ctx->ev_count = std::min(ctx->ev_count + 1, ctx->ev_slots);
  }
  switch (mode) {
    case kModeDenseVec4:
      hipExtLaunchKernelGGL((k_emit<TIn, kModeDenseVec4>), grid_emit, dim3(kEmitThreads), 0, stream,
                            ev0, ev1, 0, a);
      break;
    case kModeDenseScalar:
      hipExtLaunchKernelGGL((k_emit<TIn, kModeDenseScalar>), grid_emit, dim3(kEmitThreads), 0, stream,
                            ev0, ev1, 0, a);
      break;
    case kModePfn:
      hipExtLaunchKernelGGL((k_emit<TIn, kModePfn>), grid_emit, dim3(kEmitThreads), 0, stream, ev0,
                            ev1, 0, a);
      break;
    default:
      hipExtLaunchKernelGGL((k_emit<TIn, kModeCompact>), grid_emit, dim3(kEmitThreads), 0, stream,
                            ev0, ev1, 0, a);
      break;
  }
  {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      // the self-cleaning invariant (counts, scan words zero) can no longer be assumed
      std::memset(ctx->vox_layout_key, 0, sizeof ctx->vox_layout_key);
      set_error("voxelizer launch failed: %s", hipGetErrorString(e));
      return PP_ERR_HIP;
    }
  }
  return PP_OK;
}

}  // namespace
}  // namespace pp

using namespace pp;

extern "C" int pp_voxelize_reserve(pp_ctx_t *ctx, int batch, int64_t max_points,
                                   const pp_voxel_params_t *prm) {
  if (!ctx) {
    set_error("ctx is NULL");
    return PP_ERR_VALUE;
  }
  if (batch < 1 || batch > PP_MAX_BATCH || max_points < 0 || max_points > INT_MAX / 2) {
    set_error("reserve: batch must be in [1,%d] and 0 <= max_points < 2^30", PP_MAX_BATCH);
    return PP_ERR_VALUE;
  }
  GridGeom g;
  int rc = make_grid(prm, &g);
  if (rc) return rc;
  if (prm->max_pillars < 1) {
    set_error("max_pillars must be >= 1");
    return PP_ERR_VALUE;
  }
  DeviceGuard guard(ctx->device);
  VoxLayout l;
  return prepare_ws(ctx, nullptr, batch, max_points, g, prm->max_pillars, 16, &l);
}

extern "C" int pp_voxelize_dev(pp_ctx_t *ctx, void *stream_, const float *points_dev,
                               int64_t points_stride, const int32_t *n_points, int batch,
                               const pp_voxel_params_t *prm, float *pillars_dev,
                               int64_t *indices_dev, int32_t *num_cells_dev) {
  if (!ctx || !points_dev || !n_points || !prm || !pillars_dev || !indices_dev) {
    set_error("pp_voxelize_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  if (batch < 1 || batch > PP_MAX_BATCH) {
    set_error("batch must be in [1,%d], got %d", PP_MAX_BATCH, batch);
    return PP_ERR_VALUE;
  }
  const int P = prm->max_pillars, N = prm->max_points_per_pillar;
  if (P < 1 || N < 1 || N > 65536 || (long long)P * N > 100000000ll) {
    set_error("need 1 <= max_pillars, 1 <= max_points_per_pillar <= 65536 and P*N <= 1e8 "
              "(got P=%d N=%d)", P, N);
    return PP_ERR_VALUE;
  }
  if (points_stride < 0 || points_stride > INT_MAX / 2) {
    set_error("points_stride out of range");
    return PP_ERR_VALUE;
  }
  NPoints np;
  std::memset(&np, 0, sizeof np);
  int maxn = 0;
  for (int b = 0; b < batch; ++b) {
    if (n_points[b] < 0 || n_points[b] > points_stride) {
      set_error("n_points[%d]=%d outside [0, points_stride=%lld]", b, n_points[b],
                (long long)points_stride);
      return PP_ERR_VALUE;
    }
    np.n[b] = n_points[b];
    maxn = std::max(maxn, n_points[b]);
  }
  if ((reinterpret_cast<uintptr_t>(points_dev) & 15) || (reinterpret_cast<uintptr_t>(pillars_dev) & 15) ||
      (reinterpret_cast<uintptr_t>(indices_dev) & 7)) {
    set_error("device pointers must be 16-byte (points, pillars) / 8-byte (indices) aligned");
    return PP_ERR_VALUE;
  }
  GridGeom g;
  int rc = make_grid(prm, &g);
  if (rc) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DeviceGuard guard(ctx->device);
  VoxLayout l;
  rc = prepare_ws(ctx, stream, batch, std::max<int64_t>(points_stride, 1), g, P, 16, &l);
  if (rc) return rc;
  const int mode = (N % 4 == 0 && N <= 4096) ? kModeDenseVec4 : kModeDenseScalar;
  rc = launch_pipeline<float>(ctx, stream, points_dev, points_stride, 4, 1, 1, np, batch, maxn,
                              g, P, N, l, mode, pillars_dev,
                              reinterpret_cast<long long *>(indices_dev), nullptr, true);
  if (rc) return rc;
  if (num_cells_dev) {
    char *ws = static_cast<char *>(ctx->vox_ws.ptr);
    PP_HIP_TRY(hipMemcpyAsync(num_cells_dev, ws + l.totals, (size_t)batch * 8,
                              hipMemcpyDeviceToDevice, stream));
  }
  return PP_OK;
}

static int voxelize_pfn_impl(pp_ctx_t *ctx, void *stream_, const float *points_dev,
                             int64_t points_stride, const int32_t *n_points, int batch,
                             const pp_voxel_params_t *prm, const float *pfn_params_dev,
                             int channels, float *features_dev, int64_t *indices_dev,
                             int32_t *num_cells_dev, float *canvas_dev, int canvas_h, int canvas_w,
                             int channels_last) {
  if (!ctx || !points_dev || !n_points || !prm || !pfn_params_dev || !indices_dev ||
      (!features_dev && !canvas_dev)) {
    set_error("pp_voxelize_pfn*_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  if (channels != kPfnChannels) {
    set_error("the fused feature net is built for %d output channels (got %d)", kPfnChannels, channels);
    return PP_ERR_VALUE;
  }
  if (batch < 1 || batch > PP_MAX_BATCH) {
    set_error("batch must be in [1,%d], got %d", PP_MAX_BATCH, batch);
    return PP_ERR_VALUE;
  }
  const int P = prm->max_pillars, N = prm->max_points_per_pillar;
  if (P < 1 || N < 1 || N > 65536) {
    set_error("need max_pillars >= 1 and 1 <= max_points_per_pillar <= 65536 (got P=%d N=%d)", P, N);
    return PP_ERR_VALUE;
  }
  if (points_stride < 0 || points_stride > INT_MAX / 2) {
    set_error("points_stride out of range");
    return PP_ERR_VALUE;
  }
  NPoints np;
  std::memset(&np, 0, sizeof np);
  int maxn = 0;
  for (int b = 0; b < batch; ++b) {
    if (n_points[b] < 0 || n_points[b] > points_stride) {
      set_error("n_points[%d]=%d outside [0, points_stride=%lld]", b, n_points[b],
                (long long)points_stride);
      return PP_ERR_VALUE;
    }
    np.n[b] = n_points[b];
    maxn = std::max(maxn, n_points[b]);
  }
  if ((reinterpret_cast<uintptr_t>(points_dev) & 15) || (reinterpret_cast<uintptr_t>(features_dev) & 15) ||
      (reinterpret_cast<uintptr_t>(indices_dev) & 7) || (reinterpret_cast<uintptr_t>(pfn_params_dev) & 3)) {
    set_error("device pointers must be 16-byte (points, features) / 8-byte (indices) aligned");
    return PP_ERR_VALUE;
  }
  GridGeom g;
  int rc = make_grid(prm, &g);
  if (rc) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DeviceGuard guard(ctx->device);
  VoxLayout l;
  rc = prepare_ws(ctx, stream, batch, std::max<int64_t>(points_stride, 1), g, P, 16, &l);
  if (rc) return rc;
  if (canvas_dev) {
    // rows are (H-1) - fy with H = canvas_height (pillars.cpp:280).  The cell grid carries one
    // guard row/column for the rounding of (x - x_min)/x_step at the upper edge (never
    // populated by f32 input); a pillar whose pixel is outside the canvas is not written.
    if ((double)canvas_h != g.canvas_height || canvas_h < 1 || canvas_w < 1 ||
        (reinterpret_cast<uintptr_t>(canvas_dev) & 15)) {
      set_error("canvas %dx%d: height must equal canvas_height=%g (and the tensor be 16-byte "
                "aligned)", canvas_h, canvas_w, g.canvas_height);
      return PP_ERR_VALUE;
    }
    PP_HIP_TRY(hipMemsetAsync(canvas_dev, 0,
                              (size_t)batch * kPfnChannels * canvas_h * canvas_w * sizeof(float),
                              stream));
  }
  rc = launch_pipeline<float>(ctx, stream, points_dev, points_stride, 4, 1, 1, np, batch, maxn,
                              g, P, N, l, kModePfn, nullptr,
                              reinterpret_cast<long long *>(indices_dev), nullptr, true,
                              pfn_params_dev, features_dev, canvas_dev, canvas_h, canvas_w,
                              channels_last ? 1 : 0);
  if (rc) return rc;
  if (num_cells_dev) {
    char *ws = static_cast<char *>(ctx->vox_ws.ptr);
    PP_HIP_TRY(hipMemcpyAsync(num_cells_dev, ws + l.totals, (size_t)batch * 8,
                              hipMemcpyDeviceToDevice, stream));
  }
  return PP_OK;
}

extern "C" int pp_voxelize_pfn_dev(pp_ctx_t *ctx, void *stream_, const float *points_dev,
                                   int64_t points_stride, const int32_t *n_points, int batch,
                                   const pp_voxel_params_t *prm, const float *pfn_params_dev,
                                   int channels, float *features_dev, int64_t *indices_dev,
                                   int32_t *num_cells_dev) {
  if (!features_dev) {
    set_error("pp_voxelize_pfn_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  return voxelize_pfn_impl(ctx, stream_, points_dev, points_stride, n_points, batch, prm,
                           pfn_params_dev, channels, features_dev, indices_dev, num_cells_dev,
                           nullptr, 0, 0, 0);
}

extern "C" int pp_voxelize_pfn_canvas_dev(pp_ctx_t *ctx, void *stream_, const float *points_dev,
                                          int64_t points_stride, const int32_t *n_points,
                                          int batch, const pp_voxel_params_t *prm,
                                          const float *pfn_params_dev, int channels,
                                          float *canvas_dev, int canvas_h, int canvas_w,
                                          int channels_last, int64_t *indices_dev,
                                          int32_t *num_cells_dev) {
  if (!canvas_dev) {
    set_error("pp_voxelize_pfn_canvas_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  return voxelize_pfn_impl(ctx, stream_, points_dev, points_stride, n_points, batch, prm,
                           pfn_params_dev, channels, nullptr, indices_dev, num_cells_dev,
                           canvas_dev, canvas_h, canvas_w, channels_last);
}

extern "C" int pp_create_pillars_f64(pp_ctx_t *ctx, const void *points, int64_t n_points,
                                     int64_t ps0, int64_t ps1, void *tensor,
                                     const int64_t t_shape[3], const int64_t t_strides[3],
                                     void *indices, const int64_t i_shape[2],
                                     const int64_t i_strides[2], const pp_voxel_params_t *prm,
                                     int64_t *num_cells) {
  if (!ctx || !prm || !t_shape || !t_strides || !i_shape || !i_strides ||
      (n_points > 0 && !points)) {
    set_error("pp_create_pillars_f64: NULL argument");
    return PP_ERR_VALUE;
  }
  if (n_points < 0 || n_points > INT_MAX / 2) {
    set_error("n_points out of range");
    return PP_ERR_VALUE;
  }
  GridGeom g;
  int rc = make_grid(prm, &g);
  if (rc) return rc;
  if (num_cells) *num_cells = 0;
  if (n_points == 0) return PP_OK;
  const int n = (int)n_points;
  const int max_pillars = std::max(prm->max_pillars, 0);
  const int N = std::max(prm->max_points_per_pillar, 0);
  // there are at most min(n, ncells) non-empty cells
  const int P = std::max(1, std::min(max_pillars, std::min(n, g.ncells)));
  DeviceGuard guard(ctx->device);
  hipStream_t stream = nullptr;
  VoxLayout l;
  rc = prepare_ws(ctx, stream, 1, n, g, P, 32, &l);
  if (rc) return rc;
  // gather the (arbitrarily strided) points into pinned staging: [n][4] f64
  rc = ctx->pin_in.ensure((size_t)n * 32);
  if (rc) return rc;
  rc = ctx->stage_in.ensure((size_t)l.ncap * 32);
  if (rc) return rc;
  rc = ctx->stage_out.ensure((size_t)l.ncap * 72);
  if (rc) return rc;
  {
    double *dst = static_cast<double *>(ctx->pin_in.ptr);
    const char *src = static_cast<const char *>(points);
    for (int64_t i = 0; i < n; ++i)
      for (int c = 0; c < 4; ++c)
        std::memcpy(&dst[i * 4 + c], src + i * ps0 + c * ps1, 8);
  }
  PP_HIP_TRY(hipMemcpyAsync(ctx->stage_in.ptr, ctx->pin_in.ptr, (size_t)n * 32,
                            hipMemcpyHostToDevice, stream));
  NPoints np;
  std::memset(&np, 0, sizeof np);
  np.n[0] = n;
  rc = launch_pipeline<double>(ctx, stream, static_cast<const double *>(ctx->stage_in.ptr),
                               l.ncap, 4, 1, 1, np, 1, n, g, P, N, l, kModeCompact, nullptr,
                               nullptr, static_cast<double *>(ctx->stage_out.ptr), false);
  if (rc) return rc;
  // descriptors back: totals, errflag, pillar_meta[P]
  rc = ctx->pin_meta.ensure(256 + (size_t)P * 16);
  if (rc) return rc;
  char *ws = static_cast<char *>(ctx->vox_ws.ptr);
  char *pm = static_cast<char *>(ctx->pin_meta.ptr);
  PP_HIP_TRY(hipMemcpyAsync(pm, ws + l.totals, 8, hipMemcpyDeviceToHost, stream));
  PP_HIP_TRY(hipMemcpyAsync(pm + 16, ws + l.errflag, 4, hipMemcpyDeviceToHost, stream));
  PP_HIP_TRY(hipMemcpyAsync(pm + 256, ws + l.meta, (size_t)P * 16, hipMemcpyDeviceToHost, stream));
  PP_HIP_TRY(hipStreamSynchronize(stream));
  int tot[2], err;
  std::memcpy(tot, pm, 8);
  std::memcpy(&err, pm + 16, 4);
  if (err) {
    // leave a clean workspace behind
    (void)hipMemsetAsync(ctx->vox_ws.ptr, 0, ctx->vox_ws.bytes, stream);
    (void)hipStreamSynchronize(stream);
    set_error("cell scan timed out waiting for a predecessor tile");
    return PP_ERR_INTERNAL;
  }
  if (num_cells) *num_cells = tot[0];
  const int npil = std::min(tot[0], std::min(P, max_pillars));
  if (npil == 0) return PP_OK;
  const int4 *meta = reinterpret_cast<const int4 *>(pm + 256);
  const int64_t end = (int64_t)meta[npil - 1].y + meta[npil - 1].z;
  rc = ctx->pin_out.ensure((size_t)end * 72);
  if (rc) return rc;
  if (N > 0) {
    PP_HIP_TRY(hipMemcpyAsync(ctx->pin_out.ptr, ctx->stage_out.ptr, (size_t)end * 72,
                              hipMemcpyDeviceToHost, stream));
    PP_HIP_TRY(hipStreamSynchronize(stream));
  }
  // scatter into the caller's arrays with pybind11 .mutable_at() bounds checks,
  // pillar by pillar like pillars.cpp:335-396 (nothing else is touched)
  const double *feat = static_cast<const double *>(ctx->pin_out.ptr);
  char *tp = static_cast<char *>(tensor);
  char *ip = static_cast<char *>(indices);
  for (int p = 0; p < npil; ++p) {
    const int live = std::min(meta[p].z, N);
    for (int k = 0; k < live; ++k) {
      const double *f = feat + ((int64_t)meta[p].y + k) * 9;
      for (int d = 0; d < 9; ++d) {
        if (!tensor || p >= t_shape[0] || k >= t_shape[1] || d >= t_shape[2]) {
          set_error("create_pillars: tensor index (%d,%d,%d) out of range", p, k, d);
          return PP_ERR_INDEX;
        }
        std::memcpy(tp + p * t_strides[0] + k * t_strides[1] + d * t_strides[2], &f[d], 8);
      }
    }
    if (!indices || p >= i_shape[0] || 2 >= i_shape[1]) {
      set_error("create_pillars: indices index (%d,2) out of range", p);
      return PP_ERR_INDEX;
    }
    int cell = meta[p].x;
    if (g.order != PP_ORDER_ROW_MAJOR)
      cell = (int)(((unsigned long long)cell * g.mult_inv) % (unsigned long long)g.ncells);
    const double canvas_x = (double)(cell % g.nx);
    const double fy = (double)((g.ny - 1) - cell / g.nx);
    const double canvas_y = (g.canvas_height - 1) - fy;
    const double one = 1.0;
    std::memcpy(ip + p * i_strides[0] + 0 * i_strides[1], &one, 8);
    std::memcpy(ip + p * i_strides[0] + 1 * i_strides[1], &canvas_x, 8);
    std::memcpy(ip + p * i_strides[0] + 2 * i_strides[1], &canvas_y, 8);
  }
  return PP_OK;
}
