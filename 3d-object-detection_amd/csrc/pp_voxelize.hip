// pp_voxelize.hip -- the pillar voxelizer as hand-written HIP for gfx950 (CDNA4).
//
// Replaces create_pillars (/root/reference data/pillars.cpp:236-398) and, on the
// device-resident path, the caller glue around it (data/dataset.py:88-106):
// np.zeros + create_pillars + transpose to [9,P,N] + f64->f32 + indices->int64.
//
// Pipeline (three launches, grid.y = sweep of the batch; no global atomics, no
// workgroup that waits for another one -- the cell grid only ever exists in LDS):
//   k_split  1024 points per workgroup: point -> cell slot (f64 true division +
//            floor, half-open range test: pillars.cpp:271-280), then a STABLE
//            workgroup-local multisplit by tile (a tile = 256..4096 consecutive
//            slots of the pillar order): wave ballots find each point's peers,
//            per-wave byte histograms in LDS give its position.  Writes the
//            chunk's points grouped by tile plus one {offset,count} entry per
//            (tile, chunk).
//   k_tile   one workgroup per tile: walks the tile's runs chunk by chunk (=
//            input order), LDS histogram over the tile's cells, prefix sums ->
//            bucket starts; second walk places every point at bucket start +
//            rank, rank from ballots and per-wave byte counts -> buckets hold
//            their points in INPUT order (pillars.cpp:98 push_back order).
//            The tile's CSR offset is the sum of its runs' offsets in their
//            chunks; its occupied cells' descriptors go to a tile-local list.
//   k_emit   one wave per 4 consecutive pillars: prefix sum over the tiles'
//            occupied-cell counts (-> which tile's list holds pillar p),
//            coalesced bucket read, LDS-staged, sequential running mean
//            (pillars.cpp:311-328), N-cap, 9 features (pillars.cpp:30-31,48-56,
//            381-383), dense [9,P,N] f32 store incl. the zero padding, [P,3]
//            int64 indices
//
//
// Streaming form (pp_voxelize_step_dev): the same stages -- plus an ORDER stage that copies the
// tiles' descriptor lists into pillar order one launch ahead -- of FOUR consecutive batches as
// roles of ONE launch per call (k_step): split(batch i) | tile(i-1) | order(i-2) | emit(i-3),
// the launch boundary being the only synchronisation.
//
// The path is HBM-bound: 97% of the bytes are the dense store of
// k_emit.  Every 128-byte line of the output is written once, whole.
// All arithmetic that decides a value is f64 with contraction disabled
// (-ffp-contract=off), matching the reference's x86-64 build.

#include "pp_common.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstring>

namespace pp {

// ------------------------------------------------------------------------- //
// constants                                                                  //
// ------------------------------------------------------------------------- //
constexpr int kWave = 64;
#ifndef PP_CHUNK
#define PP_CHUNK 1024
#endif
constexpr int kChunk = PP_CHUNK;             // points per split workgroup, one per thread
constexpr int kSplitThreads = kChunk;
constexpr int kSplitWaves = kSplitThreads / kWave;
constexpr int kMinTileSlots = 256, kMaxTileSlots = 4096;  // a tile's cells live in LDS (20 B each)
constexpr int kTargetTiles = 256;            // tiles (= k_tile workgroups, split bins) aimed at
constexpr int kMaxTiles = 4096;              // split bins: byte histograms [16][T] must fit LDS
constexpr int kEmitWaves = 4;                // waves per emit workgroup
constexpr int kEmitThreads = kEmitWaves * kWave;
#ifndef PP_KW
#define PP_KW 4
#endif
constexpr int KW = PP_KW;  // pillars per emit wave
#ifndef PP_CAPW
#define PP_CAPW (PP_KW >= 2 ? 32 * PP_KW : 64)
#endif
constexpr int CAPW = PP_CAPW;  // pooled bucket capacity (points, 4-padded per pillar) per emit wave

using u64 = unsigned long long;

struct NPoints {
  int n[PP_MAX_BATCH];
};

__device__ __forceinline__ void wave_sync() {
  // LDS operations of one wave execute in program order; this only stops the
  // compiler from moving LDS accesses across a cross-lane hand-off.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// x mod ncells for x < 2^60 without a 64-bit division: q = floor(x * floor(2^64/n) / 2^64)
// is the quotient or one less (Barrett), so at most one correction.
__device__ __forceinline__ int mod_ncells(u64 x, const GridGeom &g) {
  const u64 q = __umul64hi(x, g.barrett);
  u64 r = x - q * (u64)g.ncells;
  if (r >= (u64)g.ncells) r -= (u64)g.ncells;
  return (int)r;
}
__device__ __forceinline__ int cell_to_slot(int cell, const GridGeom &g) {
  if (g.order == PP_ORDER_ROW_MAJOR) return cell;
  return mod_ncells((u64)cell * g.mult, g);
}
__device__ __forceinline__ int slot_to_cell(int slot, const GridGeom &g) {
  if (g.order == PP_ORDER_ROW_MAJOR) return slot;
  return mod_ncells((u64)slot * g.mult_inv, g);
}

template <typename T> struct Rec4;
template <> struct Rec4<float> { using type = float4; };
template <> struct Rec4<double> { using type = double4; };

// ------------------------------------------------------------------------- //
// k_split                                                                     //
// ------------------------------------------------------------------------- //
template <typename T>
__device__ __forceinline__ typename Rec4<T>::type load_point(const T *pts, int64_t row,
                                                             int64_t s0, int64_t s1, bool contig) {
  typename Rec4<T>::type v;
  if (contig) {
    v = reinterpret_cast<const typename Rec4<T>::type *>(pts)[row];  // 16/32 B per lane, coalesced
  } else {
    const T *p = pts + row * s0;
    v.x = p[0]; v.y = p[s1]; v.z = p[2 * s1]; v.w = p[3 * s1];
  }
  return v;
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

__device__ __forceinline__ u64 lanes_below(int lane) {
  return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// Lanes of this wave that hold the same key (the low `bits` bits), among the
// lanes in `valid`.  One ballot per key bit.
__device__ __forceinline__ u64 wave_peers(unsigned key, int bits, bool valid) {
  u64 peers = __ballot(valid);
  for (int k = 0; k < bits; ++k) {
    const bool bit = (key >> k) & 1u;
    const u64 bk = __ballot(bit);
    peers &= bit ? bk : ~bk;
  }
  return peers;
}


#ifdef PP_STAMPS  // tools/lab builds: PP_STAMPS=1 stamps k_tile, =2 k_split, =3 k_emit (8 stamps per wave)
#define PP_STAMP_AT(which, k)                                                        \
  do {                                                                               \
    if (PP_STAMPS == (which) && stamps && lane == 0)                                  \
      stamps[(((size_t)stamp_by * stamp_nx + stamp_bx) * 16 + w) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define PP_STAMP_AT(which, k) do {} while (0)
#endif
#define PP_STAMP(k) PP_STAMP_AT(1, k)
#define PP_STAMP_S(k) PP_STAMP_AT(2, k)
#define PP_STAMP_E(k) PP_STAMP_AT(3, k)

// LDS of k_split: byte histograms [kSplitWaves][Tp], bin offsets u16 [Tp], wave totals.
__host__ __device__ inline int split_tp(int ntiles) { return (ntiles + 3) & ~3; }
__host__ __device__ inline size_t split_lds_bytes(int ntiles) {
  const size_t Tp = (size_t)split_tp(ntiles);
  return kSplitWaves * Tp + 2 * Tp + 4 * kSplitWaves;
}

template <typename T>
__global__ __launch_bounds__(kSplitThreads) void k_split(
    const T *__restrict__ pts, int64_t sweep_stride, int64_t s0, int64_t s1, int contig,
    NPoints np, GridGeom g, int ncap, int nchunks_cap, int *__restrict__ kslot,
    typename Rec4<T>::type *__restrict__ kpts, int2 *__restrict__ mat, u64 *stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char split_smem[];
  using Rec = typename Rec4<T>::type;
  const int b = blockIdx.y;
  const int n = np.n[b];
  const int chunk = blockIdx.x;
  if (chunk * kChunk >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  [[maybe_unused]] const int stamp_nx = gridDim.x, stamp_bx = blockIdx.x, stamp_by = blockIdx.y;
  PP_STAMP_S(0);
  const int ntiles = g.ntiles;
  const int Tp = split_tp(ntiles);
  unsigned char *whist = split_smem;
  unsigned short *binoff = reinterpret_cast<unsigned short *>(split_smem + kSplitWaves * Tp);
  unsigned *wtot = reinterpret_cast<unsigned *>(split_smem + kSplitWaves * Tp + 2 * Tp);
  for (int i = tid; i < kSplitWaves * Tp / 4; i += kSplitThreads)
    reinterpret_cast<unsigned *>(whist)[i] = 0u;
  // the point, its slot and tile
  const int i = chunk * kChunk + tid;
  Rec rec;
  rec.x = rec.y = rec.z = rec.w = 0;
  int slot = -1;
  if (i < n) {
    rec = load_point<T>(pts + (int64_t)b * sweep_stride * 4, i, s0, s1, contig != 0);
    const int cell = point_cell((double)rec.x, (double)rec.y, (double)rec.z, g);
    if (cell >= 0) slot = cell_to_slot(cell, g);
  }
  const bool valid = slot >= 0;
  const unsigned tile = valid ? (unsigned)slot >> g.tile_shift : 0u;
  const u64 peers = wave_peers(tile, g.tile_bits, valid);
  const int rank_w = __popcll(peers & lanes_below(lane));
  PP_STAMP_S(1);
  __syncthreads();
  PP_STAMP_S(2);
  if (valid && rank_w == 0) whist[w * Tp + tile] = (unsigned char)__popcll(peers);  // <= 64
  __syncthreads();
  PP_STAMP_S(3);
  // bin totals over the 16 waves, exclusive scan over the bins (consecutive bins per thread)
  const int nb = (ntiles + kSplitThreads - 1) / kSplitThreads;  // <= kMaxTiles / 1024 = 4
  unsigned tot[4] = {0u, 0u, 0u, 0u};
  unsigned mine = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int bin = tid * nb + e;
    if (e < nb && bin < ntiles) {
      unsigned t = 0;
#pragma unroll
      for (int ww = 0; ww < kSplitWaves; ++ww) t += whist[ww * Tp + bin];
      tot[e] = t;
      mine += t;
    }
  }
  const int inc = (int)wave_scan_u32(mine);
  if (lane == kWave - 1) wtot[w] = (unsigned)inc;
  PP_STAMP_S(4);
  __syncthreads();
  unsigned base = (unsigned)inc - mine;
#pragma unroll
  for (int ww = 0; ww < kSplitWaves; ++ww)
    if (ww < w) base += wtot[ww];
  int2 *mrow = mat + (int64_t)b * ntiles * nchunks_cap + chunk;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int bin = tid * nb + e;
    if (e < nb && bin < ntiles) {
      binoff[bin] = (unsigned short)base;  // < 1024
      mrow[(int64_t)bin * nchunks_cap] = make_int2((int)base, (int)tot[e]);
      base += tot[e];
    }
  }
  PP_STAMP_S(5);
  __syncthreads();
  PP_STAMP_S(6);
  if (valid) {
    unsigned pos = binoff[tile] + (unsigned)rank_w;
    for (int ww = 0; ww < w; ++ww) pos += whist[ww * Tp + tile];
    const int64_t dst = (int64_t)b * ncap + (int64_t)chunk * kChunk + pos;
    kslot[dst] = slot;
    kpts[dst] = rec;
  }
  PP_STAMP_S(7);
}

// k_split's work for a workgroup of PW < 16 physical waves (k_step's split role): the chunk is still
// 1024 points = 16 "virtual waves" of 64 (the unit of the ballots and of the byte histogram's rows,
// so `mat` and the grouped arrays come out exactly as k_split leaves them); physical wave w walks
// the virtual waves [w*R, w*R + R), R = 16 / PW, one point per lane and virtual wave.
template <typename T, int PW>
__device__ __forceinline__ void split_body(
    const T *__restrict__ pts, int64_t sweep_stride, int64_t s0, int64_t s1, int contig,
    const NPoints &np, const GridGeom &g, int ncap, int nchunks_cap, int *__restrict__ kslot,
    typename Rec4<T>::type *__restrict__ kpts, int2 *__restrict__ mat, unsigned char *split_smem,
    int chunk, int b) {
  using Rec = typename Rec4<T>::type;
  constexpr int THREADS = PW * kWave;
  constexpr int R = kSplitWaves / PW;
  static_assert(kSplitWaves % PW == 0, "virtual waves per physical wave");
  const int n = np.n[b];
  if (chunk * kChunk >= n) return;  // the whole workgroup
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ntiles = g.ntiles;
  const int Tp = split_tp(ntiles);
  unsigned char *whist = split_smem;
  unsigned short *binoff = reinterpret_cast<unsigned short *>(split_smem + kSplitWaves * Tp);
  unsigned *wtot = reinterpret_cast<unsigned *>(split_smem + kSplitWaves * Tp + 2 * Tp);
  for (int i = tid; i < kSplitWaves * Tp / 4; i += THREADS) reinterpret_cast<unsigned *>(whist)[i] = 0u;
  Rec rec[R];
  int slot[R], rank_w[R], cnt_w[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = chunk * kChunk + (w * R + r) * kWave + lane;
    rec[r].x = rec[r].y = rec[r].z = rec[r].w = 0;
    slot[r] = -1;
    if (i < n) {
      rec[r] = load_point<T>(pts + (int64_t)b * sweep_stride * 4, i, s0, s1, contig != 0);
      const int cell = point_cell((double)rec[r].x, (double)rec[r].y, (double)rec[r].z, g);
      if (cell >= 0) slot[r] = cell_to_slot(cell, g);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool valid = slot[r] >= 0;
    const unsigned tile = valid ? (unsigned)slot[r] >> g.tile_shift : 0u;
    const u64 peers = wave_peers(tile, g.tile_bits, valid);
    rank_w[r] = __popcll(peers & lanes_below(lane));
    cnt_w[r] = __popcll(peers);
  }
  __syncthreads();  // histogram zeroed
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (slot[r] >= 0 && rank_w[r] == 0)
      whist[(w * R + r) * Tp + ((unsigned)slot[r] >> g.tile_shift)] = (unsigned char)cnt_w[r];  // <= 64
  __syncthreads();
  // bin totals over the 16 virtual waves, exclusive scan over the bins (consecutive bins per thread)
  const int nb = (ntiles + THREADS - 1) / THREADS;
  unsigned mine = 0;
  for (int e = 0; e < nb; ++e) {
    const int bin = tid * nb + e;
    if (bin < ntiles) {
#pragma unroll
      for (int vv = 0; vv < kSplitWaves; ++vv) mine += whist[vv * Tp + bin];
    }
  }
  const int inc = (int)wave_scan_u32(mine);
  if (lane == kWave - 1) wtot[w] = (unsigned)inc;
  __syncthreads();
  unsigned base = (unsigned)inc - mine;
#pragma unroll
  for (int ww = 0; ww < PW; ++ww)
    if (ww < w) base += wtot[ww];
  int2 *mrow = mat + (int64_t)b * ntiles * nchunks_cap + chunk;
  for (int e = 0; e < nb; ++e) {
    const int bin = tid * nb + e;
    if (bin < ntiles) {
      unsigned t = 0;
#pragma unroll
      for (int vv = 0; vv < kSplitWaves; ++vv) t += whist[vv * Tp + bin];
      binoff[bin] = (unsigned short)base;  // < 1024
      mrow[(int64_t)bin * nchunks_cap] = make_int2((int)base, (int)t);
      base += t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (slot[r] >= 0) {
      const unsigned tile = (unsigned)slot[r] >> g.tile_shift;
      unsigned pos = binoff[tile] + (unsigned)rank_w[r];
      for (int vv = 0; vv < w * R + r; ++vv) pos += whist[vv * Tp + tile];
      const int64_t dst = (int64_t)b * ncap + (int64_t)chunk * kChunk + pos;
      kslot[dst] = slot[r];
      kpts[dst] = rec[r];
    }
  }
}

// ------------------------------------------------------------------------- //
// k_tile                                                                      //
// ------------------------------------------------------------------------- //
// A tile's points, in INPUT order, are the concatenation over the chunks of the
// chunk's run for that tile.  The workgroup walks them through "windows" of kWin
// chunks: one wave takes four {offset,count} entries per lane, prefix sums across
// the wave, the sums staged in LDS; every position (one per thread and round) then
// finds its run by a binary search.
constexpr int kWin = 256;    // chunks per window
constexpr int kCapT = 2048;  // positions per tile whose {source, cell} stay cached in LDS for pass 2
constexpr int kP1 = 4;       // pass 1: rounds whose gathers are in flight together
constexpr int kP2 = 3;       // pass 2: rounds fetched ahead

struct TileLds {
  unsigned *cur;          // [tile slots] population, then bucket cursor
  unsigned *hist;         // [tile slots][WAVES bytes] per-round, per-wave point counts of a cell
  unsigned *pre, *srcb;   // [kWin] inclusive run ends / run source minus run start
  unsigned *csrc;         // [kCapT]
  unsigned short *cq;     // [kCapT]
};
__host__ __device__ inline size_t tile_lds_bytes(int tile_slots, int waves) {
  return ((size_t)tile_slots * (1 + waves / 4) + 2 * kWin + kCapT + kCapT / 2) * 4;
}

// Stages window `win` of the tile's row (ONE wave); returns the number of positions in it.
__device__ __forceinline__ int tile_stage_window(const int2 *__restrict__ row, int nch, int win,
                                                 const TileLds &L, int lane, unsigned *offsets) {
  const int c0 = win * kWin + 4 * lane;
  int2 e[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) e[i] = (c0 + i < nch) ? row[c0 + i] : make_int2(0, 0);
  // the runs' offsets in their chunks = points of all earlier tiles in those chunks
  *offsets = (unsigned)__builtin_amdgcn_readlane((int)wave_scan_u32((unsigned)(e[0].x + e[1].x + e[2].x + e[3].x)), kWave - 1);
  const int loc = e[0].y + e[1].y + e[2].y + e[3].y;
  const int incl = (int)wave_scan_u32((unsigned)loc);
  const int G = __builtin_amdgcn_readlane(incl, kWave - 1);
  int run = incl - loc;  // exclusive
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    L.srcb[4 * lane + i] = (unsigned)((c0 + i) * kChunk + e[i].x - run);
    run += e[i].y;
    L.pre[4 * lane + i] = (unsigned)run;
  }
  return G;
}

// source index (into the sweep's split arrays) of window position j < G
__device__ __forceinline__ int tile_find(const TileLds &L, int j) {
  int lo = 0;  // number of runs that end at or before position j (< kWin for j < G)
#pragma unroll
  for (int s = kWin / 2; s >= 1; s >>= 1)
    if (L.pre[lo + s - 1] <= (unsigned)j) lo += s;
  return (int)(L.srcb[lo] + (unsigned)j);
}

__device__ __forceinline__ unsigned byte_sum(unsigned v) { return __builtin_amdgcn_sad_u8(v, 0u, 0u); }

// One workgroup of WAVES waves per tile (1 << tile_shift consecutive slots, all of them
// in LDS).  Tiles need nothing from each other:
//
//   pass 1   population of the tile's cells (order-free LDS atomics); {source, cell} of
//            the first kCapT positions stay in LDS.  The tile's first place in the CSR array
//            comes for free: k_split stored, per chunk, the number of points of all EARLIER
//            tiles (the run's offset in the chunk), so the sum of the tile's offsets over the
//            chunks is the number of points before the tile
//   scan     prefix sums over the cells -> bucket starts, descriptors {slot, start, count} of
//            the occupied cells in a tile-local list, {points, occupied cells} of the tile
//            (k_emit prefix-sums the latter over the tiles: the pillar index of a cell is the
//            only thing that depends on other tiles, and only k_emit needs it)
//   pass 2   every point to bucket start + rank, rank in input order: rounds of THREADS
//            consecutive positions, wave w takes positions [64w, 64w+64) of the round; a
//            point's rank = cursor (earlier rounds) + points of its cell in earlier waves of
//            the round (byte histogram column) + earlier lanes of its wave (ballots)
// (the body of k_tile; k_step runs it as one of its roles with tile / b decoded from a flat block id)
template <typename T, int WAVES>
__device__ __forceinline__ void tile_body(
    const NPoints &np, const GridGeom &g, int ncap, int nchunks_cap, const int *__restrict__ kslot,
    const typename Rec4<T>::type *__restrict__ kpts, const int2 *__restrict__ mat,
    typename Rec4<T>::type *__restrict__ sorted_pts, int4 *__restrict__ tile_meta,
    u64 *__restrict__ tile_agg, u64 *stamps, unsigned *tile_smem, int tile, int b) {
  using Rec = typename Rec4<T>::type;
  constexpr int THREADS = WAVES * kWave;
  constexpr int HW = WAVES / 4;  // dwords of a cell's per-wave byte counts
  __shared__ int s_G;
  __shared__ unsigned s_before;  // points of all earlier tiles
  __shared__ u64 s_wave[WAVES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  [[maybe_unused]] const int stamp_nx = g.ntiles, stamp_bx = tile, stamp_by = b;
  PP_STAMP(0);
  const int TS = 1 << g.tile_shift;
  TileLds L;
  L.cur = tile_smem;
  L.hist = L.cur + TS;
  L.pre = L.hist + HW * TS;
  L.srcb = L.pre + kWin;
  L.csrc = L.srcb + kWin;
  L.cq = reinterpret_cast<unsigned short *>(L.csrc + kCapT);
  for (int q = tid; q < (1 + HW) * TS; q += THREADS) L.cur[q] = 0u;  // cur and hist
  if (tid == 0) s_before = 0u;
  PP_STAMP(1);
  const int n = np.n[b];
  const int nch = (n + kChunk - 1) / kChunk;
  const int nwin = (nch + kWin - 1) / kWin;
  const int2 *row = mat + ((int64_t)b * g.ntiles + tile) * nchunks_cap;
  const int *ks = kslot + (int64_t)b * ncap;
  const Rec *kp = kpts + (int64_t)b * ncap;
  const unsigned qmask = (unsigned)TS - 1u;
  // pass 1
  // A sweep of at most kWin chunks (262 144 points) is ONE window, and the first kP2 rounds of pass 2 then
  // read the same sources pass 1 has just found: their records are fetched HERE, beside the slot loads, and
  // ride through the scan in registers -- one dependent memory round trip less per tile (under k_step's
  // store load every round trip is microseconds).
  static_assert(kP2 <= kP1, "pass 2's first rounds are a prefix of pass 1's first trip");
  const bool one_win = nwin == 1;
  static_assert(kP2 == 3, "rec_e0..2 below (named values handed to run_rounds: captured by reference they stay in scratch)");
  Rec rec_e0, rec_e1, rec_e2;
  rec_e0.x = rec_e0.y = rec_e0.z = rec_e0.w = 0;
  rec_e1 = rec_e0;
  rec_e2 = rec_e0;
  int ntile = 0;  // positions walked so far = points of the tile
  for (int win = 0; win < nwin; ++win) {
    __syncthreads();  // window arrays free (and cur/hist zeroed)
    if (w == 0) {
      unsigned offs = 0;
      const int G0 = tile_stage_window(row, nch, win, L, lane, &offs);
      if (lane == 0) {
        s_G = G0;
        s_before += offs;
      }
    }
    __syncthreads();
    const int G = s_G;
    for (int J = 0; J < G; J += kP1 * THREADS) {
      int src[kP1], sl[kP1];
#pragma unroll
      for (int u = 0; u < kP1; ++u) {
        const int j = J + u * THREADS + tid;
        src[u] = (j < G) ? tile_find(L, j) : -1;
      }
#pragma unroll
      for (int u = 0; u < kP1; ++u) sl[u] = (src[u] >= 0) ? ks[src[u]] : 0;
      if (one_win && J == 0) {
        if (src[0] >= 0) rec_e0 = kp[src[0]];
        if (src[1] >= 0) rec_e1 = kp[src[1]];
        if (src[2] >= 0) rec_e2 = kp[src[2]];
      }
#pragma unroll
      for (int u = 0; u < kP1; ++u) {
        if (src[u] >= 0) {
          const unsigned q = (unsigned)sl[u] & qmask;
          __hip_atomic_fetch_add(&L.cur[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const int jg = ntile + J + u * THREADS + tid;
          if (jg < kCapT) {
            L.csrc[jg] = (unsigned)src[u];
            L.cq[jg] = (unsigned short)q;
          }
        }
      }
    }
    ntile += G;
  }
  __syncthreads();
  PP_STAMP(2);
  // prefix sums over the tile's cells: thread owns cpt consecutive cells
  const int cpt = TS > THREADS ? TS / THREADS : 1;  // 1 .. 16
  u64 mine = 0;
  if (tid * cpt < TS)
    for (int e = 0; e < cpt; ++e) {
      const unsigned c = L.cur[tid * cpt + e];
      mine += ((u64)c << 32) | (u64)(c > 0);
    }
  const u64 inc = wave_scan_2x32(mine);  // {points, occupied cells}: two independent counts
  if (lane == kWave - 1) s_wave[w] = inc;
  PP_STAMP(3);
  __syncthreads();
  PP_STAMP(4);
  u64 wave_off = 0, agg = 0;
#pragma unroll
  for (int k = 0; k < WAVES; ++k) {
    if (k < w) wave_off += s_wave[k];
    agg += s_wave[k];
  }
  if (tid == 0) tile_agg[(int64_t)b * g.ntiles + tile] = agg;  // {points << 32 | occupied cells}
  // descriptors of the tile's occupied cells (tile-local list); the counters become cursors
  const unsigned sbase = s_before;
  if (tid * cpt < TS) {
    const u64 base = wave_off + (inc - mine);
    int k = (int)(base & 0xFFFFFFFFull);
    unsigned st = sbase + (unsigned)(base >> 32);
    int4 *tm = tile_meta + ((int64_t)b * g.ntiles + tile) * TS;
    for (int e = 0; e < cpt; ++e) {
      const int q = tid * cpt + e;
      const unsigned c = L.cur[q];
      if (c > 0) tm[k++] = make_int4(tile * TS + q, (int)st, (int)c, 0);
      L.cur[q] = st;
      st += c;
    }
  }
  PP_STAMP(5);
  // pass 2
  Rec *sp = sorted_pts + (int64_t)b * ncap;
  const u64 below = lanes_below(lane);
  unsigned char *histb = reinterpret_cast<unsigned char *>(L.hist);
  unsigned wmask[HW];  // bytes of a column that belong to waves < w
#pragma unroll
  for (int i = 0; i < HW; ++i) {
    const int nb = min(max(w - 4 * i, 0), 4);
    wmask[i] = nb == 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
  }
  auto place = [&](bool v, unsigned q, const Rec &rec) {
    const u64 peers = wave_peers(q, g.tile_shift, v);
    const int rank = __popcll(peers & below);
    const bool leader = v && rank == 0;
    if (leader) histb[q * WAVES + w] = (unsigned char)__popcll(peers);  // <= 64
    __syncthreads();
    unsigned before = 0, total = 0, old = 0;
    if (v) {
#pragma unroll
      for (int i = 0; i < HW; ++i) {
        const unsigned col = L.hist[q * HW + i];
        before += byte_sum(col & wmask[i]);
        total += byte_sum(col);
      }
      old = L.cur[q];
    }
    __syncthreads();
    if (leader) {
      histb[q * WAVES + w] = 0;                 // self-cleaning
      if (before == 0) L.cur[q] = old + total;  // first wave of the cell this round
    }
    if (v) sp[old + before + (unsigned)rank] = rec;
  };
  // rounds over a list of `count` positions; fetch(j, q, rec) reads position j.
  // kP2 rounds are fetched ahead of the one being placed.
  auto run_rounds = [&](int count, bool early, const Rec e0, const Rec e1, const Rec e2, auto &&fetch) {
    bool v[kP2];
    unsigned q[kP2];
    Rec rec[kP2];
#pragma unroll
    for (int d = 0; d < kP2; ++d) {
      const int j = d * THREADS + tid;
      v[d] = j < count;
      q[d] = 0;
      rec[d].x = rec[d].y = rec[d].z = rec[d].w = 0;
      if (early) {  // (uniform) position j's record came in during pass 1
        if (v[d]) q[d] = L.cq[j];
        rec[d] = d == 0 ? e0 : d == 1 ? e1 : e2;
      } else if (v[d]) {
        fetch(j, q[d], rec[d]);
      }
    }
    for (int J = 0; J < count; J += kP2 * THREADS) {
#pragma unroll
      for (int d = 0; d < kP2; ++d) {
        if (J + d * THREADS >= count) break;  // uniform
        place(v[d], q[d], rec[d]);
        const int j = J + (kP2 + d) * THREADS + tid;
        v[d] = j < count;
        if (v[d]) fetch(j, q[d], rec[d]);
      }
    }
  };
  if (ntile <= kCapT) {
    __syncthreads();  // cursors written
    run_rounds(ntile, one_win, rec_e0, rec_e1, rec_e2, [&](int j, unsigned &q, Rec &rec) {
      q = L.cq[j];
      rec = kp[L.csrc[j]];
    });
  } else {  // a crowded tile: walk it again
    for (int win = 0; win < nwin; ++win) {
      __syncthreads();
      if (w == 0) {
        unsigned offs = 0;
        const int G0 = tile_stage_window(row, nch, win, L, lane, &offs);
        if (lane == 0) s_G = G0;
      }
      __syncthreads();
      run_rounds(s_G, false, rec_e0, rec_e1, rec_e2, [&](int j, unsigned &q, Rec &rec) {
        const int src = tile_find(L, j);
        q = (unsigned)ks[src] & qmask;
        rec = kp[src];
      });
    }
  }
  PP_STAMP(7);
}

// The fused feature-net mode's canvas (PPScatter, model/model.py:53-62) must read zero wherever no pillar
// lands.  A caller that hands the SAME canvas back call after call (PillarPipeline.forward_fused) does not
// need the whole 64 MB per sweep cleared again: only the pixels the previous call wrote are non-zero, and
// the previous call's indices say which.  Extra workgroups of the k_tile launch zero exactly those (a kernel
// boundary before k_emit writes the new ones): 3 MB per sweep instead of 64 MB.
struct UnscatterArgs {
  const long long *prev_idx;  // [B][P][3] {flag, col, row} of the call that filled the canvas, or NULL
  float *canvas;
  int P, h, w, nhwc, nblocks;
};
constexpr int kPfnChannels = 64;  // one lane per output channel (model/model.py:28: 9 -> 64)
constexpr int kUnscatterPixels = 256;  // pixels per extra workgroup

__device__ __forceinline__ void unscatter_body(const UnscatterArgs &u, int blk, int b, int threads) {
  const int tid = threadIdx.x;
  const int64_t plane = (int64_t)u.h * u.w;
  const int p_end = min(u.P, (blk + 1) * kUnscatterPixels);
  if (u.nhwc) {
    // a pixel = 64 channels = 256 contiguous bytes: 16 lanes x 16 bytes
    const int sub = tid & 15;
    for (int p = blk * kUnscatterPixels + (tid >> 4); p < p_end; p += threads >> 4) {
      const long long *e = u.prev_idx + ((int64_t)b * u.P + p) * 3;
      if (e[0] == 0) continue;
      const long long col = e[1], row = e[2];
      if (row < 0 || row >= u.h || col < 0 || col >= u.w) continue;
      reinterpret_cast<float4 *>(u.canvas + (((int64_t)b * u.h + row) * u.w + col) * kPfnChannels)[sub] =
          make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
    const int c = tid & 63;
    for (int p = blk * kUnscatterPixels + (tid >> 6); p < p_end; p += threads >> 6) {
      const long long *e = u.prev_idx + ((int64_t)b * u.P + p) * 3;
      if (e[0] == 0) continue;
      const long long col = e[1], row = e[2];
      if (row < 0 || row >= u.h || col < 0 || col >= u.w) continue;
      u.canvas[((int64_t)b * kPfnChannels + c) * plane + row * u.w + col] = 0.0f;
    }
  }
}

// Which tile a tile workgroup takes.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one
// and its L2 -- observed, not promised: MI355X_MICROARCH.md); neighbouring tiles gather neighbouring runs of every
// chunk of the split arrays (a 128-byte line of kpts holds the runs of ~4 tiles at BASELINE config 5's 800k points),
// so with tile = block every line was fetched by several XCDs -- k_tile's FETCH_SIZE was 1.4-1.8x what it gathers.
// Blocks t, t + 8, t + 16, ... take CONSECUTIVE tiles instead: an eighth of the tile range per XCD.  (A permutation of
// the tiles whatever the dispatch order is: placement changes the traffic, never the result.)
__device__ __forceinline__ int tile_of_block(int t, int nt) {
#ifdef PP_TILE_NO_XCD_MAP  // tools/lab: A/B
  return t;
#else
  const int k = t & 7, q = t >> 3;
  return k * (nt >> 3) + min(k, nt & 7) + q;
#endif
}

template <typename T, int WAVES>
__global__ __launch_bounds__(WAVES * kWave) void k_tile(
    NPoints np, GridGeom g, int ncap, int nchunks_cap, const int *__restrict__ kslot,
    const typename Rec4<T>::type *__restrict__ kpts, const int2 *__restrict__ mat,
    typename Rec4<T>::type *__restrict__ sorted_pts, int4 *__restrict__ tile_meta,
    u64 *__restrict__ tile_agg, u64 *stamps, UnscatterArgs un) {
  extern __shared__ __attribute__((aligned(16))) unsigned tile_smem[];
  if ((int)blockIdx.x >= g.ntiles) {  // (only launched when un.nblocks > 0)
    unscatter_body(un, (int)blockIdx.x - g.ntiles, (int)blockIdx.y, WAVES * kWave);
    return;
  }
  tile_body<T, WAVES>(np, g, ncap, nchunks_cap, kslot, kpts, mat, sorted_pts, tile_meta, tile_agg, stamps,
                      tile_smem, tile_of_block((int)blockIdx.x, g.ntiles), (int)blockIdx.y);
}

// ------------------------------------------------------------------------- //
// k_emit                                                                      //
// ------------------------------------------------------------------------- //
// CAP: the wave's pooled bucket capacity (points, 4-padded per pillar).  CAPW = 128 for the dense modes; the
// fused feature-net mode takes 64: its waves write almost nothing, so its rate IS its occupancy, and the
// smaller pool (14 KB per workgroup) with a tighter register bound gives it six waves per SIMD instead of
// four (k_emit<pfn> 26.9 -> 23.3 us at C2 B=4; the dense modes lose with 64 on crowded waves -- C5 B=1
// 20.7 -> 25.4 us, row-major 28.0 -> 30.7 us -- and keep 128).
template <typename TIn, int CAP = CAPW>
struct alignas(32) WaveLds {
  static constexpr int kCap = CAP;
  TIn px[CAP], py[CAP], pz[CAP], pr[CAP];  // points, input order per pillar, 4-aligned buckets
  union {
    double4 cq[CAP];  // chain operands of one point: {n/(n+1), x/(n+1), y/(n+1), z/(n+1)}
    float feat[PP_NUM_FEATURES][CAP];  // f32 features (dense / fused-net modes), aliases cq
  } u;
  double mean[KW][3];
  double cx[KW], cy[KW];  // canvas_x / canvas_y (pillars.cpp:278-280), once per pillar
  int cnt[KW], live[KW], slot[KW], start[KW];
  int pix[KW];            // fused feature-net mode: the pillar's canvas pixel row * W + col, -1: none / off the canvas
};

struct EmitArgs {
  GridGeom g;
  NPoints np;
  int P, N, ncap;
  const int4 *tile_meta;    // [B][tiles][tile slots] {slot, start, count, -} of a tile's occupied cells
  const u64 *tile_agg;      // [B][tiles] {points << 32 | occupied cells}
  int4 *pillar_meta;        // [B][P] the same descriptors in pillar order (written in compact mode: host path)
  const int4 *ordered_meta;     // k_step: [B][P] descriptors in pillar order, written by the ORDER role of the previous
  const int2 *ordered_totals;   // launch, and [B] {cells, points}; NULL -> every wave takes the tile prefix itself
  int2 *totals;             // [B] {cells, points}, written here
  const void *sorted_pts;   // [B][ncap] point record (x,y,z,r), CSR order, input order per bucket
  // dense mode
  float *out;          // [B][9][P][N]
  long long *idx_out;  // [B][P][3]
  // compact mode
  int4 *pillar_meta_host;   // compact mode, optional: the same descriptors and totals ALSO into device-visible host memory
  int2 *totals_host;        // (pp_create_pillars_f64: they are on the host when the kernel is done, no copy engine)
  double *feat_out;  // [B][ncap][9]; or, feat_pack = 1 (f32-valued clouds), [B][ncap] records of kPackedFeat bytes
  int feat_pack;
  // fused feature-net mode
  const float *pfn_w;  // [64][12]: w[0..8], bias, bn scale, bn shift per output channel
  float *pfn_out;      // [B][64][P], or NULL when only the canvas is wanted
  // ... scattered straight into the BEV canvas (PPScatter, model/model.py:53-62)
  float *canvas;       // NULL, [B][H][W][64] (channels last) or [B][64][H][W]
  int canvas_h, canvas_w, canvas_nhwc;
  u64 *stamps;         // tools/lab builds only
};

enum { kModeDenseVec4 = 0, kModeDenseScalar = 1, kModeCompact = 2, kModePfn = 3 };
constexpr int kAuxPlain = 0, kAuxSc1 = 16;           // buffer-store cache policy bits (gfx950: sc1 = 16)
constexpr size_t kSc1MaxBytes = 128u << 20;          // write-through stores pay off up to about half the Infinity Cache
constexpr size_t kTileXcdMapBytes = 8u << 20;        // k_step: split records (16 + 4 B per point) of the tile role's batch
                                                     // beyond which it takes its tiles XCD by XCD (two L2s' worth)
constexpr size_t kStepLaunchBytes = 0;               // dense output per k_step launch; 0 = one launch per call (see
                                                     // pp_voxelize_step_dev: splitting was measured and gains nothing)


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

// The running means (pillars.cpp:311-328) of the pillars in `mask` -- every occupied pillar of a wave whose buckets do
// not fit the LDS pool together -- streamed through the pool in slices, in input order (k_tile left them that way), all
// of them SIDE BY SIDE: the chain of pillar k and coordinate c runs in lane 3 k + c, like the pooled pillars' chains
// in emit_group.  A chain step is a dependent f64 multiply and add whose operands come from LDS; the wave pays per
// instruction, not per lane, so the pillars of a wave cost what the LONGEST of them costs (round 5 took them run by
// run: in row-major order at BASELINE config 1's 100 x 100 grid a wave's four neighbours sum to 1127 points, the
// longest single one is 381).  What the emit-wave stamps of config 1's shapes said about the first form of this pass
// (fixed 32-point slices, operands fetched eight steps at a time from C++): 12-14 us for the wave that holds the
// 381-point cell, of a launch of 23 -- 75 cycles per step, of which the two dependent f64 operations are ~20.  Hence:
//   * every lane's chain runs the SAME number of steps per round: a slot behind a pillar's last point holds the
//     identity operands (scale 1.0, value -0.0: m * 1.0 + -0.0 == m for every m, signed zeros and NaNs included), and
//     the first point is an ordinary step too (scale 0 / 1 = 0.0, value x / 1 = x, from m = -0.0: -0.0 * 0.0 + x == x
//     bit for bit, for x = +-0.0 as well) -- no per-lane trip counts, no special first step;
//   * the uniform loop is inline assembly: four steps per group, two register sets in turn, the NEXT group's operands
//     asked for before THIS group's four dependent steps -- the LDS round trip runs under the arithmetic.  (From C++
//     the compiler cannot be made to do this: a fetch inside a branch is waited for at the join; fetched
//     unconditionally the reads are moved down to their uses; with reads and waits as separate asm statements it
//     copies a destination register before the wait that makes it valid.  Sixteen steps fetched together from C++ cost
//     the dense kernels 30 VGPRs and k_step its first spills.)
//   * the slice adapts: the pool is shared by the pillars that still have points, 128 / 64 / 32 slots each for 1 / 2 /
//     3-4 of them (CAP = 128) -- fewer rounds, hence fewer rounds of the operands' four f64 divisions per point,
//     once the short buckets are through.
// Results: L.mean[k].
#ifdef PP_STAMPS  // tools/lab builds: where streamed_means' time goes (shader clocks, maxima over the launch's waves)
__device__ unsigned long long g_means_prof[8];  // 0 total, 1 chains (asm loops), 2 staging (operands), 3 steps, 4 rounds
#define PP_MEANS_CLK() __builtin_readcyclecounter()
#else
#define PP_MEANS_CLK() 0ull
#endif
template <typename TIn, int CAP>
__device__ void streamed_means(WaveLds<TIn, CAP> &L, const EmitArgs &a, int b, unsigned mask, int lane) {
  using Rec = typename Rec4<TIn>::type;
  constexpr int kPre = CAP / 64;
  int cnt[KW];
  const Rec *sp[KW];
  int maxcnt = 0;
#pragma unroll
  for (int k = 0; k < KW; ++k) {
    cnt[k] = ((mask >> k) & 1u) ? L.cnt[k] : 0;  // (wave-uniform)
    sp[k] = reinterpret_cast<const Rec *>(a.sorted_pts) + (int64_t)b * a.ncap + L.start[k];
    maxcnt = max(maxcnt, cnt[k]);
  }
  // a round's plan, wave-uniform: the pillars with points left share the pool
  struct Plan {
    unsigned act;  // pillars with points at or behind `base`
    int shift;     // log2 of the slice
  };
  auto plan_at = [&](int base) {
    Plan p;
    p.act = 0;
#pragma unroll
    for (int k = 0; k < KW; ++k)
      if (cnt[k] > base) p.act |= 1u << k;
    const int nb = __popc(p.act);
    int cap_log = 0;
    while ((2 << cap_log) <= CAP) ++cap_log;  // log2(CAP)
    p.shift = cap_log - (nb <= 1 ? 0 : nb == 2 ? 1 : 2);
    return p;
  };
  // entry j of a round = slot `off` of the slice of the q-th active pillar; its operands live at cq[j]
  auto fetch = [&](const Plan &p, int base, Rec rec[kPre], int ek[kPre], int ei[kPre]) {
#pragma unroll
    for (int it = 0; it < kPre; ++it) {
      const int j = lane + it * kWave;
      const int q = j >> p.shift, off = j & ((1 << p.shift) - 1);
      unsigned rest = p.act;
      for (int u = 0; u < q; ++u) rest &= rest - 1;  // drop the q lowest set bits
      ek[it] = rest ? __ffs((int)rest) - 1 : -1;
      ei[it] = base + off;
      rec[it].x = rec[it].y = rec[it].z = rec[it].w = 0;
      if (ek[it] >= 0) {
        int c = 0;
        const Rec *s = sp[0];
#pragma unroll
        for (int k = 0; k < KW; ++k)
          if (k == ek[it]) c = cnt[k], s = sp[k];
        if (ei[it] < c) rec[it] = s[ei[it]];
        else ek[it] = -2 - ek[it];  // a slot behind the pillar's last point: identity operands
      }
    }
  };
  const int kq = lane / 3, cq_ = lane - 3 * kq;
  const bool chain = lane < 3 * KW && ((mask >> kq) & 1u);
  [[maybe_unused]] unsigned long long pf_t0 = PP_MEANS_CLK(), pf_chain = 0, pf_stage = 0, pf_steps = 0, pf_rounds = 0, pf_fetch = 0;
  double m = -0.0;
  Rec rec[kPre];
  int ek[kPre], ei[kPre];
  Plan cur = plan_at(0);
  fetch(cur, 0, rec, ek, ei);
  for (int base = 0; base < maxcnt;) {
    const int slice = 1 << cur.shift;
    int left = 0;  // steps of this round: the longest remainder, at most a slice, in whole groups of four
#pragma unroll
    for (int k = 0; k < KW; ++k) left = max(left, cnt[k] - base);
    const int steps = (min(left, slice) + 3) & ~3;
    [[maybe_unused]] const unsigned long long pf_a = PP_MEANS_CLK();
#pragma unroll
    for (int it = 0; it < kPre; ++it) {
      const int j = lane + it * kWave;
      if (ek[it] >= 0) {
        const double n = (double)ei[it], den = n + 1;
        // pillars.cpp:322-326: n/(n+1) and v/(n+1), true f64 divisions, lane-parallel
        L.u.cq[j] = make_double4(n / den, (double)rec[it].x / den, (double)rec[it].y / den, (double)rec[it].z / den);
      } else if (ek[it] < -1 && (j & (slice - 1)) < steps) {
        L.u.cq[j] = make_double4(1.0, -0.0, -0.0, -0.0);
      }
    }
    wave_sync();
    [[maybe_unused]] const unsigned long long pf_c = PP_MEANS_CLK();
    // the next round's records travel while this round's chains run
    const int base_next = base + slice;
    const Plan nxt = plan_at(base_next);
    if (base_next < maxcnt) fetch(nxt, base_next, rec, ek, ei);
    [[maybe_unused]] const unsigned long long pf_b = PP_MEANS_CLK();
    if (chain && ((cur.act >> kq) & 1u)) {
      const int qpos = __popc(cur.act & ((1u << kq) - 1u)) << cur.shift;
      // (the generic pointer's low 32 bits ARE the LDS byte address: aperture base in the high half + offset)
      unsigned sa = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) double4 *)&L.u.cq[qpos];
      unsigned va = sa + 8u * (unsigned)(1 + cq_);
      int ng = __builtin_amdgcn_readfirstlane(steps >> 2);  // >= 1
      // Register sets A = v[32:47], B = v[48:63] (named in the clobber list: the reads are ds_read2_b64 -- two steps'
      // scales, or two steps' values, per instruction -- whose four-register destinations are used half by half, which
      // an asm operand cannot express).  Per group of four steps: four LDS instructions, eight dependent f64 operations,
      // the next group's operands in flight.  Measured (tools/lab/emit_stamps.py, profiles/r06/NOTES.md): 31 clocks per
      // step; the bare dependent pair is 15-17 (tools/lab/f64_rate.cpp).  A wave that is alone on its SIMD issues one
      // instruction every four clocks, whatever its kind, so every instruction of the loop counts: one ds_read_b64 per
      // operand (eight reads per group) gave 33.5, a third register set (operands two groups ahead) 35 and the first
      // spills -- the LDS round trip is covered, what is left is the loop's own instruction count.
      asm volatile(
          "s_waitcnt lgkmcnt(0)\n\t"
          "ds_read2_b64 v[32:35], %1 offset1:4\n\t"
          "ds_read2_b64 v[40:43], %2 offset1:4\n\t"
          "ds_read2_b64 v[36:39], %1 offset0:8 offset1:12\n\t"
          "ds_read2_b64 v[44:47], %2 offset0:8 offset1:12\n"
          "1:\n\t"
          "ds_read2_b64 v[48:51], %1 offset0:16 offset1:20\n\t"
          "ds_read2_b64 v[56:59], %2 offset0:16 offset1:20\n\t"
          "ds_read2_b64 v[52:55], %1 offset0:24 offset1:28\n\t"
          "ds_read2_b64 v[60:63], %2 offset0:24 offset1:28\n\t"
          "s_waitcnt lgkmcnt(4)\n\t"
          "v_mul_f64 %0, %0, v[32:33]\n\t"
          "v_add_f64 %0, %0, v[40:41]\n\t"
          "v_mul_f64 %0, %0, v[34:35]\n\t"
          "v_add_f64 %0, %0, v[42:43]\n\t"
          "v_mul_f64 %0, %0, v[36:37]\n\t"
          "v_add_f64 %0, %0, v[44:45]\n\t"
          "v_mul_f64 %0, %0, v[38:39]\n\t"
          "v_add_f64 %0, %0, v[46:47]\n\t"
          "s_add_i32 %3, %3, -1\n\t"
          "s_cmp_eq_u32 %3, 0\n\t"
          "s_cbranch_scc1 2f\n\t"
          "ds_read2_b64 v[32:35], %1 offset0:32 offset1:36\n\t"
          "ds_read2_b64 v[40:43], %2 offset0:32 offset1:36\n\t"
          "ds_read2_b64 v[36:39], %1 offset0:40 offset1:44\n\t"
          "ds_read2_b64 v[44:47], %2 offset0:40 offset1:44\n\t"
          "v_add_u32 %1, 0x100, %1\n\t"
          "v_add_u32 %2, 0x100, %2\n\t"
          "s_waitcnt lgkmcnt(4)\n\t"
          "v_mul_f64 %0, %0, v[48:49]\n\t"
          "v_add_f64 %0, %0, v[56:57]\n\t"
          "v_mul_f64 %0, %0, v[50:51]\n\t"
          "v_add_f64 %0, %0, v[58:59]\n\t"
          "v_mul_f64 %0, %0, v[52:53]\n\t"
          "v_add_f64 %0, %0, v[60:61]\n\t"
          "v_mul_f64 %0, %0, v[54:55]\n\t"
          "v_add_f64 %0, %0, v[62:63]\n\t"
          "s_add_i32 %3, %3, -1\n\t"
          "s_cmp_lg_u32 %3, 0\n\t"
          "s_cbranch_scc1 1b\n"
          "2:\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "+v"(m), "+v"(sa), "+v"(va), "+s"(ng)
          :
          : "scc", "memory", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44",
            "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60",
            "v61", "v62", "v63");
    }
    wave_sync();
#ifdef PP_STAMPS
    pf_stage += pf_c - pf_a;
    pf_fetch += pf_b - pf_c;
    pf_chain += PP_MEANS_CLK() - pf_b;
    pf_steps += (unsigned long long)steps;
    pf_rounds += 1;
#endif
    base = base_next;
    cur = nxt;
  }
  if (chain) L.mean[kq][cq_] = m;
  wave_sync();
#ifdef PP_STAMPS
  if (lane == 0) {
    const unsigned long long tot = PP_MEANS_CLK() - pf_t0;
    if (tot > atomicMax(&g_means_prof[0], tot)) {  // (the slowest wave's breakdown, more or less)
      g_means_prof[1] = pf_chain;
      g_means_prof[2] = pf_stage;
      g_means_prof[3] = pf_steps;
      g_means_prof[4] = pf_rounds;
      g_means_prof[5] = pf_fetch;
    }
  }
#endif
}

// The features of a big pillar's first min(count, N) points (its mean is in L.mean[k]: streamed_means).
// Compact mode's store of one point's nine features (the host drop-in's transport, pp_create_pillars_f64).  When the
// cloud is f32-valued the first four -- x, y, z, intensity, which ARE the input's floats -- travel as floats: 56 bytes a
// point instead of 72 over PCIe, widened again by the host's scatter; the five differences keep their doubles.
constexpr int kPackedFeat = 16 + 5 * 8;
__device__ __forceinline__ void store_compact_features(const EmitArgs &a, int64_t idx, const double (&f)[9]) {
  if (a.feat_pack) {
    char *o = reinterpret_cast<char *>(a.feat_out) + idx * kPackedFeat;  // 8-byte aligned
    *reinterpret_cast<float2 *>(o) = make_float2((float)f[0], (float)f[1]);
    *reinterpret_cast<float2 *>(o + 8) = make_float2((float)f[2], (float)f[3]);
#pragma unroll
    for (int d = 4; d < 9; ++d) *reinterpret_cast<double *>(o + 16 + (d - 4) * 8) = f[d];
  } else {
    double *o = a.feat_out + idx * 9;
#pragma unroll
    for (int d = 0; d < 9; ++d) o[d] = f[d];
  }
}

template <typename TIn, int MODE, int CAP>
__device__ void emit_big_pillar(WaveLds<TIn, CAP> &L, const EmitArgs &a, int b, int k, int p,
                                int lane, PfnAcc *acc = nullptr, float *rmax = nullptr,
                                float *rmin = nullptr) {
  using Rec = typename Rec4<TIn>::type;
  const Rec *sp = reinterpret_cast<const Rec *>(a.sorted_pts) + (int64_t)b * a.ncap + L.start[k];
  const double m0 = L.mean[k][0], m1 = L.mean[k][1], m2 = L.mean[k][2];
  const double mean[3] = {m0, m1, m2};
  const double cx = L.cx[k], cy = L.cy[k];
  const int N = a.N;
  const int live = L.live[k];
  for (int base = 0; base < live; base += kWave) {
    const int n = base + lane;
    if (n < live) {
      const Rec rec = sp[n];
      double f[9];
      point_features((double)rec.x, (double)rec.y, (double)rec.z, (double)rec.w, cx, cy, mean, f);
      if (MODE == kModeCompact) {
        store_compact_features(a, (int64_t)b * a.ncap + L.start[k] + n, f);
      } else if (MODE == kModePfn) {
#pragma unroll
        for (int d = 0; d < 9; ++d) L.u.feat[d][lane] = (float)f[d];
      } else {
        float *o = a.out + (int64_t)b * 9 * a.P * N;
#pragma unroll
        for (int d = 0; d < 9; ++d) o[((int64_t)d * a.P + p) * N + n] = (float)f[d];
      }
    }
    if (MODE == kModePfn) {
      // every lane is a channel: fold this chunk's live points into its max/min
      wave_sync();
      const int c = min(kWave, live - base);
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

// One run of the streamed path: the LIVE points (the first min(count, N), input order) of the run's pillars, pillar k at
// LDS positions [spg[k], spg[k] + lvg[k]) (4-aligned starts, gpad positions in all).  A lane owns a position: it loads
// that point from the CSR array and turns it into its nine features (the means are in L.mean: streamed_means) -- into
// L.u.feat for the slab store / the fold (dense vec4 and fused-net modes), or straight to memory (the other modes).
template <typename TIn, int MODE, int CAP>
__device__ __forceinline__ void emit_live_run(WaveLds<TIn, CAP> &L, const EmitArgs &a, int b, int p0, int lane,
                                              const int spg[KW], const int lvg[KW], int gpad) {
  using Rec = typename Rec4<TIn>::type;
  constexpr int kPre = CAP / 64;
  const int N = a.N;
  const Rec *base = reinterpret_cast<const Rec *>(a.sorted_pts) + (int64_t)b * a.ncap;
  Rec rec[kPre];
  int kq[kPre], rq[kPre];
#pragma unroll
  for (int it = 0; it < kPre; ++it) {
    const int q = lane + it * kWave;
    kq[it] = -1;
    rq[it] = 0;
    rec[it].x = rec[it].y = rec[it].z = rec[it].w = 0;
    if (q < gpad) {
#pragma unroll
      for (int kk = 0; kk < KW; ++kk)
        if (q >= spg[kk] && q < spg[kk] + lvg[kk]) {
          kq[it] = kk;
          rq[it] = q - spg[kk];
        }
    }
    if (kq[it] >= 0) rec[it] = base[L.start[kq[it]] + rq[it]];
  }
  float *outb = a.out + (int64_t)b * 9 * a.P * N;
#pragma unroll
  for (int it = 0; it < kPre; ++it) {
    const int k = kq[it], n = rq[it], q = lane + it * kWave;
    if (k < 0) continue;
    double f[9];
    point_features((double)rec[it].x, (double)rec[it].y, (double)rec[it].z, (double)rec[it].w, L.cx[k], L.cy[k],
                   L.mean[k], f);
    if (MODE == kModeCompact) {
      store_compact_features(a, (int64_t)b * a.ncap + L.start[k] + n, f);
    } else if (MODE == kModeDenseScalar) {
#pragma unroll
      for (int d = 0; d < 9; ++d) outb[((int64_t)d * a.P + (p0 + k)) * N + n] = (float)f[d];
    } else {
#pragma unroll
      for (int d = 0; d < 9; ++d) L.u.feat[d][q] = (float)f[d];
    }
  }
  wave_sync();
}

// Pooled pillars [kbeg,kend) of this wave: entries j (pooled bucket position)
// were prefetched by the caller into registers (rec_r[it] for j = lane + 64*it)
// straight from the CSR array, where every bucket already is in input order.
// Leaves the f32 features of the live points in L.u.feat (dense vec4 mode) or
// stores them (other modes).
template <typename TIn, int MODE, int CAP>
__device__ __forceinline__ void emit_group(WaveLds<TIn, CAP> &L, const EmitArgs &a, int b, int p0,
                                           int kbeg, int kend, int lane,
                                           const typename Rec4<TIn>::type rec_r[CAP / 64],
                                           int segbeg[KW],
                                           int segpad[KW], int cntk[KW], int T) {
  constexpr int kPre = CAP / 64;  // bucket entries prefetched per lane
  const int N = a.N;
  // LDS arrays use 4-aligned bucket starts (16-byte reads in the store pass)
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
      const typename Rec4<TIn>::type rec = rec_r[it];
      const int r = j - sb;  // position in the bucket = input order
      const int pos = sp + r;
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
  // sequential running mean, pillars.cpp:311-328: m <- m * (n/(n+1)) + v/(n+1), one chain per pillar and
  // coordinate.  A chain is serial, and an f64 operation occupies the pipe for the whole wave whatever the
  // number of active lanes -- so the 3*KW chains run SIDE BY SIDE, one lane each (two f64 operations per
  // step for all of them; one lane per pillar with its x, y, z chains in sequence cost six).
  if (lane < 3 * KW) {
    const int kq = lane / 3, cq_ = lane - 3 * kq;
    int sb = 0, sc = 0;
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      sb = (kq == kk) ? segpad[kk] : sb;
      sc = (kq == kk) ? cntk[kk] : sc;
    }
    if (sc > 0) {
      double m = cq_ == 0 ? (double)L.px[sb] : cq_ == 1 ? (double)L.py[sb] : (double)L.pz[sb];
      // operands of point n: {scale, x', y', z'} = four doubles; the chain is serial in m, its operands are
      // not: four points' operands are fetched ahead of the four dependent mul+add steps
      const double *op = reinterpret_cast<const double *>(&L.u.cq[sb]);
      int n = 1;
      for (; n + 3 < sc; n += 4) {
        const double s0 = op[4 * n], v0 = op[4 * n + 1 + cq_], s1 = op[4 * n + 4], v1 = op[4 * n + 5 + cq_];
        const double s2 = op[4 * n + 8], v2 = op[4 * n + 9 + cq_], s3 = op[4 * n + 12], v3 = op[4 * n + 13 + cq_];
        m = m * s0 + v0;
        m = m * s1 + v1;
        m = m * s2 + v2;
        m = m * s3 + v3;
      }
      for (; n < sc; ++n) m = m * op[4 * n] + op[4 * n + 1 + cq_];
      L.mean[kq][cq_] = m;
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
        store_compact_features(a, (int64_t)b * a.ncap + start0 + j, f);
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

template <int PASS, typename TIn, int AUX>
__device__ __forceinline__ void store_slab(const WaveLds<TIn, CAPW> &L, const SlabGeom &sg,
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
        __builtin_amdgcn_raw_buffer_store_b128(v[d], rs, voff, d * sg.pn4_bytes, AUX);
    }
  }
}

// AUX: cache policy of the dense tensor's 16-byte stores (kAuxPlain / kAuxSc1, see launch_pipeline)
// (the body of k_emit; k_step runs it as one of its roles: bx / b from a flat block id, nbx = blocks per sweep)
// ORDERED: the descriptors come in pillar order from the previous launch's order role (k_step) instead of
// through the tile prefix (k_emit) -- a compile-time choice: the prefix path's registers (the lane's tile
// counts, the scan) would otherwise cost every instance a wave per SIMD
template <typename TIn, int MODE, int AUX, int CAP, bool ORDERED>
__device__ __forceinline__ void emit_body(const EmitArgs &a, WaveLds<TIn, CAP> *lds, int bx, int b, int nbx) {
  using Rec = typename Rec4<TIn>::type;
  constexpr int CAPW = CAP;        // (shadows the file-scope default inside this body)
  constexpr int kPre = CAP / 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int N = a.N, P = a.P;
  WaveLds<TIn, CAP> &L = lds[w];
  // wave-uniform by construction; readfirstlane lets the compiler keep everything
  // derived from it (slab geometry, line masks, buffer offsets) in SGPRs
  const int p0 = __builtin_amdgcn_readfirstlane((bx * kEmitWaves + w) * KW);
  if (p0 >= P) return;
  [[maybe_unused]] u64 *stamps = a.stamps;
  [[maybe_unused]] const int stamp_nx = nbx, stamp_bx = bx, stamp_by = b;
  PP_STAMP_E(0);
  const int kw_eff = min(KW, P - p0);
  // (1) pillar descriptors.  The pillar index of a cell = occupied cells of all earlier
  //     tiles + its place in its tile's list; k_tile left the former open, so every wave
  //     prefix-sums the tiles' totals itself (lane l owns nv consecutive tiles: no LDS, no
  //     barrier) and finds the tiles of its KW pillars with ballots.  (k_step: the previous launch's
  //     ORDER role already wrote the descriptors in pillar order -- one load instead of that chain.)
  int4 m = make_int4(-1, 0, 0, 0);
  if constexpr (ORDERED) {
    // the sweep's totals and the descriptors are loaded side by side (a descriptor beyond the pillar count is
    // whatever an earlier batch left there -- readable, and dropped below): one round trip, not two
    const int2 tt = a.ordered_totals[b];
    int4 mo = m;
    if (lane < KW && p0 + lane < P) mo = a.ordered_meta[(int64_t)b * P + p0 + lane];
    if (bx == 0 && w == 0 && lane == 0) a.totals[b] = tt;
    const int npil_o = min(tt.x, P);
    if (lane < KW && p0 + lane < npil_o) m = mo;
  } else {
  const int ntiles = a.g.ntiles;
  const int nv = (ntiles + kWave - 1) / kWave;  // tiles per lane, <= 64
  const u64 *agg = a.tile_agg + (int64_t)b * ntiles;
  u64 mine = 0;
  constexpr int kCv = 4;
  unsigned cv[kCv] = {0u, 0u, 0u, 0u};  // the lane's tiles' occupied-cell counts (all of them when nv <= 4)
  for (int e = 0; e < nv; ++e) {
    const int t = lane * nv + e;
    if (t < ntiles) {
      const u64 v = agg[t];
      mine += v;
#pragma unroll
      for (int i = 0; i < kCv; ++i)
        if (e == i) cv[i] = (unsigned)(v & 0xFFFFFFFFull);
    }
  }
  const u64 inc = wave_scan_2x32(mine);
  const u64 tot = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(inc >> 32), kWave - 1) << 32) |
                  (unsigned)__builtin_amdgcn_readlane((int)(inc & 0xFFFFFFFFull), kWave - 1);
  const unsigned lane_excl = (unsigned)((inc - mine) & 0xFFFFFFFFull);  // occupied cells before my tiles
  if (bx == 0 && w == 0 && lane == 0) {
    a.totals[b] = make_int2((int)(tot & 0xFFFFFFFFull), (int)(tot >> 32));
    if (MODE == kModeCompact && a.totals_host) a.totals_host[b] = make_int2((int)(tot & 0xFFFFFFFFull), (int)(tot >> 32));
  }
  const int npil = min((int)(tot & 0xFFFFFFFFull), P);
  PP_STAMP_E(1);
#pragma unroll
  for (int k = 0; k < KW; ++k) {
    const unsigned p = (unsigned)(p0 + k);
    if ((int)p >= npil) break;  // uniform
    // the last lane whose first tile starts at or before p holds p's tile
    const int owner = __popcll(__ballot(lane_excl <= p)) - 1;
    int tile_k = 0;
    unsigned before_k = 0;
    if (lane == owner) {
      unsigned run = lane_excl;
      tile_k = lane * nv;
      for (int e = 0; e < nv; ++e) {
        const int t = lane * nv + e;
        unsigned c = 0;
        if (e < kCv) {
#pragma unroll
          for (int i = 0; i < kCv; ++i)
            if (e == i) c = cv[i];
        } else if (t < ntiles) {
          c = (unsigned)(agg[t] & 0xFFFFFFFFull);
        }
        if (run + c > p) {
          tile_k = t;
          break;
        }
        run += c;
      }
      before_k = run;
    }
    const int owner_s = __builtin_amdgcn_readfirstlane(owner);  // uniform: v_readlane, not a permute
    tile_k = __builtin_amdgcn_readlane(tile_k, owner_s);
    before_k = (unsigned)__builtin_amdgcn_readlane((int)before_k, owner_s);
    if (lane == k) {
      m = a.tile_meta[((int64_t)b * ntiles + tile_k) * (1 << a.g.tile_shift) + (p - before_k)];
      if (MODE == kModeCompact) {
        a.pillar_meta[(int64_t)b * P + p] = m;
        if (a.pillar_meta_host) a.pillar_meta_host[(int64_t)b * P + p] = m;
      }
    }
  }
  }  // tile-prefix path
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
  // The counts come straight from the lanes that hold the descriptors, and the bucket read is issued
  // BEFORE the descriptors go to LDS (the hand-off's fence would otherwise hold the loads back):
  // their latency covers the canvas arithmetic and the LDS writes below.
  int cnts[KW], segbeg[KW], segpad[KW];
  int T = 0, Tpad = 0;
#pragma unroll
  for (int k = 0; k < KW; ++k) {
    cnts[k] = __builtin_amdgcn_readlane(m.z, k);  // lane k holds pillar k's descriptor: no LDS round trip
    segbeg[k] = T;       // position in the CSR range
    segpad[k] = Tpad;    // position in the sorted LDS arrays: 4-aligned bucket starts
    T += cnts[k];
    Tpad += (cnts[k] + 3) & ~3;
  }
  const bool pooled = (Tpad <= CAPW);
  // (2) the wave's pooled bucket in one coalesced read: consecutive pillars own
  //     consecutive CSR ranges
  Rec rec_r[kPre];
#pragma unroll
  for (int it = 0; it < kPre; ++it) rec_r[it].x = rec_r[it].y = rec_r[it].z = rec_r[it].w = 0;
  if (pooled && T > 0) {
    // the first occupied pillar's start (empty rows only follow occupied ones)
    const int start0 = __builtin_amdgcn_readlane(m.y, 0);
    const Rec *srec = reinterpret_cast<const Rec *>(a.sorted_pts) + (int64_t)b * a.ncap + start0;
#pragma unroll
    for (int it = 0; it < kPre; ++it)
      if (lane + it * kWave < T) rec_r[it] = srec[lane + it * kWave];
  }
  if (lane < KW) {
    L.slot[lane] = m.x;
    L.start[lane] = m.y;
    L.cnt[lane] = m.z;
    L.live[lane] = min(m.z, N);
    double cx = 0, cy = 0;
    if (m.z > 0) pillar_canvas(m.x, a.g, cx, cy);
    L.cx[lane] = cx;
    L.cy[lane] = cy;
    if constexpr (MODE == kModePfn) {
      // the pillar's canvas pixel, once, on the lane that holds its descriptor (the f64 -> int64 conversions, the bounds
      // test and the 64-bit index arithmetic per pillar on EVERY lane were most of pfn_finish's 1.1 us)
      int pix = -1;
      if (m.z > 0 && a.canvas) {
        const int64_t col = (int64_t)cx, row = (int64_t)cy;
        if (row >= 0 && row < a.canvas_h && col >= 0 && col < a.canvas_w) pix = (int)(row * a.canvas_w + col);
      }
      L.pix[lane] = pix;
    }
  }
  wave_sync();
  PP_STAMP_E(2);
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
        // pooled together; or, when the pool overflowed, its LIVE points staged in a run (emit_live_run)
        if (cnts[k] > 0 && (pooled || ((min(cnts[k], N) + 3) & ~3) <= CAPW)) sg.pooled |= 1u << k;
      }
      // Everything fits the LDS pool (the normal case): ONE store pass after the point
      // phase writes every line whole.  (Measured: a separate zero pass before the
      // bucket data arrives plus a late pass for the live lines costs 10 us of 34 at
      // 4 sweeps per launch -- the second pass's issue slots, not its bytes.)
      if (!pooled || T == 0) {
        sg.late_lines = slab_late_lines(sg, sg.pooled);
        if constexpr (MODE == kModeDenseVec4) store_slab<kPassEarly, TIn, AUX>(L, sg, rs, lane, segbeg);
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
    int j = sb;
    for (; j + 1 < sb + live; j += 2) {  // two points per trip: their LDS reads do not wait for each other
      float xv[9], yv[9];
#pragma unroll
      for (int d = 0; d < 9; ++d) {
        xv[d] = L.u.feat[d][j];
        yv[d] = L.u.feat[d][j + 1];
      }
      const float r0 = pfn_relu_conv(acc, xv), r1 = pfn_relu_conv(acc, yv);
      mx = fmaxf(mx, fmaxf(r0, r1));
      mn = fminf(mn, fminf(r0, r1));
    }
    for (; j < sb + live; ++j) {
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
      const int64_t hw = (int64_t)a.canvas_h * a.canvas_w;
      float *cb = a.canvas_nhwc ? a.canvas + (int64_t)b * hw * kPfnChannels + lane
                                : a.canvas + ((int64_t)b * kPfnChannels + lane) * hw;
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        const int pix = L.pix[k];   // (-1 for an empty row: rows beyond kw_eff hold no descriptor and are empty)
        if (pix < 0) continue;
        if (a.canvas_nhwc)
          cb[(int64_t)pix * kPfnChannels] = yv[k];
        else
          cb[pix] = yv[k];
      }
    }
  };
  // (4) the points
  if (T == 0) {
    if constexpr (MODE == kModePfn) pfn_finish();
    return;
  }
  PP_STAMP_E(3);
  if (pooled) {
    emit_group<TIn, MODE, CAP>(L, a, b, p0, 0, KW, lane, rec_r, segbeg, segpad, cnts, T);
    PP_STAMP_E(4);
    if constexpr (MODE == kModeDenseVec4) store_slab<kPassAll, TIn, AUX>(L, sg, rs, lane, segpad);
    if constexpr (MODE == kModePfn) {
#pragma unroll
      for (int k = 0; k < KW; ++k) pfn_fold(k, segpad[k], min(cnts[k], N));
    }
  } else {
    // The pool overflowed: crowded cells (BASELINE config 1's 100 x 100 grid, row-major strips through the middle of
    // the cloud).  Round 5 took greedy runs of whole buckets -- load, stage, chain, features, store, run after run, a
    // bucket beyond the pool on its own -- and the emit-wave stamps of config 1's shapes showed EVERY such wave (one in
    // twelve) at 13-23 us against a median of 6.5: the launch was those waves.  Two passes now:
    //  (a) the running means of ALL the wave's occupied pillars side by side (streamed_means: slices of every bucket
    //      through the pool, the 3 KW chains in as many lanes -- the wave pays for its LONGEST bucket, not their sum);
    //  (b) the features.  Only the first min(count, N) points of a pillar are ever emitted, so the runs are formed
    //      over the LIVE points: more pillars per run, no chain operands to stage (and their four f64 divisions per
    //      point only in (a)), a lane loads its point straight from the CSR array and keeps it in registers.
    unsigned occ = 0;
#pragma unroll
    for (int kk = 0; kk < KW; ++kk)
      if (cnts[kk] > 0) occ |= 1u << kk;
    streamed_means<TIn, CAP>(L, a, b, occ, lane);
    PP_STAMP_E(4);
    int k = 0;
#pragma unroll 1
    while (k < KW) {
      if (cnts[k] == 0) {
        ++k;
        continue;
      }
      if (((min(cnts[k], N) + 3) & ~3) > CAPW) {  // more LIVE points than the pool holds (N beyond the pool)
        if constexpr (MODE == kModePfn) {
          float mx = -INFINITY, mn = INFINITY;
          emit_big_pillar<TIn, MODE, CAP>(L, a, b, k, p0 + k, lane, &acc, &mx, &mn);
#pragma unroll
          for (int kk = 0; kk < KW; ++kk)
            if (kk == k) {
              acc.rmax[kk] = fmaxf(acc.rmax[kk], mx);
              acc.rmin[kk] = fminf(acc.rmin[kk], mn);
            }
        } else {
          emit_big_pillar<TIn, MODE, CAP>(L, a, b, k, p0 + k, lane);
        }
        wave_sync();
        ++k;
        continue;
      }
      int gpad = 0;
      int spg[KW], lvg[KW];
#pragma unroll
      for (int kk = 0; kk < KW; ++kk) {
        spg[kk] = 0;
        lvg[kk] = 0;
      }
      while (k < KW && gpad + ((min(cnts[k], N) + 3) & ~3) <= CAPW) {  // (an empty row joins any run)
        const int lv = min(cnts[k], N);
#pragma unroll
        for (int kk = 0; kk < KW; ++kk)
          if (kk == k) {
            spg[kk] = gpad;
            lvg[kk] = lv;
          }
        gpad += (lv + 3) & ~3;
        ++k;
      }
      unsigned gmask = 0;
#pragma unroll
      for (int kk = 0; kk < KW; ++kk)
        if (lvg[kk] > 0) gmask |= 1u << kk;
      emit_live_run<TIn, MODE, CAP>(L, a, b, p0, lane, spg, lvg, gpad);
      if constexpr (MODE == kModePfn) {
#pragma unroll
        for (int kk = 0; kk < KW; ++kk)
          if (lvg[kk] > 0) pfn_fold(kk, spg[kk], lvg[kk]);
        wave_sync();
      }
      if (MODE == kModeDenseVec4) {
        // late pass restricted to this run's lines: a line shared with a pillar of
        // another run is completed by that run's own pass (both write identical
        // zeros outside their own head groups)
        SlabGeom s1 = sg;
        s1.pooled = gmask;
        s1.late_lines = slab_late_lines(sg, gmask);
        if constexpr (MODE == kModeDenseVec4) store_slab<kPassLate, TIn, AUX>(L, s1, rs, lane, spg);
        wave_sync();
      }
    }
  }
  PP_STAMP_E(5);
  if constexpr (MODE == kModePfn) pfn_finish();
  PP_STAMP_E(6);
}

#ifndef PP_EMIT_MINWAVES
#define PP_EMIT_MINWAVES 4
#endif
constexpr int emit_cap(int mode) { return mode == kModePfn ? 64 : CAPW; }
#ifndef PP_PFN_MINWAVES
#define PP_PFN_MINWAVES 6
#endif
constexpr int emit_minwaves(int mode) { return mode == kModePfn ? PP_PFN_MINWAVES : PP_EMIT_MINWAVES; }
template <typename TIn, int MODE, int AUX = 0>
__global__ __launch_bounds__(kEmitThreads, emit_minwaves(MODE)) void k_emit(EmitArgs a) {
  __shared__ WaveLds<TIn, emit_cap(MODE)> lds[kEmitWaves];
  emit_body<TIn, MODE, AUX, emit_cap(MODE), false>(a, lds, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}

// ------------------------------------------------------------------------- //
// k_step: the four stages of four CONSECUTIVE batches in one launch           //
// ------------------------------------------------------------------------- //
// Software pipelining across calls.  k_split / k_tile are latency chains that leave the memory system
// idle, k_emit is the dense store; inside one call they depend on each other, but split(batch i),
// tile(batch i-1), order(batch i-2: the tiles' descriptor lists copied into pillar order) and
// emit(batch i-3) do not.  One launch runs all four as ROLES of one grid -- the block id decides: a
// prefetch role first, then groups of {one binning block, one emit block} (tile role, order role,
// split role: the few latency-bound ones are dispatched early and run beside the thousands that
// store), then the remaining emit blocks -- and the launch boundary is the only synchronisation:
// every role reads what a role of the PREVIOUS launch wrote for its batch.  One launch per call
// instead of three, one drain instead of three, and the binning chains hide behind the store.
// All roles run 256-thread workgroups (emit: 4 waves x 4 pillars as in k_emit; tile: 4 waves; split:
// 4 physical waves x 4 virtual waves) and share the dynamic LDS.
struct SplitRole {
  const void *pts;
  int64_t sweep_stride;
  NPoints np;
  GridGeom g;
  int ncap, nchunks_cap, nchunks;
  int *kslot;
  void *kpts;
  int2 *mat;
};
struct TileRole {
  NPoints np;
  GridGeom g;
  int ncap, nchunks_cap;
  const int *kslot;
  const void *kpts;
  const int2 *mat;
  void *sorted_pts;
  int4 *tile_meta;
  u64 *tile_agg;
};
// What the prefetch role streams through (k_step): the arrays the emit role reads were written by the
// PREVIOUS launch; when other work ran in between (the network's activations: a GiB per step) they are in HBM
// again, and the role is a chain of dependent loads -- every hop would pay an HBM miss under a full store
// load.  A few workgroups at the head of the grid read those arrays once, front to back, with many loads in
// flight: the lines land in the memory-side Infinity Cache before most of the chains ask for them.  (The list
// can also hold the binning roles' inputs -- PP_STEP_PREFETCH_SETS -- which paid while binning and emit blocks
// alternated; with the binning blocks first they fetch for themselves at the same moment: see step_impl.)
struct PrefetchRole {
  const void *ptr[7];
  unsigned n16[7];       // 16-byte units
  const u64 *tile_agg;   // [nlists] {points << 32 | occupied cells}: how much of each tile's list is in use
  const int4 *tile_meta;
  int nlists, list_stride;
};
// ORDER role: one wave per (sweep, tile) copies the tile's descriptor list to its place in pillar order (the
// exclusive prefix of the tiles' occupied-cell counts: the one thing that depends on other tiles, known now
// that the previous launch's tile role has finished), capped at P; the last tile's wave leaves the totals.
// The emit role of the NEXT launch then starts with one descriptor load instead of prefix + search + load.
struct OrderRole {
  GridGeom g;
  int P, B, b0;   // this launch: sweeps [b0, b0 + B) of the batch
  const int4 *tile_meta;
  const u64 *tile_agg;
  int4 *ordered_meta;
  int2 *ordered_totals;
};
__device__ __forceinline__ void order_body(const OrderRole &o, int blk) {
  const int lane = threadIdx.x & 63, wid = blk * kEmitWaves + (threadIdx.x >> 6);
  const int ntiles = o.g.ntiles;
  if (wid >= o.B * ntiles) return;
  const int b = o.b0 + wid / ntiles, t = wid - (wid / ntiles) * ntiles;
  const u64 *agg = o.tile_agg + (int64_t)b * ntiles;
  const int nv = (ntiles + kWave - 1) / kWave;
  u64 before = 0, all = 0;
  for (int e = 0; e < nv; ++e) {
    const int tt = lane * nv + e;
    if (tt < ntiles) {
      const u64 v = agg[tt];
      all += v;
      if (tt < t) before += v & 0xFFFFFFFFull;
    }
  }
  const unsigned before_t = (unsigned)__builtin_amdgcn_readlane((int)wave_scan_u32((unsigned)before), kWave - 1);
  const unsigned cells_t = (unsigned)(agg[t] & 0xFFFFFFFFull);
  const int4 *lst = o.tile_meta + ((int64_t)b * ntiles + t) * (1 << o.g.tile_shift);
  int4 *dst = o.ordered_meta + (int64_t)b * o.P;
  for (unsigned k = lane; k < cells_t; k += kWave) {
    const unsigned p = before_t + k;
    if (p < (unsigned)o.P) dst[p] = lst[k];
  }
  if (t == ntiles - 1) {
    const u64 tot = wave_scan_2x32(all);
    if (lane == kWave - 1) o.ordered_totals[b] = make_int2((int)(tot & 0xFFFFFFFFull), (int)(tot >> 32));
  }
}

struct StepArgs {
  int n_pref_blocks;
  int n_unscatter_blocks;  // fused feature-net form: un.nblocks workgroups per sample of the canvas being cleared
  UnscatterArgs un;
  int n_order_blocks;
  OrderRole o;
  PrefetchRole pf;
  int n_tile_blocks, n_split_blocks, emit_nbx;
  int tile_xcd_map;  // the tile role takes its tiles XCD by XCD (tile_of_block): step_impl, by the size of its gathers
  int tile_b0, split_b0, emit_b0;  // first sweep of each role's batch in this launch (a call whose dense output is
                                   // beyond the Infinity Cache goes out as several launches, a few sweeps each)
  int mix, mix_groups;  // block order: mix_groups groups of {1 binning block, mix-1 emit blocks}, then the rest
  TileRole t;
  SplitRole s;
  EmitArgs e;
};
constexpr int kStepChipSlots = 5 * 256;  // k_step workgroups the chip holds at once (LDS and VGPRs: 5 per CU)
constexpr int kStepWaves = 4;
constexpr int kStepThreads = kStepWaves * kWave;
static_assert(kStepWaves == kEmitWaves, "the emit role is k_emit's workgroup");

#ifndef PP_STEP_PFN_MINWAVES
#define PP_STEP_PFN_MINWAVES 5
#endif
constexpr int step_minwaves(int mode) { return mode == kModePfn ? PP_STEP_PFN_MINWAVES : 5; }
template <int MODE, int AUX>
__global__ __launch_bounds__(kStepThreads, step_minwaves(MODE)) void k_step(StepArgs a) {   // five waves per SIMD: <= 96 VGPRs
  extern __shared__ __attribute__((aligned(32))) unsigned char step_smem[];
  int id = (int)blockIdx.x;
  if (id < a.n_pref_blocks) {
    // prefetch role: stream the previous launch's arrays into the cache hierarchy (values unused)
    const unsigned worker = (unsigned)id * kStepThreads + threadIdx.x, nworkers = (unsigned)a.n_pref_blocks * kStepThreads;
    unsigned acc = 0;
    // the occupied heads of the tiles' descriptor lists: one wave per list
    for (int t = (int)(worker >> 6); t < a.pf.nlists; t += (int)(nworkers >> 6)) {
      const unsigned c = (unsigned)(a.pf.tile_agg[t] & 0xFFFFFFFFull);
      const int4 *lst = a.pf.tile_meta + (int64_t)t * a.pf.list_stride;
      for (unsigned e = threadIdx.x & 63; e < c; e += kWave) acc ^= (unsigned)lst[e].x;
    }
#pragma unroll 1
    for (int r = 0; r < 7; ++r) {
      const uint4 *p = reinterpret_cast<const uint4 *>(a.pf.ptr[r]);
      const unsigned n = a.pf.n16[r];
      for (unsigned i = worker; i < n; i += 8u * nworkers) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const unsigned j = i + (unsigned)u * nworkers;
          v[u] = j < n ? p[j] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x;
      }
    }
    if (acc == 0x9E3779B9u && a.emit_nbx < 0) a.e.idx_out[0] = (long long)acc;  // never: keeps the loads alive
    return;
  }
  id -= a.n_pref_blocks;
  if constexpr (MODE == kModePfn) {
    // CLEAR role (fused feature-net form): zero the pixels of the OTHER canvas -- the one the previous call
    // filled and the network has consumed since; its indices say which.  The canvas this launch's emit role
    // writes was cleared the same way one call ago, so neither role waits for the other.
    if (id < a.n_unscatter_blocks) {
      unscatter_body(a.un, id % a.un.nblocks, id / a.un.nblocks, kStepThreads);
      return;
    }
    id -= a.n_unscatter_blocks;
  }
  // Block order.  Workgroups are dispatched in id order.  The binning roles' (tile, order, split) are few and
  // latency-bound, the emit role's many and store-bound; a binning chain that starts late is the launch's
  // tail.  mix_groups = 0 (host side: the binning blocks fit the chip at once): all binning blocks first,
  // then the emit blocks.  mix_groups > 0 (more binning blocks than that, or PP_STEP_MIX = m >= 2): the grid
  // starts with groups of one binning block and m-1 emit blocks, so that the stores also flow from the first
  // microsecond; what is left of either kind follows.
  {
    const int nbin = a.n_tile_blocks + a.n_order_blocks + a.n_split_blocks;
    const int head = a.mix_groups * a.mix;
    if (id < head) {
      const int grp = id / a.mix, j = id - grp * a.mix;
      id = j == 0 ? grp : nbin + grp * (a.mix - 1) + (j - 1);
    } else {
      const int r = id - head;                    // the rest: remaining binning blocks, then remaining emit blocks
      const int bin_left = nbin - a.mix_groups;
      id = r < bin_left ? a.mix_groups + r : nbin + a.mix_groups * (a.mix - 1) + (r - bin_left);
    }
  }
  if (id < a.n_tile_blocks) {
    const int nt = a.t.g.ntiles;
    // (tile_xcd_map: set by step_impl once the tile role's gathers exceed what the L2s hold; the logical id is what is
    // mapped -- under the mixed block order binning block g sits at hardware block g * mix, so blocks with equal g % 8
    // still share an XCD, whichever one it is)
    const int t_blk = id - (id / nt) * nt;
    const int b = a.tile_b0 + id / nt, tile = a.tile_xcd_map ? tile_of_block(t_blk, nt) : t_blk;
    tile_body<float, kStepWaves>(a.t.np, a.t.g, a.t.ncap, a.t.nchunks_cap, a.t.kslot,
                                 reinterpret_cast<const float4 *>(a.t.kpts), a.t.mat,
                                 reinterpret_cast<float4 *>(a.t.sorted_pts), a.t.tile_meta, a.t.tile_agg,
                                 nullptr, reinterpret_cast<unsigned *>(step_smem), tile, b);
    return;
  }
  id -= a.n_tile_blocks;
  if (id < a.n_order_blocks) {
    order_body(a.o, id);
    return;
  }
  id -= a.n_order_blocks;
  if (id < a.n_split_blocks) {
    const int nc = a.s.nchunks;
    const int b = a.split_b0 + id / nc, chunk = id - (id / nc) * nc;
    split_body<float, kStepWaves>(reinterpret_cast<const float *>(a.s.pts), a.s.sweep_stride, 4, 1, 1, a.s.np,
                                  a.s.g, a.s.ncap, a.s.nchunks_cap, a.s.kslot,
                                  reinterpret_cast<float4 *>(a.s.kpts), a.s.mat, step_smem, chunk, b);
    return;
  }
  id -= a.n_split_blocks;
  const int b = a.emit_b0 + id / a.emit_nbx, bx = id - (id / a.emit_nbx) * a.emit_nbx;
  emit_body<float, MODE, AUX, emit_cap(MODE), true>(a.e, reinterpret_cast<WaveLds<float, emit_cap(MODE)> *>(step_smem), bx,
                                                    b, a.emit_nbx);
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

int make_grid(const pp_voxel_params_t *prm, GridGeom *g, int step_mode) {
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
  g->ncells = (int)nc;
  // tiles: the smallest power-of-two run of slots that keeps the split at <= kTargetTiles bins;
  // a k_tile workgroup holds its tile's cells in LDS, a k_split workgroup a byte histogram per bin
  int ts = kMinTileSlots;
  static const int forced_tiles = [] {  // development knob
    const char *e = getenv("PP_TARGET_TILES");
    const int v = e ? atoi(e) : 0;
    return v >= 16 && v <= kMaxTiles ? v : 0;
  }();
  // row-major tiles are strips of the plane and lidar clouds are centre-heavy: finer tiles
  // (at most 2048 slots) bound the crowded ones; scrambled tiles are uniform and fewer, larger
  // ones cost less.  Beyond kMaxTiles of them the tiles grow to their LDS limit either way.
  // k_step's tile role shares the launch's dynamic LDS with the emit role (27.5 KB per workgroup): tiles of
  // at most 2048 slots (30.7 KB with 4 waves) keep the emit role at 5 workgroups per CU.  Fewer, fatter
  // tile workgroups hold fewer of the chip's workgroup slots while they wait out their latency chains:
  // 123 tiles of 2048 slots beat 245 of 1024 (C2: 13.1 / 33.6 us per step against 14.7 / 36.8, B = 1 / 4).
  // Round 5, k_step with the row-major order: tiles of 512 slots (489 at C2) -- its crowded strips were the launch
  // (C2: 53.0 -> 41.9 us at 4 sweeps per launch, 26.6 -> 14.0 us at one; tools/lab/sweep_rowmajor.sh); the scrambled
  // order's uniform tiles stay at 2048 slots (38.5 us; 39.9 with 512).
  const int target_tiles = forced_tiles ? forced_tiles
                         : prm->order == PP_ORDER_ROW_MAJOR ? (step_mode ? 2 * kTargetTiles : kTargetTiles)
                                                            : kTargetTiles / 2;
  const int soft_cap = ((prm->order == PP_ORDER_ROW_MAJOR && !forced_tiles) || step_mode) ? 2048 : kMaxTileSlots;
  while ((nc + ts - 1) / ts > target_tiles && ts < soft_cap) ts *= 2;
  while ((nc + ts - 1) / ts > kMaxTiles && ts < kMaxTileSlots) ts *= 2;
  const long long nt = (nc + ts - 1) / ts;
  if (nt > kMaxTiles) {
    set_error("cell grid too large (%lld cells; limit %d)", nc, kMaxTiles * kMaxTileSlots);
    return PP_ERR_VALUE;
  }
  g->tile_shift = 0;
  while ((1 << g->tile_shift) < ts) ++g->tile_shift;
  g->ntiles = (int)nt;
  g->tile_bits = 0;
  while ((1ll << g->tile_bits) < nt) ++g->tile_bits;
  g->order = prm->order;
  g->mult = 1;
  g->mult_inv = 1;
  g->barrett = nc > 1 ? ~0ull / (unsigned long long)nc : 0ull;  // floor((2^64-1)/n) = floor(2^64/n) unless n | 2^64
  if (nc > 1 && (nc & (nc - 1)) == 0) g->barrett += 1;  // n a power of two: 2^64/n exactly
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
  size_t kslot, kpts, mat, sorted_pts, tile_meta, tile_agg, meta, totals, otot, stamps, bytes;
  int ncap;         // point capacity per sweep, a multiple of the split chunk
  int nchunks_cap;  // split chunks per sweep at capacity
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

VoxLayout vox_layout(int B, int64_t max_points, const GridGeom &g, int P, int rec_bytes) {
  VoxLayout l;
  l.ncap = (int)align_up((size_t)std::max<int64_t>(max_points, 1), kChunk);
  l.nchunks_cap = l.ncap / kChunk;
  const size_t ts = (size_t)1 << g.tile_shift;
  size_t off = 0;
  l.kslot = off;
  off = align_up(off + (size_t)B * l.ncap * 4, 256);
  l.kpts = off;
  off = align_up(off + (size_t)B * l.ncap * rec_bytes, 256);
  l.mat = off;
  off = align_up(off + (size_t)B * g.ntiles * l.nchunks_cap * 8, 256);
  l.sorted_pts = off;
  off = align_up(off + (size_t)B * l.ncap * rec_bytes, 256);
  l.tile_meta = off;  // a tile's list has at most min(tile slots, points) entries; sized for the former
  off = align_up(off + (size_t)B * g.ntiles * ts * 16, 256);
  l.tile_agg = off;
  off = align_up(off + (size_t)B * g.ntiles * 8, 256);
  l.meta = off;       // compact mode only (host drop-in)
  off = align_up(off + (size_t)B * P * 16, 256);
  l.totals = off;
  off = align_up(off + (size_t)B * 8, 256);
  l.otot = off;       // k_step: {cells, points} per sweep, left by the order role for the emit role
  off = align_up(off + (size_t)B * 8, 256);
  l.stamps = off;
#ifdef PP_STAMPS
  off = align_up(off + (size_t)B * std::max(std::max(g.ntiles, l.nchunks_cap), (P + KW * kEmitWaves - 1) / (KW * kEmitWaves)) * 16 * 64, 256);
#endif
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

// Makes the workspace fit (B, max_points, grid, P); zeroed whenever the layout changed
// (nothing depends on it: every array is written before it is read within one call).
int prepare_ws(pp_ctx *ctx, hipStream_t stream, int B, int64_t max_points,
               const GridGeom &g, int P, int rec_bytes, VoxLayout *out, int slot = 0) {
  VoxLayout l = vox_layout(B, max_points, g, P, rec_bytes);
  const unsigned long long key[6] = {(unsigned long long)B, (unsigned long long)l.ncap,
                                     ((unsigned long long)g.ntiles << 8) | (unsigned)g.tile_shift,
                                     (unsigned long long)P, (unsigned long long)l.bytes,
                                     (unsigned long long)rec_bytes};
  bool grew = false;
  int rc = ctx->vox_ws[slot].ensure(l.bytes, &grew);
  if (rc) return rc;
  if (grew || std::memcmp(key, ctx->vox_layout_key[slot], sizeof key) != 0) {
    PP_HIP_TRY(hipMemsetAsync(ctx->vox_ws[slot].ptr, 0, ctx->vox_ws[slot].bytes, stream));
    std::memcpy(ctx->vox_layout_key[slot], key, sizeof key);
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
                    int canvas_nhwc = 0, const long long *prev_idx = nullptr, int feat_pack = 0,
                    int4 *meta_direct = nullptr, int2 *totals_direct = nullptr) {
  constexpr int slot = 0;  // the plain calls' workspace (k_step's batches rotate through the others)
  using Rec = typename Rec4<TIn>::type;
  char *ws = static_cast<char *>(ctx->vox_ws[slot].ptr);
  int *kslot = reinterpret_cast<int *>(ws + l.kslot);
  Rec *kpts = reinterpret_cast<Rec *>(ws + l.kpts);
  int2 *mat = reinterpret_cast<int2 *>(ws + l.mat);
  Rec *sorted_pts = reinterpret_cast<Rec *>(ws + l.sorted_pts);
  int4 *tile_meta = reinterpret_cast<int4 *>(ws + l.tile_meta);
  u64 *tile_agg = reinterpret_cast<u64 *>(ws + l.tile_agg);
  int4 *meta = reinterpret_cast<int4 *>(ws + l.meta);
  int2 *totals = reinterpret_cast<int2 *>(ws + l.totals);

  // k_tile geometry: waves per tile from the mean population of a tile
  const long long per_tile = ((long long)maxn + g.ntiles - 1) / g.ntiles;
  const int tw = ctx->force_tile_waves ? ctx->force_tile_waves
               : per_tile <= 96 ? 4
               : per_tile <= (g.order == PP_ORDER_ROW_MAJOR ? 640 : 1024) ? 8 : 16;
  const int wi = tw == 4 ? 0 : tw == 8 ? 1 : 2;
  const size_t lds_split = split_lds_bytes(g.ntiles);
  const size_t lds_tile = tile_lds_bytes(1 << g.tile_shift, tw);
  // Dynamic LDS beyond 64 KiB needs the attribute.  hipFuncSetAttribute acts on the kernel function of
  // the DEVICE, whatever context asks: every instance is set ONCE per process and device, to its worst
  // case (the largest tile / the most split bins), so no context can ever lower another one's limit.
  {
    static std::atomic<unsigned> armed_mask[64];   // per device: bit = [f64 input][k_tile 4/8/16 waves | k_split]
    const int dev = ctx->device & 63;
    const unsigned bit_tile = 1u << ((sizeof(TIn) == 8 ? 4 : 0) + wi);
    const unsigned bit_split = 1u << ((sizeof(TIn) == 8 ? 4 : 0) + 3);
    if (!(armed_mask[dev].load(std::memory_order_acquire) & bit_tile)) {
      const void *fn = tw == 4 ? reinterpret_cast<const void *>(&k_tile<TIn, 4>)
                     : tw == 8 ? reinterpret_cast<const void *>(&k_tile<TIn, 8>)
                               : reinterpret_cast<const void *>(&k_tile<TIn, 16>);
      PP_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)tile_lds_bytes(kMaxTileSlots, tw)));
      armed_mask[dev].fetch_or(bit_tile, std::memory_order_release);
    }
    if (!(armed_mask[dev].load(std::memory_order_acquire) & bit_split)) {
      PP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_split<TIn>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)split_lds_bytes(kMaxTiles)));
      armed_mask[dev].fetch_or(bit_split, std::memory_order_release);
    }
  }
  const int nchunks = std::max(1, (maxn + kChunk - 1) / kChunk);
  u64 *stamps = nullptr;
#ifdef PP_STAMPS
  stamps = reinterpret_cast<u64 *>(ws + l.stamps);
  ctx->dbg_stamps_off = l.stamps;
  ctx->dbg_stamps_bytes = (size_t)B * (PP_STAMPS == 2 ? nchunks : PP_STAMPS == 3 ? (P + KW * kEmitWaves - 1) / (KW * kEmitWaves) : g.ntiles) * 16 * 64;
#endif
  // When the timing ring is armed every launch carries its own start/stop events
  // (hipExtLaunchKernelGGL binds them to the dispatch packet, so a pair brackets the
  // kernel alone, like a profiler's kernel trace, not the gaps around it).
  hipEvent_t ev0[3] = {nullptr, nullptr, nullptr}, ev1[3] = {nullptr, nullptr, nullptr};
  if (timed && ctx->ev_slots > 0) {
    for (int k = 0; k < 3; ++k) {
      ev0[k] = ctx->ev_start[k][ctx->ev_next];
      ev1[k] = ctx->ev_stop[k][ctx->ev_next];
    }
    if (ctx->ev_columns != 7) ctx->ev_count = 0;  // the ring held k_step launches: their entries have no split / tile pair
    ctx->ev_next = (ctx->ev_next + 1) % ctx->ev_slots;
    ctx->ev_count = std::min(ctx->ev_count + 1, ctx->ev_slots);
    ctx->ev_columns = 7;  // all three kernels recorded
  }
  hipExtLaunchKernelGGL((k_split<TIn>), dim3((unsigned)nchunks, (unsigned)B), dim3(kSplitThreads),
                        lds_split, stream, ev0[PP_KERNEL_SPLIT], ev1[PP_KERNEL_SPLIT], 0, pts,
                        sweep_stride, s0, s1, contig, np, g, l.ncap, l.nchunks_cap, kslot, kpts, mat,
                        stamps);
  UnscatterArgs un;
  std::memset(&un, 0, sizeof un);
  if (canvas && prev_idx) {
    un.prev_idx = prev_idx;
    un.canvas = canvas;
    un.P = P;
    un.h = canvas_h;
    un.w = canvas_w;
    un.nhwc = canvas_nhwc;
    un.nblocks = (P + kUnscatterPixels - 1) / kUnscatterPixels;
  }
  auto launch_tile = [&](auto kern) {
    hipExtLaunchKernelGGL(kern, dim3((unsigned)(g.ntiles + un.nblocks), (unsigned)B), dim3(tw * kWave), lds_tile,
                          stream, ev0[PP_KERNEL_TILE], ev1[PP_KERNEL_TILE], 0, np, g, l.ncap,
                          l.nchunks_cap, kslot, kpts, mat, sorted_pts, tile_meta, tile_agg, stamps, un);
  };
  if (tw == 4) launch_tile(&k_tile<TIn, 4>);
  else if (tw == 8) launch_tile(&k_tile<TIn, 8>);
  else launch_tile(&k_tile<TIn, 16>);
  EmitArgs a;
  a.g = g;
  a.np = np;
  a.P = P;
  a.N = N;
  a.ncap = l.ncap;
  a.tile_meta = tile_meta;
  a.tile_agg = tile_agg;
  a.pillar_meta = meta;
  a.ordered_meta = nullptr;
  a.ordered_totals = nullptr;
  a.totals = totals;
  a.sorted_pts = sorted_pts;
  a.out = out;
  a.idx_out = idx_out;
  a.feat_out = feat_out;
  a.feat_pack = feat_pack;
  a.pillar_meta_host = meta_direct;
  a.totals_host = totals_direct;
  a.pfn_w = pfn_w;
  a.pfn_out = pfn_out;
  a.canvas = canvas;
  a.canvas_h = canvas_h;
  a.canvas_w = canvas_w;
  a.canvas_nhwc = canvas_nhwc;
  a.stamps = stamps;
  const dim3 grid_emit((unsigned)((P + KW * kEmitWaves - 1) / (KW * kEmitWaves)), (unsigned)B);
  switch (mode) {
    case kModeDenseVec4: {
      // Store policy of the dense tensor.  Write-through (sc1) stores leave no dirty lines for the
      // end-of-kernel write-back: on k_emit's store pattern alone (tools/lab/fill_pattern.cpp) 43.2 MB take
      // 6.3 us against 7.4 us plain, 172.8 MB 20.8 / 22.6 us, 432 MB (beyond the Infinity Cache) 84 / 80 us.
      // In k_emit itself: 43.2 MB 10.4 / 11.5 us, 108 MB 21.5 / 22.0 us, 172.8 MB 29.3 / 29.0 us -- the
      // larger launches are not bound by their tail.  Write-through up to 128 MB.
      static const int forced = [] {  // development knob: PP_EMIT_SC1=0/1
        const char *e = getenv("PP_EMIT_SC1");
        return e ? (atoi(e) ? 1 : 0) : -1;
      }();
      const bool sc1 = forced >= 0 ? forced == 1 : (size_t)B * 36u * (size_t)P * (size_t)N <= kSc1MaxBytes;
      if (sc1)
        hipExtLaunchKernelGGL((k_emit<TIn, kModeDenseVec4, kAuxSc1>), grid_emit, dim3(kEmitThreads), 0, stream,
                              ev0[PP_KERNEL_EMIT], ev1[PP_KERNEL_EMIT], 0, a);
      else
        hipExtLaunchKernelGGL((k_emit<TIn, kModeDenseVec4, kAuxPlain>), grid_emit, dim3(kEmitThreads), 0, stream,
                              ev0[PP_KERNEL_EMIT], ev1[PP_KERNEL_EMIT], 0, a);
      break;
    }
    case kModeDenseScalar:
      hipExtLaunchKernelGGL((k_emit<TIn, kModeDenseScalar>), grid_emit, dim3(kEmitThreads), 0, stream,
                            ev0[PP_KERNEL_EMIT], ev1[PP_KERNEL_EMIT], 0, a);
      break;
    case kModePfn:
      hipExtLaunchKernelGGL((k_emit<TIn, kModePfn>), grid_emit, dim3(kEmitThreads), 0, stream,
                            ev0[PP_KERNEL_EMIT], ev1[PP_KERNEL_EMIT], 0, a);
      break;
    default:
      hipExtLaunchKernelGGL((k_emit<TIn, kModeCompact>), grid_emit, dim3(kEmitThreads), 0, stream,
                            ev0[PP_KERNEL_EMIT], ev1[PP_KERNEL_EMIT], 0, a);
      break;
  }
  {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      // whatever the slot holds is unknown now: lay it out (and clear it) again on the next call
      std::memset(ctx->vox_layout_key[slot], 0, sizeof ctx->vox_layout_key[slot]);
      set_error("voxelizer launch failed: %s", hipGetErrorString(e));
      return PP_ERR_HIP;
    }
  }
  return PP_OK;
}

// The host drop-in's way back for the features (pp_create_pillars_f64): piece `part` of `parts` of the compact records
// [0, end) from the device staging into device-visible host memory, 16 bytes a lane.  `end` -- how many points the call
// emitted -- is read from the descriptors k_emit has just left on the DEVICE (same stream), so the pieces are enqueued
// behind the kernels without the host having seen a single number: the round trip "descriptors back, then size and start
// the copy" (a wait, a copy engine's start-up) is gone, and what crosses the link is exactly the emitted records.
// The host cuts [0, end) at the same places (piece_begin).
constexpr int kCopyThreads = 256;
__host__ __device__ inline long long piece_begin(long long end, int part, int parts) { return end * part / parts; }
__global__ __launch_bounds__(kCopyThreads) void k_records_to_host(const int2 *totals, const int4 *meta, int P, int max_pillars,
                                                                   int rec, const char *src, char *dst, int part, int parts) {
  const int npil = min(totals[0].x, min(P, max_pillars));
  if (npil <= 0) return;
  const int4 last = meta[npil - 1];
  const long long end = (long long)last.y + last.z;
  const long long b0 = piece_begin(end, part, parts) * rec, b1 = piece_begin(end, part + 1, parts) * rec;  // multiples of 8
  // 16-byte units from the first 16-aligned byte; the (at most one) 8-byte piece at either end by one lane each
  const long long a0 = (b0 + 15) & ~15ll, a1 = b1 & ~15ll;
  const long long tid = (long long)blockIdx.x * kCopyThreads + threadIdx.x, nthr = (long long)gridDim.x * kCopyThreads;
  if (a0 >= a1) {  // (a piece shorter than a unit)
    for (long long o = b0 + tid * 8; o < b1; o += nthr * 8)
      *reinterpret_cast<uint2 *>(dst + o) = *reinterpret_cast<const uint2 *>(src + o);
    return;
  }
  if (tid == 0 && b0 < a0) *reinterpret_cast<uint2 *>(dst + b0) = *reinterpret_cast<const uint2 *>(src + b0);
  if (tid == 1 && a1 < b1) *reinterpret_cast<uint2 *>(dst + a1) = *reinterpret_cast<const uint2 *>(src + a1);
  for (long long o = a0 + tid * 16; o < a1; o += nthr * 16)
    *reinterpret_cast<uint4 *>(dst + o) = *reinterpret_cast<const uint4 *>(src + o);
}

}  // namespace
}  // namespace pp

using namespace pp;

#ifdef PP_STAMPS
// development builds only (tools/lab): streamed_means' slowest wave {total, chains, staging (shader clocks), steps, rounds}
extern "C" int pp_debug_means_prof(unsigned long long *host, int reset) {
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(pp::g_means_prof), sizeof pp::g_means_prof) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(pp::g_means_prof), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
// development builds only (tools/lab): k_tile's phase stamps of sweep 0, 8 per wave
extern "C" int pp_debug_stamps(pp_ctx_t *ctx, unsigned long long *host, int cap) {
  const size_t n = std::min((size_t)cap * 8, ctx->dbg_stamps_bytes);
  if (hipMemcpy(host, static_cast<char *>(ctx->vox_ws[0].ptr) + ctx->dbg_stamps_off, n,
                hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  return (int)(n / 8);
}
#endif

extern "C" int pp_voxelize_check(pp_ctx_t *ctx, void *stream_) {
  if (!ctx) {
    set_error("ctx is NULL");
    return PP_ERR_VALUE;
  }
  DeviceGuard guard(ctx->device);
  hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(stream_));
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) {
    std::memset(ctx->vox_layout_key, 0, sizeof ctx->vox_layout_key);
    for (auto &sb : ctx->step_batch) sb.valid = false;
    set_error("voxelizer: the stream reports %s", hipGetErrorString(e));
    return PP_ERR_HIP;
  }
  return PP_OK;
}

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
    char *ws = static_cast<char *>(ctx->vox_ws[0].ptr);
    PP_HIP_TRY(hipMemcpyAsync(num_cells_dev, ws + l.totals, (size_t)batch * 8,
                              hipMemcpyDeviceToDevice, stream));
  }
  return PP_OK;
}

static int check_dev_args(const char *what, int batch, const pp_voxel_params_t *prm, int64_t points_stride,
                          const int32_t *n_points, NPoints *np, int *maxn) {
  if (batch < 1 || batch > PP_MAX_BATCH) {
    set_error("%s: batch must be in [1,%d], got %d", what, PP_MAX_BATCH, batch);
    return PP_ERR_VALUE;
  }
  const int P = prm->max_pillars, N = prm->max_points_per_pillar;
  if (P < 1 || N < 1 || N > 65536 || (long long)P * N > 100000000ll) {
    set_error("%s: need 1 <= max_pillars, 1 <= max_points_per_pillar <= 65536 and P*N <= 1e8 "
              "(got P=%d N=%d)", what, P, N);
    return PP_ERR_VALUE;
  }
  if (points_stride < 0 || points_stride > INT_MAX / 2) {
    set_error("%s: points_stride out of range", what);
    return PP_ERR_VALUE;
  }
  std::memset(np, 0, sizeof *np);
  *maxn = 0;
  for (int b = 0; b < batch; ++b) {
    if (n_points[b] < 0 || n_points[b] > points_stride) {
      set_error("%s: n_points[%d]=%d outside [0, points_stride=%lld]", what, b, n_points[b],
                (long long)points_stride);
      return PP_ERR_VALUE;
    }
    np->n[b] = n_points[b];
    *maxn = std::max(*maxn, n_points[b]);
  }
  return PP_OK;
}

/* ---- software-pipelined mode: one k_step launch per call ---- */
namespace {
struct SlotArrays {
  int *kslot;
  float4 *kpts, *sorted_pts;
  int2 *mat, *totals, *otot;
  int4 *tile_meta, *meta;
  u64 *tile_agg;
};
SlotArrays slot_arrays(pp_ctx *ctx, int slot, const VoxLayout &l) {
  char *ws = static_cast<char *>(ctx->vox_ws[slot].ptr);
  SlotArrays a;
  a.kslot = reinterpret_cast<int *>(ws + l.kslot);
  a.kpts = reinterpret_cast<float4 *>(ws + l.kpts);
  a.mat = reinterpret_cast<int2 *>(ws + l.mat);
  a.sorted_pts = reinterpret_cast<float4 *>(ws + l.sorted_pts);
  a.tile_meta = reinterpret_cast<int4 *>(ws + l.tile_meta);
  a.tile_agg = reinterpret_cast<u64 *>(ws + l.tile_agg);
  a.totals = reinterpret_cast<int2 *>(ws + l.totals);
  a.otot = reinterpret_cast<int2 *>(ws + l.otot);
  a.meta = reinterpret_cast<int4 *>(ws + l.meta);
  return a;
}
int step_geometry(const pp_step_batch &sb, GridGeom *g, VoxLayout *l) {
  int rc = make_grid(&sb.prm, g, 1);
  if (rc) return rc;
  *l = vox_layout(sb.batch, std::max<int64_t>(sb.points_stride, 1), *g, sb.prm.max_pillars, 16);
  return PP_OK;
}
}  // namespace

namespace {
// the emit role as the fused feature net (pp_voxelize_step_pfn_canvas_dev): canvas instead of the dense tensor
struct StepPfn {
  const float *params;            // [64][12]
  float *canvas;                  // receives the batch that is due; all zero on entry
  int h, w, nhwc;
  float *clear_canvas;            // the other canvas (may be NULL) ...
  const int64_t *clear_indices;   // ... and the indices of the batch that filled it: its non-zero pixels
  int clear_batch;
};
}  // namespace

static int step_impl(pp_ctx_t *ctx, void *stream_, const float *points_dev, int64_t points_stride,
                     const int32_t *n_points, int batch, const pp_voxel_params_t *prm, float *pillars_dev,
                     int64_t *indices_dev, int32_t *num_cells_dev, int *emitted, const StepPfn *pfn) {
  if (emitted) *emitted = 0;
  if (!ctx) {
    set_error("ctx is NULL");
    return PP_ERR_VALUE;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DeviceGuard guard(ctx->device);
  pp_step_batch &sb_tile = ctx->step_batch[0], &sb_order = ctx->step_batch[1], &sb_emit = ctx->step_batch[2];
  // Stream order is what carries a batch from one role to the next: a call on another stream while batches
  // are in flight would race with the launch that wrote what it reads.
  if ((sb_tile.valid || sb_order.valid || sb_emit.valid) && stream != ctx->step_stream) {
    set_error("pp_voxelize_step_dev: pipeline in flight on another stream (drain it, or pp_voxelize_step_reset)");
    return PP_ERR_VALUE;
  }
  // (ctx->step_stream is set where the call can no longer be refused: every PP_ERR_VALUE below leaves the context as it was)
  if (sb_emit.valid) {
    if (!(pfn ? (void *)pfn->canvas : (void *)pillars_dev) || !indices_dev) {
      set_error("pp_voxelize_step*_dev: a batch is due, its output buffers are NULL");
      return PP_ERR_VALUE;
    }
    if (pfn && (!pfn->params || (reinterpret_cast<uintptr_t>(pfn->canvas) & 15) || pfn->h < 1 || pfn->w < 1 ||
                (double)pfn->h != sb_emit.prm.canvas_height)) {
      set_error("pp_voxelize_step_pfn_canvas_dev: canvas %dx%d: height must equal the due batch's canvas_height=%g "
                "(16-byte aligned tensor, parameters not NULL)", pfn->h, pfn->w, sb_emit.prm.canvas_height);
      return PP_ERR_VALUE;
    }
    if ((reinterpret_cast<uintptr_t>(pillars_dev) & 15) || (reinterpret_cast<uintptr_t>(indices_dev) & 7) ||
        (reinterpret_cast<uintptr_t>(num_cells_dev) & 7)) {
      set_error("device pointers must be 16-byte (pillars) / 8-byte (indices, num_cells) aligned");
      return PP_ERR_VALUE;
    }
  }
  StepArgs a;
  std::memset(&a, 0, sizeof a);
  size_t lds = 0;
  int rc;
  // split role: the new batch
  pp_step_batch sb_new;
  if (points_dev) {
    if (!n_points || !prm) {
      set_error("pp_voxelize_step_dev: NULL argument");
      return PP_ERR_VALUE;
    }
    if (reinterpret_cast<uintptr_t>(points_dev) & 15) {
      set_error("device pointers must be 16-byte aligned (points)");
      return PP_ERR_VALUE;
    }
    NPoints np;
    int maxn = 0;
    rc = check_dev_args("pp_voxelize_step_dev", batch, prm, points_stride, n_points, &np, &maxn);
    if (rc) return rc;
    sb_new.valid = true;
    sb_new.batch = batch;
    sb_new.maxn = maxn;
    std::memcpy(sb_new.n_points, np.n, sizeof sb_new.n_points);
    sb_new.points_stride = points_stride;
    sb_new.prm = *prm;
    sb_new.slot = ctx->step_next_slot;
    GridGeom g;
    VoxLayout l;
    rc = step_geometry(sb_new, &g, &l);
    if (rc) return rc;
    // the slot's last batch was emitted by the previous launch: resizing / clearing it is stream-ordered
    VoxLayout l2;
    rc = prepare_ws(ctx, stream, batch, std::max<int64_t>(points_stride, 1), g, prm->max_pillars, 16, &l2,
                    sb_new.slot);
    if (rc) return rc;
    const SlotArrays w = slot_arrays(ctx, sb_new.slot, l);
    a.s.pts = points_dev;
    a.s.sweep_stride = points_stride;
    a.s.np = np;
    a.s.g = g;
    a.s.ncap = l.ncap;
    a.s.nchunks_cap = l.nchunks_cap;
    a.s.nchunks = std::max(1, (maxn + kChunk - 1) / kChunk);
    a.s.kslot = w.kslot;
    a.s.kpts = w.kpts;
    a.s.mat = w.mat;
    a.n_split_blocks = a.s.nchunks * batch;
    lds = std::max(lds, split_lds_bytes(g.ntiles));
  }
  if (sb_tile.valid) {
    GridGeom g;
    VoxLayout l;
    rc = step_geometry(sb_tile, &g, &l);
    if (rc) return rc;
    const SlotArrays w = slot_arrays(ctx, sb_tile.slot, l);
    std::memcpy(a.t.np.n, sb_tile.n_points, sizeof a.t.np.n);
    a.t.g = g;
    a.t.ncap = l.ncap;
    a.t.nchunks_cap = l.nchunks_cap;
    a.t.kslot = w.kslot;
    a.t.kpts = w.kpts;
    a.t.mat = w.mat;
    a.t.sorted_pts = w.sorted_pts;
    a.t.tile_meta = w.tile_meta;
    a.t.tile_agg = w.tile_agg;
    a.n_tile_blocks = g.ntiles * sb_tile.batch;
    lds = std::max(lds, tile_lds_bytes(1 << g.tile_shift, kStepWaves));
  }
  if (sb_order.valid) {
    GridGeom g;
    VoxLayout l;
    rc = step_geometry(sb_order, &g, &l);
    if (rc) return rc;
    const SlotArrays w = slot_arrays(ctx, sb_order.slot, l);
    a.o.g = g;
    a.o.P = sb_order.prm.max_pillars;
    a.o.B = sb_order.batch;
    a.o.tile_meta = w.tile_meta;
    a.o.tile_agg = w.tile_agg;
    a.o.ordered_meta = w.meta;
    a.o.ordered_totals = w.otot;
    a.n_order_blocks = (g.ntiles * sb_order.batch + kEmitWaves - 1) / kEmitWaves;   // one wave per (sweep, tile)
  }
  int mode = kModeDenseVec4;
  bool sc1 = false;  // write-through stores: decided per launch, by the bytes it writes
  if (sb_emit.valid) {
    GridGeom g;
    VoxLayout l;
    rc = step_geometry(sb_emit, &g, &l);
    if (rc) return rc;
    const SlotArrays w = slot_arrays(ctx, sb_emit.slot, l);
    const int P = sb_emit.prm.max_pillars, N = sb_emit.prm.max_points_per_pillar;
    a.e.g = g;
    std::memcpy(a.e.np.n, sb_emit.n_points, sizeof a.e.np.n);
    a.e.P = P;
    a.e.N = N;
    a.e.ncap = l.ncap;
    a.e.tile_meta = w.tile_meta;
    a.e.tile_agg = w.tile_agg;
    a.e.totals = num_cells_dev ? reinterpret_cast<int2 *>(num_cells_dev) : w.totals;
    a.e.ordered_meta = w.meta;
    a.e.ordered_totals = w.otot;
    a.e.sorted_pts = w.sorted_pts;
    a.e.out = pillars_dev;
    a.e.idx_out = reinterpret_cast<long long *>(indices_dev);
    a.emit_nbx = (P + KW * kEmitWaves - 1) / (KW * kEmitWaves);
    mode = (N % 4 == 0 && N <= 4096) ? kModeDenseVec4 : kModeDenseScalar;
    lds = std::max(lds, pfn ? sizeof(WaveLds<float, emit_cap(kModePfn)>) * kEmitWaves : sizeof(WaveLds<float>) * kEmitWaves);
    if (pfn) {
      mode = kModePfn;
      a.e.out = nullptr;
      a.e.pfn_w = pfn->params;
      a.e.pfn_out = nullptr;
      a.e.canvas = pfn->canvas;
      a.e.canvas_h = pfn->h;
      a.e.canvas_w = pfn->w;
      a.e.canvas_nhwc = pfn->nhwc;
    }
  }
  if (pfn && pfn->clear_canvas && pfn->clear_indices && pfn->clear_batch > 0) {
    // the other canvas: only the pixels its last batch wrote are non-zero (same geometry as the due batch's)
    if ((reinterpret_cast<uintptr_t>(pfn->clear_canvas) & 15) || (reinterpret_cast<uintptr_t>(pfn->clear_indices) & 7) ||
        pfn->clear_batch > PP_MAX_BATCH || !sb_emit.valid) {
      set_error("pp_voxelize_step_pfn_canvas_dev: bad clear arguments (alignment, batch <= %d, and a batch must be due "
                "whose shapes describe the canvas)", PP_MAX_BATCH);
      return PP_ERR_VALUE;
    }
    a.un.prev_idx = reinterpret_cast<const long long *>(pfn->clear_indices);
    a.un.canvas = pfn->clear_canvas;
    a.un.P = sb_emit.prm.max_pillars;
    a.un.h = pfn->h;
    a.un.w = pfn->w;
    a.un.nhwc = pfn->nhwc;
    a.un.nblocks = (a.un.P + kUnscatterPixels - 1) / kUnscatterPixels;
    a.n_unscatter_blocks = a.un.nblocks * pfn->clear_batch;
  }
  if (a.emit_nbx == 0) a.emit_nbx = 1;
  // The tile role takes its tiles XCD by XCD (tile_of_block) once its gathers are beyond what the L2s keep anyway: at
  // BASELINE config 5 (800k points per launch: 16 MB of split records) k_step's FETCH_SIZE halves, 56.9 -> 29.0 MiB, and the
  // launch is 1.5-4 us shorter (five alternations on two boxes, tools/lab/knobs_c5.sh / knobs_shapes.sh); at configs[1]'s
  // 240k points (4.8 MB, which every XCD's 4 MB L2 nearly holds) it measures the same to 0.2-0.4 us worse: tile = block.
  static const int tile_xcd = [] {  // development knob: PP_STEP_TILE_XCD=0/1 (unset: by size)
    const char *e = getenv("PP_STEP_TILE_XCD");
    return e ? (atoi(e) ? 1 : 0) : -1;
  }();
  a.tile_xcd_map = tile_xcd >= 0 ? tile_xcd
                                 : (sb_tile.valid && (size_t)sb_tile.batch * (size_t)sb_tile.maxn * 20u > kTileXcdMapBytes);
  static const int pref_blocks = [] {  // development knob: PP_STEP_PREFETCH=<workgroups> (0 = off)
    const char *e = getenv("PP_STEP_PREFETCH");
    return e ? std::max(0, atoi(e)) : 128;
  }();
  static const int mix_env = [] {  // development knob: PP_STEP_MIX=m (1 = binning blocks first)
    const char *e = getenv("PP_STEP_MIX");
    return e ? std::max(1, atoi(e)) : 0;
  }();
  // A call CAN go out as several launches of a few sweeps each (sweeps are independent, so every role's
  // batch splits the same way).  Round 3's driver line suggested it should when the dense output exceeds the
  // 256 MB Infinity Cache: per sweep, one 172.8 MB sweep per launch ran at 0.73 of the roofline, four in
  // one launch at 0.64.  Measured in round 4 (tools/lab/sweep_sub.sh): four launches of one sweep each take
  // 154 us against 146 us for the one launch -- the one-sweep figure came from re-writing the SAME 172.8 MB
  // buffer call after call, which the memory-side cache absorbs; four launches writing four different
  // regions go to HBM like the single launch does, and pay three more launch ramps.  Default: one launch.
  static const size_t sub_bytes = [] {  // development knob: PP_STEP_SUB_MB (dense MB per launch; 0 = never split)
    const char *e = getenv("PP_STEP_SUB_MB");
    return e ? (size_t)std::max(0, atoi(e)) << 20 : kStepLaunchBytes;
  }();
  int n_launch = 1;
  if (sb_emit.valid && sub_bytes && !pfn) {
    const size_t per_sweep = 36u * (size_t)sb_emit.prm.max_pillars * (size_t)sb_emit.prm.max_points_per_pillar;
    const int per_launch = (int)std::max<size_t>(1, sub_bytes / std::max<size_t>(per_sweep, 1));
    n_launch = (sb_emit.batch + per_launch - 1) / per_launch;
  }
  const StepArgs whole = a;
  const int nt_t = sb_tile.valid ? whole.t.g.ntiles : 0, nt_o = sb_order.valid ? whole.o.g.ntiles : 0;
  ctx->step_stream = stream;  // every argument check has passed: from here on a failure is not a refusal
  for (int k = 0; k < n_launch; ++k) {
    auto part = [&](bool valid, int B, int *lo, int *n) {  // role's share of launch k: sweeps [lo, lo + n)
      *lo = valid ? (int)((long long)B * k / n_launch) : 0;
      *n = valid ? (int)((long long)B * (k + 1) / n_launch) - *lo : 0;
    };
    int tb, tn, ob, on, sb, sn, eb, en;
    part(sb_tile.valid, sb_tile.batch, &tb, &tn);
    part(sb_order.valid, sb_order.batch, &ob, &on);
    part(sb_new.valid, sb_new.batch, &sb, &sn);
    part(sb_emit.valid, sb_emit.batch, &eb, &en);
    a = whole;
    a.tile_b0 = tb;
    a.n_tile_blocks = nt_t * tn;
    a.o.b0 = ob;
    a.o.B = on;
    a.n_order_blocks = (nt_o * on + kEmitWaves - 1) / kEmitWaves;   // one wave per (sweep, tile)
    a.split_b0 = sb;
    a.n_split_blocks = sb_new.valid ? whole.s.nchunks * sn : 0;
    a.emit_b0 = eb;
#ifdef PP_STEP_SKIP_KNOB  // tools/lab timing builds only (results are garbage): PP_STEP_SKIP = bit mask of roles left out
    {
      static const int skip_all = [] { const char *e = getenv("PP_STEP_SKIP"); return e ? atoi(e) : 0; }();
      // PP_STEP_SKIP_AFTER=n: the first n launches run every role (the slots then hold real lists of the SAME cloud)
      static const int skip_after = [] { const char *e = getenv("PP_STEP_SKIP_AFTER"); return e ? atoi(e) : 0; }();
      static int launches = 0;
      const int skip = ++launches > skip_after ? skip_all : 0;
      if (skip & 1) a.n_tile_blocks = 0;
      if (skip & 2) a.n_order_blocks = 0;
      if (skip & 4) a.n_split_blocks = 0;
      if (skip & 8) a.n_unscatter_blocks = 0;   // the fused form's clear role
      if (skip & 16) en = 0;                    // the emit role: what the binning (and clear) roles take by themselves
    }
#endif
    const int n_emit = whole.emit_nbx * en;
    if (sb_emit.valid && !pfn) {
      static const int forced = [] {  // development knob: PP_EMIT_SC1=0/1
        const char *e = getenv("PP_EMIT_SC1");
        return e ? (atoi(e) ? 1 : 0) : -1;
      }();
      sc1 = forced >= 0 ? forced == 1 : (size_t)en * 36u * (size_t)a.e.P * (size_t)a.e.N <= kSc1MaxBytes;
    }
    {
      int r = 0;
      auto add = [&](const void *ptr, size_t per_sweep, int lo, int n) {
        a.pf.ptr[r] = static_cast<const char *>(ptr) + per_sweep * (size_t)lo;
        a.pf.n16[r] = (unsigned)(per_sweep * (size_t)n / 16);
        ++r;
      };
      std::memset(&a.pf, 0, sizeof a.pf);
      // Only the EMIT role's arrays are worth it (tools/lab/prefetch_sets.sh, round 4): its blocks start after the
      // binning blocks, so the lines are there when they ask.  The binning roles' own inputs (the order role's
      // lists, the tile role's split arrays, the split role's points) were in this list while binning and emit
      // blocks alternated; with the binning blocks FIRST they start in the same microsecond as the prefetch
      // blocks and fetch for themselves -- the prefetch only doubled that traffic (headline 34.9 -> 33.7 us
      // without, C5 B=4 111-118 -> 104-109 us).
      static const int pref_sets = [] {  // development knob: PP_STEP_PREFETCH_SETS = mask {1 emit, 2 order, 4 tile} arrays
        const char *e = getenv("PP_STEP_PREFETCH_SETS");
        return e ? atoi(e) : 1;
      }();
      if (pref_blocks > 0 && en > 0 && (pref_sets & 1)) {   // what the emit role reads
        add(a.e.ordered_meta, (size_t)a.e.P * 16, eb, en);
        add(a.e.sorted_pts, (size_t)a.e.ncap * 16, eb, en);
      }
      if (pref_blocks > 0 && on > 0 && (pref_sets & 2)) {  // ... the order role: the occupied heads of the tiles' lists
        a.pf.tile_agg = a.o.tile_agg + (size_t)ob * nt_o;
        a.pf.tile_meta = a.o.tile_meta + ((size_t)ob * nt_o << a.o.g.tile_shift);
        a.pf.nlists = on * nt_o;
        a.pf.list_stride = 1 << a.o.g.tile_shift;
      }
      if (pref_blocks > 0 && tn > 0 && (pref_sets & 4)) {   // what the tile role reads
        add(a.t.mat, (size_t)nt_t * a.t.nchunks_cap * 8, tb, tn);
        add(a.t.kslot, (size_t)a.t.ncap * 4, tb, tn);
        add(a.t.kpts, (size_t)a.t.ncap * 16, tb, tn);
      }
      static const int pref_points = [] {  // development knob: PP_STEP_PREFETCH_POINTS=0/1
        const char *e = getenv("PP_STEP_PREFETCH_POINTS");
        return e ? atoi(e) : 0;
      }();
      if (pref_blocks > 0 && sn > 0 && pref_points)   // ... and the split role: the caller's points
        add(a.s.pts, (size_t)a.s.sweep_stride * 16, sb, sn);
      a.n_pref_blocks = r > 0 || a.pf.nlists > 0 ? pref_blocks : 0;
    }
    if (mode != kModePfn) a.n_unscatter_blocks = 0;
    const long long nblocks = (long long)a.n_pref_blocks + a.n_unscatter_blocks + a.n_tile_blocks + a.n_order_blocks +
                              a.n_split_blocks + n_emit;
    {
      const int nbin = a.n_tile_blocks + a.n_order_blocks + a.n_split_blocks;
      // Default: every binning block ahead of the emit blocks.  After a network pass (the real caller: a GiB of
      // activations between two launches) the binning roles' chains run on cold lines and are the launch's
      // tail unless they start first: 39 -> 35 us in bench.py's end-to-end loop; in a voxelizer-only loop, where
      // their inputs are still cached, the 1:1 interleave (mix 2) was 2 us better (42.4 vs 40.6 us, round 4).
      // More binning blocks than the chip holds at once (5 workgroups per CU): "first" would mean rounds of
      // binning with no store in flight -- they are spread evenly over the launch instead, one per
      // n_emit / nbin emit blocks (C5 B=4, 3 400 binning blocks against 7 500: one per two, 117-121 -> 104-109 us;
      // one per one and one per three are both worse).
      const int mix_auto = nbin > kStepChipSlots && n_emit > 0 ? std::max(2, n_emit / nbin + 1) : 1;
      a.mix = mix_env ? mix_env : mix_auto;
      a.mix_groups = a.mix > 1 ? std::min(nbin, n_emit / (a.mix - 1)) : 0;
      if (a.mix < 2) a.mix = 2, a.mix_groups = 0;
    }
    if (nblocks <= 0) continue;
    if (nblocks > INT_MAX) {
      // (PP_ERR_VALUE means "refused, nothing launched, no state changed" to every caller -- voxelizer.py keeps its
      // pipeline on it; an earlier launch of THIS call may be enqueued here, so this can't-happen case is an internal
      // error: the caller resets the pipeline)
      set_error("pp_voxelize_step_dev: grid too large");
      return PP_ERR_INTERNAL;
    }
    const void *fn = mode == kModePfn          ? reinterpret_cast<const void *>(&k_step<kModePfn, kAuxPlain>)
                     : mode == kModeDenseScalar ? reinterpret_cast<const void *>(&k_step<kModeDenseScalar, kAuxPlain>)
                     : sc1                      ? reinterpret_cast<const void *>(&k_step<kModeDenseVec4, kAuxSc1>)
                                                : reinterpret_cast<const void *>(&k_step<kModeDenseVec4, kAuxPlain>);
    {
      // dynamic LDS beyond 64 KiB (16.7 M-cell grids) needs the attribute: once per process, device and
      // instance, at the worst case
      static std::atomic<unsigned> armed[64];
      const unsigned bit = mode == kModePfn ? 8u : mode == kModeDenseScalar ? 1u : sc1 ? 2u : 4u;
      const int dev = ctx->device & 63;
      if (!(armed[dev].load(std::memory_order_acquire) & bit)) {
        const size_t worst = std::max(std::max(tile_lds_bytes(kMaxTileSlots, kStepWaves), split_lds_bytes(kMaxTiles)),
                                      sizeof(WaveLds<float>) * kEmitWaves);
        PP_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)worst));
        armed[dev].fetch_or(bit, std::memory_order_release);
      }
    }
    // timing ring: ONE entry per call -- the start event rides on the call's first launch, the stop event on
    // its last, so the entry is what the call's launches took together (bench.py's per-call duration)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->ev_slots > 0) {
      if (k == 0) {
        if (ctx->ev_columns != (1 << PP_KERNEL_EMIT)) ctx->ev_count = 0;  // the ring held three-launch calls: do not
        ctx->ev_columns = 1 << PP_KERNEL_EMIT;                              // mix k_emit and k_step durations
        e0 = ctx->ev_start[PP_KERNEL_EMIT][ctx->ev_next];
      }
      if (k == n_launch - 1) {
        e1 = ctx->ev_stop[PP_KERNEL_EMIT][ctx->ev_next];
        ctx->ev_next = (ctx->ev_next + 1) % ctx->ev_slots;
        ctx->ev_count = std::min(ctx->ev_count + 1, ctx->ev_slots);
      }
    }
    const dim3 grid((unsigned)nblocks), block(kStepThreads);
    if (mode == kModePfn)
      hipExtLaunchKernelGGL((k_step<kModePfn, kAuxPlain>), grid, block, lds, stream, e0, e1, 0, a);
    else if (mode == kModeDenseScalar)
      hipExtLaunchKernelGGL((k_step<kModeDenseScalar, kAuxPlain>), grid, block, lds, stream, e0, e1, 0, a);
    else if (sc1)
      hipExtLaunchKernelGGL((k_step<kModeDenseVec4, kAuxSc1>), grid, block, lds, stream, e0, e1, 0, a);
    else
      hipExtLaunchKernelGGL((k_step<kModeDenseVec4, kAuxPlain>), grid, block, lds, stream, e0, e1, 0, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      for (auto &sb : ctx->step_batch) sb.valid = false;
      set_error("voxelizer launch failed: %s", hipGetErrorString(e));
      return PP_ERR_HIP;
    }
  }
  if (sb_emit.valid && emitted) *emitted = 1;
  // the batches move on: ordered -> due at the next call, tiled -> to be ordered, split -> to be tiled, new -> split
  sb_emit = sb_order;
  sb_order = sb_tile;
  sb_tile = sb_new;
  // slots 1..kVoxSlots-1 rotate: a slot comes round again only after its batch was emitted, which takes
  // as many calls as there are batches in flight
  static_assert(pp_ctx::kVoxSlots - 1 >= 4, "one workspace slot per batch in flight (split, tile, order, emit)");
  if (sb_new.valid) ctx->step_next_slot = ctx->step_next_slot % (pp_ctx::kVoxSlots - 1) + 1;
  return PP_OK;
}

extern "C" int pp_voxelize_step_dev(pp_ctx_t *ctx, void *stream_, const float *points_dev,
                                    int64_t points_stride, const int32_t *n_points, int batch,
                                    const pp_voxel_params_t *prm, float *pillars_dev,
                                    int64_t *indices_dev, int32_t *num_cells_dev, int *emitted) {
  return step_impl(ctx, stream_, points_dev, points_stride, n_points, batch, prm, pillars_dev, indices_dev,
                   num_cells_dev, emitted, nullptr);
}

extern "C" int pp_voxelize_step_pfn_canvas_dev(pp_ctx_t *ctx, void *stream_, const float *points_dev,
                                               int64_t points_stride, const int32_t *n_points, int batch,
                                               const pp_voxel_params_t *prm, const float *pfn_params_dev,
                                               int channels, float *canvas_dev, int canvas_h, int canvas_w,
                                               int channels_last, int64_t *indices_dev, int32_t *num_cells_dev,
                                               float *clear_canvas_dev, const int64_t *clear_indices_dev,
                                               int clear_batch, int *emitted) {
  if (channels != kPfnChannels) {
    set_error("the fused feature net is built for %d output channels (got %d)", kPfnChannels, channels);
    return PP_ERR_VALUE;
  }
  StepPfn pfn;
  pfn.params = pfn_params_dev;
  pfn.canvas = canvas_dev;
  pfn.h = canvas_h;
  pfn.w = canvas_w;
  pfn.nhwc = channels_last ? 1 : 0;
  pfn.clear_canvas = clear_canvas_dev;
  pfn.clear_indices = clear_indices_dev;
  pfn.clear_batch = clear_batch;
  return step_impl(ctx, stream_, points_dev, points_stride, n_points, batch, prm, nullptr, indices_dev,
                   num_cells_dev, emitted, &pfn);
}

extern "C" int pp_voxelize_step_kernel_name(const pp_voxel_params_t *prm, int batch, char *name, int cap) {
  if (!prm || !name || cap < 24 || batch < 1) {
    set_error("pp_voxelize_step_kernel_name: bad argument");
    return PP_ERR_VALUE;
  }
  const int N = prm->max_points_per_pillar;
  const int mode = (N % 4 == 0 && N <= 4096) ? kModeDenseVec4 : kModeDenseScalar;
  // the same rule as step_impl's (one launch per call: the development knobs PP_EMIT_SC1 / PP_STEP_SUB_MB aside)
  const bool sc1 = mode == kModeDenseVec4 &&
                   (size_t)batch * 36u * (size_t)prm->max_pillars * (size_t)N <= kSc1MaxBytes;
  std::snprintf(name, (size_t)cap, "pp::k_step<%d, %d>", mode, sc1 ? kAuxSc1 : kAuxPlain);
  return PP_OK;
}

extern "C" int pp_voxelize_step_reset(pp_ctx_t *ctx) {
  if (!ctx) {
    set_error("ctx is NULL");
    return PP_ERR_VALUE;
  }
  for (auto &sb : ctx->step_batch) sb.valid = false;   // nothing to undo on the device: every role reads only
  return PP_OK;                                         // what an earlier launch of ITS batch wrote
}

static int voxelize_pfn_impl(pp_ctx_t *ctx, void *stream_, const float *points_dev,
                             int64_t points_stride, const int32_t *n_points, int batch,
                             const pp_voxel_params_t *prm, const float *pfn_params_dev,
                             int channels, float *features_dev, int64_t *indices_dev,
                             int32_t *num_cells_dev, float *canvas_dev, int canvas_h, int canvas_w,
                             int channels_last, const int64_t *prev_indices_dev = nullptr) {
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
    if (prev_indices_dev && (reinterpret_cast<uintptr_t>(prev_indices_dev) & 7)) {
      set_error("prev_indices_dev must be 8-byte aligned");
      return PP_ERR_VALUE;
    }
    if (!prev_indices_dev)   // an unknown canvas: all of it
      PP_HIP_TRY(hipMemsetAsync(canvas_dev, 0,
                                (size_t)batch * kPfnChannels * canvas_h * canvas_w * sizeof(float),
                                stream));
  }
  rc = launch_pipeline<float>(ctx, stream, points_dev, points_stride, 4, 1, 1, np, batch, maxn,
                              g, P, N, l, kModePfn, nullptr,
                              reinterpret_cast<long long *>(indices_dev), nullptr, true,
                              pfn_params_dev, features_dev, canvas_dev, canvas_h, canvas_w,
                              channels_last ? 1 : 0,
                              canvas_dev ? reinterpret_cast<const long long *>(prev_indices_dev) : nullptr);
  if (rc) return rc;
  if (num_cells_dev) {
    char *ws = static_cast<char *>(ctx->vox_ws[0].ptr);
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

extern "C" int pp_voxelize_pfn_canvas_reuse_dev(pp_ctx_t *ctx, void *stream_, const float *points_dev,
                                                int64_t points_stride, const int32_t *n_points,
                                                int batch, const pp_voxel_params_t *prm,
                                                const float *pfn_params_dev, int channels,
                                                float *canvas_dev, int canvas_h, int canvas_w,
                                                int channels_last, int64_t *indices_dev,
                                                int32_t *num_cells_dev, const int64_t *prev_indices_dev) {
  if (!canvas_dev) {
    set_error("pp_voxelize_pfn_canvas_reuse_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  return voxelize_pfn_impl(ctx, stream_, points_dev, points_stride, n_points, batch, prm,
                           pfn_params_dev, channels, nullptr, indices_dev, num_cells_dev,
                           canvas_dev, canvas_h, canvas_w, channels_last, prev_indices_dev);
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
  static const bool trace = getenv("PP_DROPIN_TRACE") != nullptr;  // development knob: where a call's time goes
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t_prev = now();
  double t_sec[6] = {0, 0, 0, 0, 0, 0};
  auto lap = [&](int k) {
    if (!trace) return;
    const auto t = now();
    t_sec[k] += std::chrono::duration<double, std::micro>(t - t_prev).count();
    t_prev = t;
  };
  const int n = (int)n_points;
  const int max_pillars = std::max(prm->max_pillars, 0);
  const int N = std::max(prm->max_points_per_pillar, 0);
  // there are at most min(n, ncells) non-empty cells
  const int P = std::max(1, std::min(max_pillars, std::min(n, g.ncells)));
  DeviceGuard guard(ctx->device);
  hipStream_t stream = nullptr;
  // gather the (arbitrarily strided) points into pinned staging
  rc = ctx->pin_in.ensure((size_t)n * 32);
  if (rc) return rc;
  HostPool *pool = host_pool(ctx);
  // development knobs of the host path (defaults: what tools/lab/dropin_trace.py measured best -- profiles/r06/NOTES.md:
  // 8 threads, one host-to-device copy, the features back in two chunks, polled waits, f32 transport of f32-valued
  // clouds: 0.36 -> 0.19 ms per call; sending the features AHEAD of the descriptors, PP_DROPIN_SPEC=1, measured no gain)
  static const int k_h2d_parts = [] { const char *e = getenv("PP_DROPIN_H2D_PARTS"); return e ? std::max(1, std::min(8, atoi(e))) : 1; }();
  static const int k_chunks = [] { const char *e = getenv("PP_DROPIN_CHUNKS"); return e ? std::max(1, std::min(8, atoi(e))) : 2; }();
  static const bool k_spec = [] { const char *e = getenv("PP_DROPIN_SPEC"); return e ? atoi(e) != 0 : false; }();
  static const bool k_spin = [] { const char *e = getenv("PP_DROPIN_SPIN"); return e ? atoi(e) != 0 : true; }();
  // A call waits for the device two or three times, for tens of microseconds each: polled, not slept on (an interrupt-driven
  // wake-up costs about as much as the wait itself).
  auto wait_event = [&](hipEvent_t ev) -> hipError_t {
    if (!k_spin) return hipEventSynchronize(ev);
    for (;;) {
      const hipError_t e = hipEventQuery(ev);
      if (e != hipErrorNotReady) return e;
      __builtin_ia32_pause();
    }
  };
  constexpr int kMaxEv = (int)(sizeof ctx->chunk_ev / sizeof ctx->chunk_ev[0]);
  for (int c = 0; c < kMaxEv; ++c)
    if (!ctx->chunk_ev[c]) PP_HIP_TRY(hipEventCreateWithFlags(&ctx->chunk_ev[c], hipEventDisableTiming));
  // The reference's caller hands over float64 points that ARE float32 values (the lidar files are f32, the SDK keeps the
  // transformed points in an f32 array, np.hstack widens them: data/dataset.py:51-88) -- and a point cloud whose every
  // value survives double -> float -> double goes through the f32-input kernels, which widen it again on the device:
  // the same bits in, half the bytes over PCIe (0.96 instead of 1.92 MB at BASELINE config 2's 60 000 points).  The
  // test rides on the gather; one value that does not survive (or a first gather that found one) and the call takes
  // the f64-input kernels as before.  (NaN compares unequal to itself and counts as surviving: both kernel families
  // drop the point.)
  static const bool k_try_f32 = [] { const char *e = getenv("PP_DROPIN_F32"); return e ? atoi(e) != 0 : true; }();
  // ... and on the way back such a cloud's x, y, z, intensity columns travel as the floats they are (56 bytes a point
  // instead of 72; the scatter widens them): store_compact_features.  PP_DROPIN_PACK=0: nine doubles as before.
  static const bool k_pack = [] { const char *e = getenv("PP_DROPIN_PACK"); return e ? atoi(e) != 0 : true; }();
  const char *src_pts = static_cast<const char *>(points);
  bool as_f32 = k_try_f32;
  if (as_f32) {
    float *dst = static_cast<float *>(ctx->pin_in.ptr);
    std::atomic<int> lossy{0};
    pool->run([&](int part, int parts) {
      const int64_t i0 = (int64_t)n * part / parts, i1 = (int64_t)n * (part + 1) / parts;
      int bad = 0;
      for (int64_t i = i0; i < i1; ++i)
        for (int c = 0; c < 4; ++c) {
          double v;
          std::memcpy(&v, src_pts + i * ps0 + c * ps1, 8);
          const float f = (float)v;
          bad |= ((double)f != v) & (v == v);
          dst[i * 4 + c] = f;
        }
      if (bad) lossy.store(1, std::memory_order_relaxed);
    });
    as_f32 = lossy.load() == 0;
  }
  VoxLayout l;
  rc = prepare_ws(ctx, stream, 1, n, g, P, as_f32 ? 16 : 32, &l);
  if (rc) return rc;
  rc = ctx->stage_in.ensure((size_t)l.ncap * 32);
  if (rc) return rc;
  rc = ctx->stage_out.ensure((size_t)l.ncap * 72);
  if (rc) return rc;
  // Two small transfers need no copy engine at all (PP_DROPIN_DIRECT, bit mask; f32-valued clouds): 1 -- k_split reads
  // the gathered points straight from the pinned staging (each point is read once, coalesced: the kernel runs at the
  // link's rate and the copy's start-up and completion hand-over fall away); 2 -- k_emit writes the descriptors and the
  // totals, which nothing on the device reads back, straight into pinned memory: they are there when the kernel is.
  // 4 -- the features' way back is k_records_to_host, enqueued behind the kernels and sized on the device.
  static const int k_direct = [] { const char *e = getenv("PP_DROPIN_DIRECT"); return e ? atoi(e) : 7; }();
  const size_t meta_bytes = (l.totals - l.meta) + 8;
  rc = ctx->pin_meta.ensure(meta_bytes);
  if (rc) return rc;
  char *pm = static_cast<char *>(ctx->pin_meta.ptr);
  void *pm_dev = nullptr;
  const bool meta_direct = (k_direct & 2) && hipHostGetDevicePointer(&pm_dev, pm, 0) == hipSuccess && pm_dev;
  (void)hipGetLastError();  // (a refused mapping is not an error of this call: the copies are used)
  int4 *meta_host = meta_direct ? reinterpret_cast<int4 *>(pm_dev) : nullptr;
  int2 *totals_host = meta_direct ? reinterpret_cast<int2 *>(static_cast<char *>(pm_dev) + (l.totals - l.meta)) : nullptr;
  NPoints np;
  std::memset(&np, 0, sizeof np);
  np.n[0] = n;
  if (as_f32) {
    lap(0);
    const float *pts_dev = static_cast<const float *>(ctx->stage_in.ptr);
    void *mapped = nullptr;
    if ((k_direct & 1) && hipHostGetDevicePointer(&mapped, ctx->pin_in.ptr, 0) == hipSuccess && mapped)
      pts_dev = static_cast<const float *>(mapped);
    else
      PP_HIP_TRY(hipMemcpyAsync(ctx->stage_in.ptr, ctx->pin_in.ptr, (size_t)n * 16, hipMemcpyHostToDevice, stream));
    (void)hipGetLastError();
    rc = launch_pipeline<float>(ctx, stream, pts_dev, l.ncap, 4, 1, 1, np, 1, n, g, P,
                                N, l, kModeCompact, nullptr, nullptr, static_cast<double *>(ctx->stage_out.ptr), false,
                                nullptr, nullptr, nullptr, 0, 0, 0, nullptr, k_pack ? 1 : 0, meta_host, totals_host);
    if (rc) return rc;
  } else {
    // the caller's points are a strided view (data/dataset.py:88 passes the transpose of a [4, n] array): rows split
    // across the pool's threads; a part's host-to-device copy runs under the next part's gather
    double *dst = static_cast<double *>(ctx->pin_in.ptr);
    const int parts_h2d = n >= 8192 ? k_h2d_parts : 1;
    for (int h = 0; h < parts_h2d; ++h) {
      const int64_t r0 = (int64_t)n * h / parts_h2d, r1 = (int64_t)n * (h + 1) / parts_h2d;
      pool->run([&](int part, int parts) {
        const int64_t i0 = r0 + (r1 - r0) * part / parts, i1 = r0 + (r1 - r0) * (part + 1) / parts;
        for (int64_t i = i0; i < i1; ++i)
          for (int c = 0; c < 4; ++c)
            std::memcpy(&dst[i * 4 + c], src_pts + i * ps0 + c * ps1, 8);
      });
      if (r1 > r0)
        PP_HIP_TRY(hipMemcpyAsync(static_cast<char *>(ctx->stage_in.ptr) + r0 * 32, dst + r0 * 4, (size_t)(r1 - r0) * 32,
                                  hipMemcpyHostToDevice, stream));
    }
    lap(0);
    rc = launch_pipeline<double>(ctx, stream, static_cast<const double *>(ctx->stage_in.ptr),
                                 l.ncap, 4, 1, 1, np, 1, n, g, P, N, l, kModeCompact, nullptr,
                                 nullptr, static_cast<double *>(ctx->stage_out.ptr), false,
                                 nullptr, nullptr, nullptr, 0, 0, 0, nullptr, 0, meta_host, totals_host);
    if (rc) return rc;
  }
  // descriptors back in ONE copy: pillar_meta[P] (pillar order, written by k_emit in compact mode) and, behind it in the
  // workspace, the totals
  rc = ctx->pin_out.ensure((size_t)l.ncap * 72);
  if (rc) return rc;
  char *ws = static_cast<char *>(ctx->vox_ws[0].ptr);
  if (!meta_direct) PP_HIP_TRY(hipMemcpyAsync(pm, ws + l.meta, meta_bytes, hipMemcpyDeviceToHost, stream));
  hipEvent_t ev_meta = ctx->chunk_ev[kMaxEv - 1];
  PP_HIP_TRY(hipEventRecord(ev_meta, stream));
  // The features follow WITHOUT waiting for the descriptors: how many points the call emits is only known from them, so
  // the copy is sized by what the previous call of this shape emitted (consecutive sweeps are alike) plus an eighth,
  // and topped up below when that was short.  In chunks: chunk k is scattered while chunk k + 1 is on its way.
  const bool packed = as_f32 && k_pack;
  const int64_t rec = packed ? kPackedFeat : 72;  // bytes of a point's features on the way back
  int64_t cb[kMaxEv + 1];  // chunk c = compact points [cb[c], cb[c + 1])
  int nch = 0;
  cb[0] = 0;
  // (PP_DROPIN_DIRECT & 4) the pieces go out right here, behind the kernels, sized on the device by k_records_to_host
  int direct_pieces = 0;
  {
    void *out_dev = nullptr;
    if ((k_direct & 4) && N > 0 && hipHostGetDevicePointer(&out_dev, ctx->pin_out.ptr, 0) == hipSuccess && out_dev) {
      direct_pieces = (int64_t)n * rec >= (256 << 10) ? k_chunks : 1;
      const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(256, ((int64_t)n * rec / direct_pieces + 16 * kCopyThreads - 1) /
                                                                               (16 * kCopyThreads)));
      for (int c = 0; c < direct_pieces; ++c) {
        hipLaunchKernelGGL(k_records_to_host, dim3((unsigned)blocks), dim3(kCopyThreads), 0, stream,
                           reinterpret_cast<const int2 *>(ws + l.totals), reinterpret_cast<const int4 *>(ws + l.meta), P,
                           max_pillars, (int)rec, static_cast<const char *>(ctx->stage_out.ptr), static_cast<char *>(out_dev),
                           c, direct_pieces);
        PP_HIP_TRY(hipGetLastError());
        PP_HIP_TRY(hipEventRecord(ctx->chunk_ev[c], stream));
      }
    }
    (void)hipGetLastError();
  }
  auto issue_chunks = [&](int64_t upto, int pieces) -> int {
    const int64_t from = cb[nch];
    for (int c = 0; c < pieces && nch < kMaxEv - 2; ++c) {
      const int64_t b1 = from + (upto - from) * (c + 1) / pieces;
      if (b1 <= cb[nch]) continue;
      PP_HIP_TRY(hipMemcpyAsync(static_cast<char *>(ctx->pin_out.ptr) + cb[nch] * rec,
                                static_cast<const char *>(ctx->stage_out.ptr) + cb[nch] * rec, (size_t)(b1 - cb[nch]) * rec,
                                hipMemcpyDeviceToHost, stream));
      PP_HIP_TRY(hipEventRecord(ctx->chunk_ev[nch], stream));
      cb[++nch] = b1;
    }
    return PP_OK;
  };
  if (!direct_pieces && k_spec && N > 0 && ctx->dropin_last_n == n && ctx->dropin_last_end > 0) {
    const int64_t guess = std::min<int64_t>(n, ctx->dropin_last_end + ctx->dropin_last_end / 8);
    rc = issue_chunks(guess, guess * rec >= (256 << 10) ? k_chunks : 1);
    if (rc) return rc;
  }
  PP_HIP_TRY(wait_event(ev_meta));
  lap(1);
  int tot[2];
  std::memcpy(tot, pm + (l.totals - l.meta), 8);
  if (num_cells) *num_cells = tot[0];
  const int npil = std::min(tot[0], std::min(P, max_pillars));
  if (npil == 0) {
    if (nch || direct_pieces) PP_HIP_TRY(hipStreamSynchronize(stream));  // copies sent ahead still target the pinned buffer
    return PP_OK;
  }
  const int4 *meta = reinterpret_cast<const int4 *>(pm);
  const int64_t end = (int64_t)meta[npil - 1].y + meta[npil - 1].z;
  ctx->dropin_last_n = n;
  ctx->dropin_last_end = end;
  const char *feat = static_cast<const char *>(ctx->pin_out.ptr);
  // `live` points' features from compact position `at` into a dense [live][9] run of doubles
  auto copy_rows = [&](char *dst, int64_t at, int live) {
    if (!packed) {
      std::memcpy(dst, feat + at * 72, (size_t)live * 72);
      return;
    }
    const char *src = feat + at * kPackedFeat;
    for (int k = 0; k < live; ++k, src += kPackedFeat, dst += 72) {
      float head[4];
      std::memcpy(head, src, 16);
      const double wide[4] = {(double)head[0], (double)head[1], (double)head[2], (double)head[3]};
      std::memcpy(dst, wide, 32);
      std::memcpy(dst + 32, src + 16, 40);
    }
  };
  char *tp = static_cast<char *>(tensor);
  char *ip = static_cast<char *>(indices);
  auto write_index_row = [&](int p) {  // pillars.cpp:389-391
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
  };
  // The usual caller hands in a C-contiguous [P,N,9] tensor (np.zeros, data/dataset.py:89): a pillar's live points are
  // then ONE contiguous run of live * 72 bytes -- but 7200 bytes from the next pillar's, in an 86 MB array that no CPU
  // cache holds: the scatter is a chain of 12 000 cache misses (195 us of round 4's 360 us call on one thread; 150-178
  // with the next pillars' lines prefetched for writing).  Nothing can go out of range in that layout once the shapes
  // hold the call's pillars, so the rows -- disjoint -- are split across the pool's threads.
  const bool dense_t = tensor && t_strides[2] == 8 && t_strides[1] == 72 && t_shape[2] >= 9 && t_shape[1] >= N &&
                       t_shape[0] >= npil;
  const bool dense_all = dense_t && indices && i_shape[0] >= npil && i_shape[1] >= 3;
  if (direct_pieces) {  // the device cut [0, end) at the same places
    nch = direct_pieces;
    for (int c = 0; c <= nch; ++c) cb[c] = piece_begin(end, c, nch);
  } else if (N > 0 && cb[nch] < end) {  // nothing sent ahead, or the guess was short: the rest
    rc = issue_chunks(end, (dense_all && (end - cb[nch]) * rec >= (256 << 10)) ? k_chunks : 1);
    if (rc) return rc;
    if (cb[nch] < end) {  // (out of events: cannot happen with k_chunks <= 8 / 2)
      set_error("create_pillars: internal: chunk events exhausted");
      return PP_ERR_INTERNAL;
    }
  }
  if (dense_all) {
    int p_done = 0;
    for (int c = 0; c < std::max(nch, 1); ++c) {
      if (nch) PP_HIP_TRY(wait_event(ctx->chunk_ev[c]));
      // the pillars whose points have all arrived: start + count <= cb[c + 1]
      int p_hi = npil;
      if (nch && cb[c + 1] < end) {
        int lo = p_done, hi = npil;
        while (lo < hi) {
          const int mid = (lo + hi) / 2;
          if ((int64_t)meta[mid].y + meta[mid].z <= cb[c + 1]) lo = mid + 1; else hi = mid;
        }
        p_hi = lo;
      } else if (c + 1 < nch) {
        continue;  // everything needed is here already; the later events are waited for at the end
      }
      const int p0 = p_done, p1 = p_hi;
      p_done = p_hi;
      if (p1 <= p0) continue;
      pool->run([&](int part, int parts) {
        const int q0 = p0 + (int)((int64_t)(p1 - p0) * part / parts), q1 = p0 + (int)((int64_t)(p1 - p0) * (part + 1) / parts);
        constexpr int kAhead = 8;
        for (int p = q0; p < q1; ++p) {
          if (p + kAhead < q1) {
            const char *nx = tp + (int64_t)(p + kAhead) * t_strides[0];
            const int nl = std::min(meta[p + kAhead].z, N) * 72;
            for (int o = 0; o < nl; o += 64) __builtin_prefetch(nx + o, 1, 0);
            if (nl > 0) __builtin_prefetch(nx + nl - 1, 1, 0);
          }
          const int live = std::min(meta[p].z, N);
          if (live > 0) copy_rows(tp + (int64_t)p * t_strides[0], meta[p].y, live);
          write_index_row(p);
        }
      });
    }
    if (nch) PP_HIP_TRY(wait_event(ctx->chunk_ev[nch - 1]));  // (a copy sent ahead may reach beyond `end`)
    lap(2);
    if (trace)
      fprintf(stderr, "pp_create_pillars_f64: gather + H2D issue (%d parts) %.0f us | kernels + descriptors back %.0f | features back in %d chunks + scatter on %d threads %.0f\n",
              n >= 8192 ? k_h2d_parts : 1, t_sec[0], t_sec[1], nch, pool->size(), t_sec[2]);
    return PP_OK;
  }
  if (nch) PP_HIP_TRY(hipStreamSynchronize(stream));
  lap(2);
  // any other layout (strided views, undersized arrays): one thread, element by element with pybind11 .mutable_at()'s
  // bounds checks, pillar by pillar like pillars.cpp:335-396 -- the writes made before an IndexError persist, nothing
  // behind it is touched
  for (int p = 0; p < npil; ++p) {
    const int live = std::min(meta[p].z, N);
    if (dense_t) {
      if (live > 0) copy_rows(tp + (int64_t)p * t_strides[0], meta[p].y, live);
    } else
    for (int k = 0; k < live; ++k) {
      double f[9];
      copy_rows(reinterpret_cast<char *>(f), (int64_t)meta[p].y + k, 1);
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
    write_index_row(p);
  }
  lap(3);
  if (trace)
    fprintf(stderr, "pp_create_pillars_f64: gather %.0f us | H2D + kernels + descriptors back %.0f | features back %.0f | scatter %.0f\n",
            t_sec[0], t_sec[1], t_sec[2], t_sec[3]);
  return PP_OK;
}
