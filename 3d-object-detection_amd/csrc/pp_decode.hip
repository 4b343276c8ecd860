// pp_decode.hip -- inference post-processing on the device.
//
// Replaces the per-sample tail of evaluate() / evaluate_single()
// (/root/reference evaluate.py:231-245 and :146-158): sigmoid over the class
// logits, tanh on regression element 6, max/argmax over classes, score threshold
// (cfg.DATA.VAL_POS_THRESH), axis-aligned greedy NMS over the ANCHOR rectangles
// (box_nms, evaluate.py:127-139, torchvision.ops.nms), the first 100 survivors,
// make_pred_boxes (:33-89) and move_box_to_car_space (:91-125).
//
// torchvision is absent from the image; torchvision.ops.nms is restated from its
// published CPU kernel: boxes in decreasing score order, a box is dropped when its
// IoU with an already kept box exceeds the threshold, areas/intersections in f32.
//
// Kernels: k_score (one lane per anchor; the anchors above the threshold are compacted by
// wave ballots into a candidate list of unique 50-bit keys {1 - score bits, anchor}),
// k_sort_runs (runs of 16384 candidates, one workgroup each, bitonic sort in LDS), k_nms (one
// workgroup: consumes the candidates in key order -- one run directly, up to eight runs
// through an on-the-fly merge of their next 256 entries each, more than that after an
// in-place bitonic sort in global memory -- in 256-candidate chunks: kept-list test, an
// in-chunk suppression matrix resolved serially, and the box decode of the survivors).

#include <algorithm>
#include <cstring>

#include "pp_common.h"

namespace pp {

using u64 = unsigned long long;
constexpr u64 kSentinel = ~0ull;
constexpr int kNmsThreads = 1024;  // k_nms: four threads per candidate of a chunk
constexpr int kChunkN = 256;      // candidates per NMS chunk
constexpr int kMaxOut = 1024;
constexpr int kRun = 16384;      // candidates per sorted run (128 KiB of LDS in k_sort_runs)
constexpr int kSortThreads = 1024;
constexpr int kMaxRuns = 8;       // runs k_nms merges on the fly (kMaxRuns * 256 keys in LDS)

struct DecodeArgs {
  // element (channel ch, cell) of cls / reg at [ch*stride_c + cell*stride_pix]: NCHW planes
  // (stride_c = H*W, stride_pix = 1) or channels-last rows (stride_c = 1, stride_pix = row pitch)
  const float *cls;
  const float *reg;
  int64_t cls_sc, cls_sp, reg_sc, reg_sp;
  int64_t cls_sb, reg_sb;  // floats from one sample of the batch to the next (sample = blockIdx.y)
  const double *a_centers, *a_wlh, *a_yaw, *a_xy;
  int A, Ac, C, HW;
  float pos_thresh, nms_thresh;
  int max_out;
  double canvas_height, x_step, y_step, x_min, y_min;
  u64 *keys;      // candidate keys, capacity pow2 >= A (k_score appends, k_nms sorts)
  int *ncand;     // number of candidates; zeroed again by k_nms
  int cap;        // capacity of keys (power of two)
  int run;        // candidates per sorted run = min(kRun, cap)
  int *kept;      // [max_out] anchor ids in keep order
  int *count;     // number kept
  double *boxes;  // [max_out][9] x,y,z,w,l,h,yaw,score,class
};

// the arguments of sample b = blockIdx.y of the batch: its slices of the inputs, its keys and
// counter, its output rows
__device__ __forceinline__ DecodeArgs sample_view(DecodeArgs d) {
  const int64_t b = blockIdx.y;
  d.cls += b * d.cls_sb;
  d.reg += b * d.reg_sb;
  d.keys += b * (int64_t)d.cap;
  d.ncand += b;
  d.kept += b * d.max_out;
  d.count += b;
  d.boxes += b * (int64_t)d.max_out * 9;
  return d;
}

__device__ __forceinline__ float sigmoidf_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

// scores,classes = torch.max(torch.sigmoid(cls), dim=-1)   (evaluate.py:231-235)
__device__ __forceinline__ void anchor_score(const DecodeArgs &d, int a, float &score, int &klass) {
  const int cell = a / d.Ac, k = a - cell * d.Ac;
  const float *p = d.cls + (int64_t)(k * d.C) * d.cls_sc + (int64_t)cell * d.cls_sp;
  score = -1.0f;
  klass = 0;
  for (int c = 0; c < d.C; ++c) {
    const float s = sigmoidf_ref(p[(int64_t)c * d.cls_sc]);
    if (s > score) {  // first maximum wins
      score = s;
      klass = c;
    }
  }
}

__global__ __launch_bounds__(256) void k_score(DecodeArgs d_) {
  const DecodeArgs d = sample_view(d_);
  const int a = blockIdx.x * 256 + threadIdx.x;
  float s = 0.0f;
  int c;
  if (a < d.A) anchor_score(d, a, s, c);
  const bool cand = a < d.A && s > d.pos_thresh;  // evaluate.py:237 (strict >)
  // compaction: one returning atomic per wave, ranks from the ballot
  const u64 m = __ballot(cand);
  if (!m) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == 0) base = atomicAdd(d.ncand, __popcll(m));
  base = __builtin_amdgcn_readfirstlane(base);
  if (cand) {
    // decreasing score, then increasing anchor id; s in (0,1] so its bit pattern orders like s
    // (30 significant bits of 0x3F800000 - bits(s), 20 bits of anchor id: unique 50-bit keys)
    const unsigned sb = (unsigned)__float_as_int(s);
    const u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    // (the counter is zero at the start of a call, so base + rank < A <= cap; the bound only matters
    // if an earlier call died between k_score and k_nms and left the counter non-zero)
    const int pos = base + __popcll(m & lt);
    if (pos < d.cap) d.keys[pos] = ((u64)(0x3F800000u - sb) << 20) | (u64)a;
  }
}

// ascending bitonic sort of buf[0..n2) (n2 a power of two) by one workgroup of T threads;
// LDS or global memory of that workgroup (same CU: visible behind the barrier)
template <int T>
__device__ __forceinline__ void bitonic_sort(u64 *buf, int n2, int t) {
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < (n2 >> 1); i += T) {
        const int lo = 2 * i - (i & (j - 1));  // element index with bit j clear
        const int hi = lo + j;
        const u64 x = buf[lo], y = buf[hi];
        if ((x > y) == ((lo & k) == 0)) {
          buf[lo] = y;
          buf[hi] = x;
        }
      }
      __syncthreads();
    }
}

// run r = candidates [r*run, (r+1)*run): sorted in LDS, written back padded with sentinels
__global__ __launch_bounds__(kSortThreads) void k_sort_runs(DecodeArgs d_, int run) {
  const DecodeArgs d = sample_view(d_);
  extern __shared__ __attribute__((aligned(16))) u64 s_run[];
  const int M = min(*d.ncand, d.cap);
  const int base = blockIdx.x * run;
  if (base >= M || M > kMaxRuns * run) return;  // nothing here / too many runs: k_nms sorts in place
  const int n = min(run, M - base);
  int n2 = 1;
  while (n2 < n) n2 <<= 1;
  const int t = threadIdx.x;
  for (int i = t; i < n2; i += kSortThreads) s_run[i] = i < n ? d.keys[base + i] : kSentinel;
  __syncthreads();
  bitonic_sort<kSortThreads>(s_run, n2, t);
  for (int i = t; i < run; i += kSortThreads) d.keys[base + i] = i < n2 ? s_run[i] : kSentinel;
}

struct NmsBox {
  float x1, y1, x2, y2, area;
};

// box_nms, evaluate.py:127-139: anchor_xy as float32, rows 1 and 3 flipped with (H-1) - y
__device__ __forceinline__ NmsBox load_box(const DecodeArgs &d, int a) {
  NmsBox b;
  const float h1 = (float)(d.canvas_height - 1);
  b.x1 = (float)d.a_xy[(int64_t)a * 4 + 0];
  b.y1 = h1 - (float)d.a_xy[(int64_t)a * 4 + 1];
  b.x2 = (float)d.a_xy[(int64_t)a * 4 + 2];
  b.y2 = h1 - (float)d.a_xy[(int64_t)a * 4 + 3];
  b.area = (b.x2 - b.x1) * (b.y2 - b.y1);
  return b;
}

// torchvision nms_kernel (CPU): ovr = inter / (iarea + areas[j] - inter), suppressed if ovr > thresh
__device__ __forceinline__ bool suppresses(const NmsBox &i, const NmsBox &j, float thresh) {
  const float xx1 = fmaxf(i.x1, j.x1), yy1 = fmaxf(i.y1, j.y1);
  const float xx2 = fminf(i.x2, j.x2), yy2 = fminf(i.y2, j.y2);
  const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
  const float inter = w * h;
  const float ovr = inter / (i.area + j.area - inter);
  return ovr > thresh;
}

// make_pred_boxes (evaluate.py:33-89) + move_box_to_car_space (:91-125, image=True) for output
// row i; a = kept anchor id or -1
__device__ void decode_row(const DecodeArgs &d, int i, int a) {
  double *o = d.boxes + (int64_t)i * 9;
  if (a < 0) {
    for (int k = 0; k < 9; ++k) o[k] = 0.0;
    return;
  }
  float score;
  int klass;
  anchor_score(d, a, score, klass);
  const int cell = a / d.Ac, k = a - cell * d.Ac;
  float off[8];
  for (int r = 0; r < 8; ++r) off[r] = d.reg[(int64_t)(k * 8 + r) * d.reg_sc + (int64_t)cell * d.reg_sp];
  off[6] = tanhf(off[6]);  // evaluate.py:234
  const double ax = d.a_centers[(int64_t)a * 3], ay = d.a_centers[(int64_t)a * 3 + 1],
               az = d.a_centers[(int64_t)a * 3 + 2];
  const double aw = d.a_wlh[(int64_t)a * 3], al = d.a_wlh[(int64_t)a * 3 + 1],
               ah = d.a_wlh[(int64_t)a * 3 + 2];
  const double diag = sqrt(aw * aw + al * al);
  const double bx = ax + (double)off[0] * diag;
  const double by = ay + (double)off[1] * diag;
  const double bz = az + (double)off[2] * ah;
  const double bw = (double)expf(off[3]) * aw;  // np.exp of a float32 scalar is float32
  const double bl = (double)expf(off[4]) * al;
  const double bh = (double)expf(off[5]) * ah;
  const double yaw = (double)asinf(off[6]) + d.a_yaw[a];
  const double y = (d.canvas_height - 1) - by;
  o[0] = bx * d.x_step + d.x_min;
  o[1] = y * d.y_step + d.y_min;
  o[2] = bz;
  o[3] = bw * d.y_step;
  o[4] = bl * d.x_step;
  o[5] = bh;
  o[6] = yaw;
  o[7] = (double)score;
  o[8] = (double)klass;
}

#ifdef PP_NMS_STAMPS
__device__ unsigned long long g_nms_stamps[64 * 8];
#define NMS_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.y == 0 && (c0 >> 8) < 64) g_nms_stamps[(c0 >> 8) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define NMS_STAMP(k) do {} while (0)
#endif

// One workgroup per sample; chunks of kChunkN = 256 candidates in key order, four threads per candidate.
__global__ __launch_bounds__(kNmsThreads) void k_nms(DecodeArgs d_) {
  const DecodeArgs d = sample_view(d_);
  __shared__ NmsBox s_kept[kMaxOut];
  __shared__ NmsBox s_chunk[kChunkN];
  __shared__ u64 s_mask[kChunkN][kChunkN / 64];  // s_mask[i]: later chunk members i suppresses
  __shared__ int s_alive[kChunkN];
  __shared__ NmsBox s_cbox[kChunkN];             // the chunk's survivors of the kept-list test, packed
  __shared__ int s_cid[kChunkN];
  __shared__ int s_acnt[kChunkN / 64];
  __shared__ int s_id[kChunkN];
  __shared__ int s_keptid[kMaxOut];
  __shared__ int s_nkept, s_done, s_full;
  __shared__ u64 s_merge[kMaxRuns * kChunkN];
  __shared__ int s_head[kMaxRuns];
  const int t = threadIdx.x;
  const int c = t >> 2, sub = t & 3;             // candidate of the chunk, quarter of its work
  if (t == 0) {
    s_nkept = 0;
    s_done = 0;
  }
  if (t < kMaxRuns) s_head[t] = 0;
  for (int i = t; i < d.max_out; i += kNmsThreads) s_keptid[i] = -1;
  // the candidates in increasing key order = decreasing score (torchvision nms's order)
  const int M = min(*d.ncand, d.cap);
  const int run = d.run;
  const int nruns = (M + run - 1) / run;
  __syncthreads();
  if (t == 0) *d.ncand = 0;  // armed for the next call
  if (nruns > kMaxRuns) {    // k_sort_runs left it alone: one in-place sort of everything
    int n2 = 1;
    while (n2 < M) n2 <<= 1;
    for (int i = M + t; i < n2; i += kNmsThreads) d.keys[i] = kSentinel;
    __syncthreads();
    bitonic_sort<kNmsThreads>(d.keys, n2, t);
  }
  const bool merging = nruns > 1 && nruns <= kMaxRuns;
  int m2 = 1;
  while (m2 < nruns * kChunkN) m2 <<= 1;
  for (int c0 = 0; c0 < M; c0 += kChunkN) {
    const int nk = s_nkept;
    NMS_STAMP(0);
    u64 key;
    if (merging) {
      // the next 256 keys overall are among the next 256 of every run: sort those (run id
      // in the low 4 bits; keys are unique, so the order of the keys proper is unchanged)
      for (int i = t; i < m2; i += kNmsThreads) {
        const int r = i / kChunkN, idx = s_head[min(r, kMaxRuns - 1)] + (i % kChunkN);
        u64 k0 = kSentinel;
        if (r < nruns && idx < run) k0 = d.keys[(size_t)r * run + idx];
        s_merge[i] = k0 == kSentinel ? kSentinel : (k0 << 4) | (u64)r;
      }
      __syncthreads();
      bitonic_sort<kNmsThreads>(s_merge, m2, t);
      const u64 k1 = s_merge[c];
      key = k1 == kSentinel ? kSentinel : k1 >> 4;
      if (sub == 0 && k1 != kSentinel) atomicAdd(&s_head[(int)(k1 & 15ull)], 1);
    } else {
      key = (c0 + c < M) ? d.keys[c0 + c] : kSentinel;
    }
    NMS_STAMP(1);
    const bool valid = key != kSentinel;
    const int a = (int)(key & 0xFFFFFull);
    NmsBox b = {0, 0, 0, 0, 0};
    int alive = 0;
    if (valid) {
      b = load_box(d, a);
      alive = 1;
      // this thread's quarter of the kept list, four boxes per trip: the LDS reads of a trip do
      // not wait for each other (one at a time the test was a 100 ns round trip per kept box)
      for (int k0 = sub; k0 < nk; k0 += 16) {
        bool hit = false;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = k0 + 4 * u;
          const NmsBox kb = s_kept[min(k, nk - 1)];
          hit = hit || (k < nk && suppresses(kb, b, d.nms_thresh));
        }
        if (hit) {
          alive = 0;
          break;
        }
      }
    }
    {
      // a candidate is alive when all four quarters say so (the four threads are adjacent lanes)
      const u64 bal = __ballot(alive != 0);
      const int lane = t & 63;
      alive = ((bal >> (lane & ~3)) & 0xFull) == 0xFull;
      if (sub == 0) {
        s_chunk[c] = b;
        s_alive[c] = alive;
        s_id[c] = a;
      }
    }
    NMS_STAMP(2);
    s_mask[c][sub] = 0ull;
    if (t == kNmsThreads - 1) s_full = valid;       // the chunk's last candidate exists
    __syncthreads();
    NMS_STAMP(3);
    // The members the kept list left alive, packed to the front (score order kept): the matrix and the
    // greedy pass then cost what the survivors cost -- 30 to 100 of 256 in all chunks but the first.
    int my_pos = -1;
    if (t < kChunkN) {
      const u64 bal = __ballot(s_alive[t] != 0);
      if ((t & 63) == 0) s_acnt[t >> 6] = __popcll(bal);
      if (s_alive[t]) my_pos = __popcll(bal & ((1ull << (t & 63)) - 1ull));
    }
    __syncthreads();
    const int na = s_acnt[0] + s_acnt[1] + s_acnt[2] + s_acnt[3];
    if (my_pos >= 0) {
      for (int w = 0; w < (t >> 6); ++w) my_pos += s_acnt[w];
      s_cbox[my_pos] = s_chunk[t];
      s_cid[my_pos] = s_id[t];
    }
    __syncthreads();
    // Suppression matrix among the survivors: bit j of s_mask[r] = (j > r and r suppresses j).  Row r has
    // na - 1 - r entries: rows p and na - 1 - p together have na - 1, shared by eight threads, four
    // entries per trip (all their LDS reads first, then the arithmetic, then the bits).
    {
      const int p = t >> 3, u = t & 7;
      const int rA = p, rB = na - 1 - p;
      if (rA <= rB) {
        const int nA = na - 1 - p, total = (rA == rB) ? nA : na - 1;
        const NmsBox bA = s_cbox[rA], bB = s_cbox[rB];
        for (int idx0 = u; idx0 < total; idx0 += 32) {
          NmsBox bj[4];
          int j[4];
          bool in[4], first[4], sup[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int idx = idx0 + 8 * q;
            in[q] = idx < total;
            first[q] = idx < nA;
            j[q] = in[q] ? (first[q] ? p + 1 + idx : idx + 1) : 0;
            bj[q] = s_cbox[j[q]];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            NmsBox br;
            br.x1 = first[q] ? bA.x1 : bB.x1;
            br.y1 = first[q] ? bA.y1 : bB.y1;
            br.x2 = first[q] ? bA.x2 : bB.x2;
            br.y2 = first[q] ? bA.y2 : bB.y2;
            br.area = first[q] ? bA.area : bB.area;
            sup[q] = in[q] && suppresses(br, bj[q], d.nms_thresh);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (sup[q]) atomicOr(&s_mask[first[q] ? rA : rB][j[q] >> 6], 1ull << (j[q] & 63));
        }
      }
    }
    NMS_STAMP(4);
    __syncthreads();
    NMS_STAMP(5);
    if (t < 64) {
      // greedy pass in score order, by one wave: lane l holds the matrix row of member 64*w0 + l of
      // block w0; the walk over the block's alive members is a scalar loop (a row's own-block word
      // through v_readlane), the later blocks' removed words an OR over the kept lanes.
      int n = s_nkept;
      u64 removed[kChunkN / 64] = {0, 0, 0, 0};
#pragma unroll
      for (int w0 = 0; w0 < kChunkN / 64; ++w0) {
        const int row = w0 * 64 + t;
        u64 mr[kChunkN / 64];
#pragma unroll
        for (int w = 0; w < kChunkN / 64; ++w) mr[w] = s_mask[row][w];
        const int nw = na - w0 * 64;  // survivors in this block
        const u64 am = (nw >= 64 ? ~0ull : nw > 0 ? (1ull << nw) - 1ull : 0ull) & ~removed[w0];
        u64 pend = ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(am >> 32)) << 32) |
                   (unsigned)__builtin_amdgcn_readfirstlane((int)(am & 0xFFFFFFFFull));
        u64 keptbits = 0;
        int room = d.max_out - n;
        while (pend && room > 0) {
          const int i = __builtin_ctzll(pend);
          keptbits |= 1ull << i;
          --room;
          const u64 r0 = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(mr[w0] >> 32), i) << 32) |
                         (unsigned)__builtin_amdgcn_readlane((int)(mr[w0] & 0xFFFFFFFFull), i);
          pend &= ~(r0 | (1ull << i));
        }
        const bool mine = (keptbits >> t) & 1ull;
        if (mine) {
          const int pos = n + __popcll(keptbits & ((1ull << t) - 1ull));
          s_kept[pos] = s_cbox[row];
          s_keptid[pos] = s_cid[row];
        }
        n += __popcll(keptbits);
#pragma unroll
        for (int w = w0 + 1; w < kChunkN / 64; ++w) {
          const unsigned lo = wave_or_u32(mine ? (unsigned)mr[w] : 0u), hi = wave_or_u32(mine ? (unsigned)(mr[w] >> 32) : 0u);
          removed[w] |= ((u64)hi << 32) | lo;
        }
      }
      if (t == 0) {
        s_nkept = n;
        s_done = (n >= d.max_out) ? 1 : 0;
      }
    }
    NMS_STAMP(6);
    __syncthreads();
    NMS_STAMP(7);
    // enough boxes, or the candidates ran out inside this chunk.  Both flags go to registers and a
    // barrier follows: the next iteration's `s_full = valid` (thread 1023, before that iteration's
    // first barrier when nothing is merged) must not reach a wave that has not read this one yet --
    // it would leave the loop alone and the others would wait for it at the next barrier.
    const bool stop = s_done || !s_full;
    __syncthreads();
    if (stop) break;
  }
  __syncthreads();
  if (t == 0) *d.count = s_nkept;
  // the kept list and the decoded boxes (one thread per output row)
  for (int i = t; i < d.max_out; i += kNmsThreads) {
    const int a = s_keptid[i];
    d.kept[i] = a;
    decode_row(d, i, a);
  }
}

}  // namespace pp

using namespace pp;

constexpr size_t kCounterBytes = 4096;  // one candidate counter per sample at the head of the scratch
constexpr int kMaxDecodeBatch = (int)(kCounterBytes / 4);

extern "C" int pp_decode_batch_dev(pp_ctx_t *ctx, void *stream_, int32_t batch, const float *cls_dev,
                                   const float *reg_dev, int64_t cls_stride_b, int64_t cls_stride_c,
                                   int64_t cls_stride_pix, int64_t reg_stride_b, int64_t reg_stride_c,
                                   int64_t reg_stride_pix, const double *a_centers, const double *a_wlh,
                                   const double *a_yaw, const double *a_xy, const pp_decode_params_t *prm,
                                   double *boxes_out, int32_t *kept_out, int32_t *count_out) {
  if (!ctx || !cls_dev || !reg_dev || !a_centers || !a_wlh || !a_yaw || !a_xy || !prm || !boxes_out ||
      !kept_out || !count_out) {
    set_error("pp_decode_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  const int64_t A = (int64_t)prm->fm_height * prm->fm_width * prm->anchors_per_cell;
  if (prm->fm_height < 1 || prm->fm_width < 1 || prm->anchors_per_cell < 1 || prm->num_classes < 1 ||
      A >= (1 << 20) || prm->max_out < 1 || prm->max_out > kMaxOut) {
    set_error("pp_decode_dev: need 1 <= anchors < 2^20 and 1 <= max_out <= %d", kMaxOut);
    return PP_ERR_VALUE;
  }
  if (batch < 1 || batch > kMaxDecodeBatch) {
    set_error("pp_decode_batch_dev: need 1 <= batch <= %d", kMaxDecodeBatch);
    return PP_ERR_VALUE;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != ctx->device) (void)hipSetDevice(ctx->device);
  struct Restore {
    int prev, dev;
    ~Restore() {
      if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
  } restore{prev, ctx->device};
  size_t cap = 1;
  while (cap < (size_t)A) cap <<= 1;
  // scratch: [0, 4096) candidate counters (zero between calls) | keys [batch][cap]
  bool grew = false;
  int rc = ctx->decode_ws.ensure(kCounterBytes + (size_t)batch * cap * 8, &grew);
  if (rc) return rc;
  char *ws = static_cast<char *>(ctx->decode_ws.ptr);
  if (grew)  // the counters start at zero; k_nms leaves them at zero
    PP_HIP_TRY(hipMemsetAsync(ws, 0, kCounterBytes, stream));
  DecodeArgs d;
  d.cls = cls_dev;
  d.reg = reg_dev;
  d.cls_sc = cls_stride_c;
  d.cls_sp = cls_stride_pix;
  d.reg_sc = reg_stride_c;
  d.reg_sp = reg_stride_pix;
  d.cls_sb = cls_stride_b;
  d.reg_sb = reg_stride_b;
  d.a_centers = a_centers;
  d.a_wlh = a_wlh;
  d.a_yaw = a_yaw;
  d.a_xy = a_xy;
  d.A = (int)A;
  d.Ac = prm->anchors_per_cell;
  d.C = prm->num_classes;
  d.HW = prm->fm_height * prm->fm_width;
  d.pos_thresh = (float)prm->pos_thresh;
  d.nms_thresh = (float)prm->nms_thresh;
  d.max_out = prm->max_out;
  d.canvas_height = prm->canvas_height;
  d.x_step = prm->x_step;
  d.y_step = prm->y_step;
  d.x_min = prm->x_min;
  d.y_min = prm->y_min;
  d.keys = reinterpret_cast<u64 *>(ws + kCounterBytes);
  d.ncand = reinterpret_cast<int *>(ws);
  d.cap = (int)cap;
  d.kept = kept_out;
  d.count = count_out;
  d.boxes = boxes_out;
  const unsigned nb = (unsigned)batch;
  // 128 KiB of dynamic LDS needs the attribute, once per device (a constant: the same value from
  // every context) -- BEFORE the first launch, so that a failure here leaves the counters untouched
  if (!ctx->sort_lds_armed) {
    PP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sort_runs),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kRun * 8));
    ctx->sort_lds_armed = true;
  }
  hipLaunchKernelGGL(k_score, dim3((unsigned)((A + 255) / 256), nb), dim3(256), 0, stream, d);
  d.run = (int)std::min<size_t>(kRun, cap);
  const unsigned nrun_wgs = (unsigned)std::min<size_t>(kMaxRuns, cap / d.run);
  hipLaunchKernelGGL(k_sort_runs, dim3(nrun_wgs, nb), dim3(kSortThreads), (size_t)d.run * 8, stream, d, d.run);
  hipLaunchKernelGGL(k_nms, dim3(1, nb), dim3(kNmsThreads), 0, stream, d);  // + box decode
  {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      // k_score may have run without the k_nms that zeroes the candidate counters again
      (void)hipMemsetAsync(ws, 0, kCounterBytes, stream);
      set_error("decode launch failed: %s", hipGetErrorString(e));
      return PP_ERR_HIP;
    }
  }
  return PP_OK;
}

extern "C" int pp_decode_strided_dev(pp_ctx_t *ctx, void *stream_, const float *cls_dev,
                                     const float *reg_dev, int64_t cls_stride_c,
                                     int64_t cls_stride_pix, int64_t reg_stride_c,
                                     int64_t reg_stride_pix, const double *a_centers, const double *a_wlh,
                                     const double *a_yaw, const double *a_xy, const pp_decode_params_t *prm,
                                     double *boxes_out, int32_t *kept_out, int32_t *count_out) {
  return pp_decode_batch_dev(ctx, stream_, 1, cls_dev, reg_dev, 0, cls_stride_c, cls_stride_pix, 0, reg_stride_c,
                             reg_stride_pix, a_centers, a_wlh, a_yaw, a_xy, prm, boxes_out, kept_out, count_out);
}

extern "C" int pp_decode_dev(pp_ctx_t *ctx, void *stream_, const float *cls_dev, const float *reg_dev,
                             const double *a_centers, const double *a_wlh, const double *a_yaw,
                             const double *a_xy, const pp_decode_params_t *prm, double *boxes_out,
                             int32_t *kept_out, int32_t *count_out) {
  if (!prm) {
    set_error("pp_decode_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  const int64_t hw = (int64_t)prm->fm_height * prm->fm_width;
  return pp_decode_strided_dev(ctx, stream_, cls_dev, reg_dev, hw, 1, hw, 1, a_centers, a_wlh, a_yaw,
                               a_xy, prm, boxes_out, kept_out, count_out);
}

#ifdef PP_NMS_STAMPS
extern "C" int pp_debug_nms_stamps(unsigned long long *out, int n_words) {
  if (hipDeviceSynchronize() != hipSuccess) return PP_ERR_HIP;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pp::g_nms_stamps), (size_t)n_words * 8) == hipSuccess ? PP_OK : PP_ERR_HIP;
}
#endif
