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
// wave ballots into a candidate list of unique 50-bit keys {1 - score bits, anchor}), k_nms
// (one workgroup: bitonic sort of the candidates -- in LDS up to 2048 of them, the usual
// case, else in place in global memory -- then 256-candidate chunks: kept-list test, an
// in-chunk suppression matrix resolved serially, and the box decode of the survivors).

#include <cstring>

#include "pp_common.h"

namespace pp {

using u64 = unsigned long long;
constexpr u64 kSentinel = ~0ull;
constexpr int kNmsThreads = 256;
constexpr int kMaxOut = 1024;
constexpr int kSortLds = 2048;  // candidates sorted in LDS; beyond that in place in global memory

struct DecodeArgs {
  // element (channel ch, cell) of cls / reg at [ch*stride_c + cell*stride_pix]: NCHW planes
  // (stride_c = H*W, stride_pix = 1) or channels-last rows (stride_c = 1, stride_pix = row pitch)
  const float *cls;
  const float *reg;
  int64_t cls_sc, cls_sp, reg_sc, reg_sp;
  const double *a_centers, *a_wlh, *a_yaw, *a_xy;
  int A, Ac, C, HW;
  float pos_thresh, nms_thresh;
  int max_out;
  double canvas_height, x_step, y_step, x_min, y_min;
  u64 *keys;      // candidate keys, capacity pow2 >= A (k_score appends, k_nms sorts)
  int *ncand;     // number of candidates; zeroed again by k_nms
  int cap;        // capacity of keys (power of two)
  int *kept;      // [max_out] anchor ids in keep order
  int *count;     // number kept
  double *boxes;  // [max_out][9] x,y,z,w,l,h,yaw,score,class
};

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

__global__ __launch_bounds__(256) void k_score(DecodeArgs d) {
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
  base = __shfl(base, 0, 64);
  if (cand) {
    // decreasing score, then increasing anchor id; s in (0,1] so its bit pattern orders like s
    // (30 significant bits of 0x3F800000 - bits(s), 20 bits of anchor id: unique 50-bit keys)
    const unsigned sb = (unsigned)__float_as_int(s);
    const u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    d.keys[base + __popcll(m & lt)] = ((u64)(0x3F800000u - sb) << 20) | (u64)a;
  }
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

__global__ __launch_bounds__(kNmsThreads) void k_nms(DecodeArgs d) {
  __shared__ NmsBox s_kept[kMaxOut];
  __shared__ NmsBox s_chunk[kNmsThreads];
  __shared__ u64 s_mask[kNmsThreads][kNmsThreads / 64];  // s_mask[i]: later chunk members i suppresses
  __shared__ int s_alive[kNmsThreads];
  __shared__ u64 s_alive_mask[kNmsThreads / 64];  // the same as wave ballots, for the serial pass
  __shared__ int s_id[kNmsThreads];
  __shared__ int s_keptid[kMaxOut];
  __shared__ int s_nkept, s_done;
  __shared__ u64 s_sort[kSortLds];
  const int t = threadIdx.x;
  if (t == 0) {
    s_nkept = 0;
    s_done = 0;
  }
  for (int i = t; i < d.max_out; i += kNmsThreads) s_keptid[i] = -1;
  // the candidates in increasing key order = decreasing score (torchvision nms's order)
  const int M = min(*d.ncand, d.cap);
  int n2 = 1;
  while (n2 < M) n2 <<= 1;
  const bool in_lds = n2 <= kSortLds;
  u64 *buf = in_lds ? s_sort : d.keys;
  for (int i = t; i < n2; i += kNmsThreads) {
    if (in_lds) s_sort[i] = i < M ? d.keys[i] : kSentinel;
    else if (i >= M) d.keys[i] = kSentinel;
  }
  __syncthreads();
  if (t == 0) *d.ncand = 0;  // armed for the next call
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < (n2 >> 1); i += kNmsThreads) {
        const int lo = 2 * i - (i & (j - 1));  // element index with bit j clear
        const int hi = lo + j;
        const u64 x = buf[lo], y = buf[hi];
        if ((x > y) == ((lo & k) == 0)) {
          buf[lo] = y;
          buf[hi] = x;
        }
      }
      __syncthreads();  // same workgroup, same CU: global writes are visible behind it too
    }
  for (int c0 = 0; c0 < M; c0 += kNmsThreads) {
    const int nk = s_nkept;
    const u64 key = (c0 + t < M) ? buf[c0 + t] : kSentinel;
    const bool valid = key != kSentinel;
    const int a = (int)(key & 0xFFFFFull);
    NmsBox b = {0, 0, 0, 0, 0};
    int alive = 0;
    if (valid) {
      b = load_box(d, a);
      alive = 1;
      for (int k = 0; k < nk && alive; ++k)
        if (suppresses(s_kept[k], b, d.nms_thresh)) alive = 0;
    }
    s_chunk[t] = b;
    s_alive[t] = alive;
    s_id[t] = a;
    {
      const u64 bal = __ballot(alive != 0);
      if ((t & 63) == 0) s_alive_mask[t >> 6] = bal;
    }
    __syncthreads();
    // suppression matrix inside the chunk: bit j of s_mask[t] = (j > t and t suppresses j)
    {
      u64 m[kNmsThreads / 64] = {0, 0, 0, 0};
      if (alive) {
        for (int j = t + 1; j < kNmsThreads; ++j)
          if (s_alive[j] && suppresses(b, s_chunk[j], d.nms_thresh)) m[j >> 6] |= 1ull << (j & 63);
      }
#pragma unroll
      for (int w = 0; w < kNmsThreads / 64; ++w) s_mask[t][w] = m[w];
    }
    __syncthreads();
    if (t == 0) {
      // greedy pass in score order over the ALIVE members only (ballot words, lowest bit first)
      u64 removed[kNmsThreads / 64] = {0, 0, 0, 0};
      int n = s_nkept;
#pragma unroll
      for (int w0 = 0; w0 < kNmsThreads / 64; ++w0) {
        u64 pend = s_alive_mask[w0];
        while (pend && n < d.max_out) {
          const int bpos = __builtin_ctzll(pend);
          pend &= pend - 1;
          if ((removed[w0] >> bpos) & 1ull) continue;
          const int i = w0 * 64 + bpos;
          s_kept[n] = s_chunk[i];
          s_keptid[n] = s_id[i];
          ++n;
#pragma unroll
          for (int w = 0; w < kNmsThreads / 64; ++w) removed[w] |= s_mask[i][w];
        }
      }
      s_nkept = n;
      // sorted keys: the first sentinel ends the candidates
      s_done = (n >= d.max_out) ? 1 : 0;
    }
    __syncthreads();
    if (s_done || !__syncthreads_or(valid && t == kNmsThreads - 1)) break;
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

extern "C" int pp_decode_strided_dev(pp_ctx_t *ctx, void *stream_, const float *cls_dev,
                                     const float *reg_dev, int64_t cls_stride_c,
                                     int64_t cls_stride_pix, int64_t reg_stride_c,
                                     int64_t reg_stride_pix, const double *a_centers, const double *a_wlh, const double *a_yaw,
                             const double *a_xy, const pp_decode_params_t *prm, double *boxes_out,
                             int32_t *kept_out, int32_t *count_out) {
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
  const size_t keys_bytes = cap * 8;
  bool grew = false;
  int rc = ctx->decode_ws.ensure(keys_bytes + 256, &grew);
  if (rc) return rc;
  if (grew)  // the candidate counter starts at zero; k_nms leaves it at zero
    PP_HIP_TRY(hipMemsetAsync(static_cast<char *>(ctx->decode_ws.ptr) + ctx->decode_ws.bytes - 256, 0, 256, stream));
  char *ws = static_cast<char *>(ctx->decode_ws.ptr);
  DecodeArgs d;
  d.cls = cls_dev;
  d.reg = reg_dev;
  d.cls_sc = cls_stride_c;
  d.cls_sp = cls_stride_pix;
  d.reg_sc = reg_stride_c;
  d.reg_sp = reg_stride_pix;
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
  d.keys = reinterpret_cast<u64 *>(ws);
  d.ncand = reinterpret_cast<int *>(ws + ctx->decode_ws.bytes - 256);
  d.cap = (int)cap;
  d.kept = kept_out;
  d.count = count_out;
  d.boxes = boxes_out;
  hipLaunchKernelGGL(k_score, dim3((unsigned)((A + 255) / 256)), dim3(256), 0, stream, d);
  hipLaunchKernelGGL(k_nms, dim3(1), dim3(kNmsThreads), 0, stream, d);  // + box decode
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
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
