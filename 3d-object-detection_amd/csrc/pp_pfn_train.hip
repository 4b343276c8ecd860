// pp_pfn_train.hip -- PPFeatureNet in TRAINING mode without the [B,64,P,N] intermediate.
//
// /root/reference model/model.py:31-40 in train(): z = W x + b, r = ReLU(z), BatchNorm2d
// with batch statistics over (B,P,N), max over N.  PyTorch-ROCm materialises the 64x inflated
// intermediate (1.2 GB at B=4, P=12000, N=100) five times forward and again backward: 9.6 ms
// of a 29.6 ms training step.  Everything the step needs is a handful of PER-CHANNEL SUMS
// over that intermediate plus per-(b,c,p) work on the argmax element:
//
//   forward   batch statistics       sum (r-c0), sum (r-c0)^2, c0 = max(b,0) (k_pfn_train_stats)
//             y = s*r + t is monotone, so out = s >= 0 ? s*max r + t : s*min r + t
//                                                                        (pp_pfn_dense_dev)
//   backward  dy is G[b,c,p] at the argmax element and 0 elsewhere, so
//               dbeta = sum G,  dgamma = sum G*xhat*                     (k_pfn_train_bwd)
//               dr    = s*(dy - dbeta/M - xhat*dgamma/M) = s*dy + A_c + B_c*r   (dense part)
//               dz    = dr * [z > 0]
//               dW[c,d] = sum dz*x_d = sparse + A_c*S1[c,d] + B_c*S2[c,d]
//               db[c]   = sum dz     = sparse + A_c*cnt[c]  + B_c*sum r
//             with S1[c,d] = sum_{z>0} x_d, S2[c,d] = sum r*x_d, cnt = #{z>0}: independent of
//             G, accumulated in the SAME pass as the statistics.
//
// Both kernels recompute z with the forward's fmaf chain (bias first, features in order), so
// the z > 0 decisions agree everywhere.  Lanes carry channels, the wave's pillars' feature rows
// are staged in LDS; per-lane f32 accumulators over a grid-stride range of pillars, one partial
// row per workgroup, summed in f64 by a second tiny kernel: no atomics, deterministic.
// VALU-bound: 31 (stats) / 12 (backward) operations per (point, channel).

#include "pp_common.h"

namespace pp {

constexpr int kTrC = 64;       // channels = lanes
constexpr int kTrWaves = 4;    // waves per workgroup
constexpr int kTrChunk = 256;  // points staged per pass
constexpr int kTrStats = 21;   // cnt, sum r, sum r^2, S1[9], S2[9]
constexpr int kTrBwd = 12;     // dbeta, dgamma, db(sparse), dW(sparse)[9]
constexpr int kTrMaxWg = 1024;

__device__ __forceinline__ void tr_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct TrLane {
  float w[9], bias;
};

__device__ __forceinline__ float tr_z(const TrLane &A, const float x[9]) {
  float z = A.bias;
#pragma unroll
  for (int d = 0; d < 9; ++d) z = fmaf(A.w[d], x[d], z);
  return z;
}

// sums[k][c] partial of this workgroup -> part[wg][k][c]
template <int K>
__device__ __forceinline__ void tr_flush(const float (&acc)[K], float (*s_red)[K][kTrC], int wave,
                                         int lane, float *__restrict__ part) {
#pragma unroll
  for (int k = 0; k < K; ++k) s_red[wave][k][lane] = acc[k];
  __syncthreads();
  for (int i = threadIdx.x; i < K * kTrC; i += kTrWaves * 64) {
    const int k = i / kTrC, c = i - k * kTrC;
    float v = 0.0f;
#pragma unroll
    for (int w = 0; w < kTrWaves; ++w) v += s_red[w][k][c];
    part[((int64_t)blockIdx.x * K + k) * kTrC + c] = v;
  }
}

// x [B][9][P][N]; wb [64][10] {w[0..8], bias}; part [gridDim.x][21][64]
__global__ __launch_bounds__(kTrWaves * 64) void k_pfn_train_stats(const float *__restrict__ x,
                                                                   const float *__restrict__ wb,
                                                                   float *__restrict__ part, int B,
                                                                   int P, int N) {
  __shared__ float s_x[kTrWaves][9][kTrChunk];
  __shared__ float s_red[kTrWaves][kTrStats][kTrC];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float(*sx)[kTrChunk] = s_x[wave];
  TrLane A;
#pragma unroll
  for (int d = 0; d < 9; ++d) A.w[d] = wb[lane * 10 + d];
  A.bias = wb[lane * 10 + 9];
  // Most slots are zero padding, where r = max(bias, 0) exactly: the statistics are summed
  // about that value (shifted data), or E[r^2] - E[r]^2 would cancel catastrophically for a
  // channel whose live points barely move it.
  const float c0 = fmaxf(A.bias, 0.0f);
  float acc[kTrStats];
#pragma unroll
  for (int k = 0; k < kTrStats; ++k) acc[k] = 0.0f;
  const int64_t plane = (int64_t)P * N;
  // the points of one sweep are one contiguous span of P*N floats in every feature row
  const int64_t span = plane;
  const int64_t chunks_per_sweep = (span + kTrChunk - 1) / kTrChunk;
  const int64_t total = chunks_per_sweep * B;
  for (int64_t ch = (int64_t)blockIdx.x * kTrWaves + wave; ch < total;
       ch += (int64_t)gridDim.x * kTrWaves) {
    const int b = (int)(ch / chunks_per_sweep);
    const int64_t q0 = (ch - (int64_t)b * chunks_per_sweep) * kTrChunk;
    const int cn = (int)min((int64_t)kTrChunk, span - q0);
    const float *row = x + (int64_t)b * 9 * plane + q0;
    for (int i = lane; i < cn; i += 64) {
#pragma unroll
      for (int d = 0; d < 9; ++d) sx[d][i] = row[d * plane + i];
    }
    tr_wave_sync();
    for (int j = 0; j < cn; ++j) {
      float xv[9];
#pragma unroll
      for (int d = 0; d < 9; ++d) xv[d] = sx[d][j];
      const float z = tr_z(A, xv);
      const float r = fmaxf(z, 0.0f);
      const float mk = z > 0.0f ? 1.0f : 0.0f;
      const float rs = r - c0;  // exactly 0 on a zero-padded slot
      acc[0] += mk;
      acc[1] += rs;
      acc[2] = fmaf(rs, rs, acc[2]);
#pragma unroll
      for (int d = 0; d < 9; ++d) {
        acc[3 + d] = fmaf(mk, xv[d], acc[3 + d]);
        acc[12 + d] = fmaf(r, xv[d], acc[12 + d]);
      }
    }
    tr_wave_sync();
  }
  tr_flush<kTrStats>(acc, s_red, wave, lane, part);
}

// part [nwg][K][64] f32 -> sums [K][64] f64; one workgroup per k: 16 row groups x 64 channels,
// four independent accumulators per thread (the loads are a latency chain otherwise)
__global__ __launch_bounds__(1024) void k_pfn_train_reduce(const float *__restrict__ part, int nwg, int K,
                                                           double *__restrict__ sums) {
  __shared__ double s[16][kTrC];
  const int k = blockIdx.x, c = threadIdx.x & 63, q = threadIdx.x >> 6;
  double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
  const float *p = part + (int64_t)k * kTrC + c;
  const int64_t stride = (int64_t)K * kTrC;
  int w = q;
  for (; w + 48 < nwg; w += 64) {
    v0 += (double)p[(int64_t)w * stride];
    v1 += (double)p[(int64_t)(w + 16) * stride];
    v2 += (double)p[(int64_t)(w + 32) * stride];
    v3 += (double)p[(int64_t)(w + 48) * stride];
  }
  for (; w < nwg; w += 16) v0 += (double)p[(int64_t)w * stride];
  s[q][c] = (v0 + v1) + (v2 + v3);
  __syncthreads();
  if (q == 0) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += s[i][c];
    sums[k * kTrC + c] = t;
  }
}

// Backward, the per-(b,c,p) part.  prm [64][12] {w[0..8], bias, scale, shift} (the forward
// table), mu / invstd [64], g [B][64][P]; part [gridDim.x][12][64]
__global__ __launch_bounds__(kTrWaves * 64) void k_pfn_train_bwd(
    const float *__restrict__ x, const float *__restrict__ prm, const float *__restrict__ mu,
    const float *__restrict__ invstd, const float *__restrict__ g, float *__restrict__ part, int B,
    int P, int N) {
  __shared__ float s_x[kTrWaves][9][kTrChunk];
  __shared__ float s_red[kTrWaves][kTrBwd][kTrC];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float(*sx)[kTrChunk] = s_x[wave];
  TrLane A;
#pragma unroll
  for (int d = 0; d < 9; ++d) A.w[d] = prm[lane * 12 + d];
  A.bias = prm[lane * 12 + 9];
  const float scale = prm[lane * 12 + 10];
  const float mu_c = mu[lane], is_c = invstd[lane];
  float acc[kTrBwd];
#pragma unroll
  for (int k = 0; k < kTrBwd; ++k) acc[k] = 0.0f;
  const int64_t plane = (int64_t)P * N;
  const int64_t total = (int64_t)B * P;
  for (int64_t pp = (int64_t)blockIdx.x * kTrWaves + wave; pp < total;
       pp += (int64_t)gridDim.x * kTrWaves) {
    const int b = (int)(pp / P), p = (int)(pp - (int64_t)b * P);
    const float *row = x + (int64_t)b * 9 * plane + (int64_t)p * N;
    // the extreme of z over the pillar's N slots that the forward selected: the maximum for a
    // non-negative BatchNorm scale, else the minimum; first occurrence (torch.max's index)
    float best = 0.0f;
    int best_n = -1;
    for (int n0 = 0; n0 < N; n0 += kTrChunk) {
      const int cn = min(kTrChunk, N - n0);
      for (int i = lane; i < cn; i += 64) {
#pragma unroll
        for (int d = 0; d < 9; ++d) sx[d][i] = row[d * plane + n0 + i];
      }
      tr_wave_sync();
      for (int j = 0; j < cn; ++j) {
        float xv[9];
#pragma unroll
        for (int d = 0; d < 9; ++d) xv[d] = sx[d][j];
        const float r = fmaxf(tr_z(A, xv), 0.0f);
        const bool better = best_n < 0 || (scale >= 0.0f ? r > best : r < best);
        best = better ? r : best;
        best_n = better ? n0 + j : best_n;
      }
      tr_wave_sync();
    }
    const float gv = g[((int64_t)b * kTrC + lane) * P + p];
    acc[0] += gv;                                   // dbeta
    acc[1] = fmaf(gv, (best - mu_c) * is_c, acc[1]);  // dgamma: G * xhat at the selected element
    if (best > 0.0f) {                              // ReLU passes the gradient only where z > 0
      const float dz = gv * scale;
      acc[2] += dz;
#pragma unroll
      for (int d = 0; d < 9; ++d) acc[3 + d] = fmaf(dz, row[d * plane + best_n], acc[3 + d]);
    }
  }
  tr_flush<kTrBwd>(acc, s_red, wave, lane, part);
}

static int tr_grid(int64_t items) {
  const int64_t wg = (items + kTrWaves - 1) / kTrWaves;
  return (int)std::max<int64_t>(1, std::min<int64_t>(wg, kTrMaxWg));
}

}  // namespace pp

using namespace pp;

static int tr_check(const char *who, pp_ctx_t *ctx, const void *a, const void *b, const void *c,
                    int batch, int P, int N) {
  if (!ctx || !a || !b || !c) {
    set_error("%s: NULL argument", who);
    return PP_ERR_VALUE;
  }
  if (batch < 1 || batch > 65535 || P < 1 || N < 1 || (int64_t)P * N > (1ll << 31)) {
    set_error("%s: bad sizes (batch=%d P=%d N=%d)", who, batch, P, N);
    return PP_ERR_VALUE;
  }
  return PP_OK;
}

extern "C" int pp_pfn_train_stats_dev(pp_ctx_t *ctx, void *stream_, const float *pillars_dev, int batch,
                                      int max_pillars, int max_points_per_pillar,
                                      const float *weight_bias_dev, int channels, double *sums_dev) {
  int rc = tr_check("pp_pfn_train_stats_dev", ctx, pillars_dev, weight_bias_dev, sums_dev, batch,
                    max_pillars, max_points_per_pillar);
  if (rc) return rc;
  if (channels != kTrC) {
    set_error("the feature-net kernels are built for %d output channels (got %d)", kTrC, channels);
    return PP_ERR_VALUE;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != ctx->device) (void)hipSetDevice(ctx->device);
  struct Restore {
    int prev, dev;
    ~Restore() {
      if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
  } restore{prev, ctx->device};
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int64_t span = (int64_t)max_pillars * max_points_per_pillar;
  const int nwg = tr_grid(((span + kTrChunk - 1) / kTrChunk) * batch);
  rc = ctx->pfn_ws.ensure((size_t)kTrMaxWg * kTrStats * kTrC * sizeof(float));
  if (rc) return rc;
  float *part = static_cast<float *>(ctx->pfn_ws.ptr);
  hipLaunchKernelGGL(k_pfn_train_stats, dim3(nwg), dim3(kTrWaves * 64), 0, st, pillars_dev,
                     weight_bias_dev, part, batch, max_pillars, max_points_per_pillar);
  hipLaunchKernelGGL(k_pfn_train_reduce, dim3(kTrStats), dim3(1024), 0, st, part, nwg, kTrStats, sums_dev);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}

extern "C" int pp_pfn_train_backward_dev(pp_ctx_t *ctx, void *stream_, const float *pillars_dev,
                                         int batch, int max_pillars, int max_points_per_pillar,
                                         const float *pfn_params_dev, const float *mean_dev,
                                         const float *invstd_dev, const float *grad_out_dev,
                                         int channels, double *sums_dev) {
  int rc = tr_check("pp_pfn_train_backward_dev", ctx, pillars_dev, pfn_params_dev, sums_dev, batch,
                    max_pillars, max_points_per_pillar);
  if (rc) return rc;
  if (!mean_dev || !invstd_dev || !grad_out_dev) {
    set_error("pp_pfn_train_backward_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  if (channels != kTrC) {
    set_error("the feature-net kernels are built for %d output channels (got %d)", kTrC, channels);
    return PP_ERR_VALUE;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != ctx->device) (void)hipSetDevice(ctx->device);
  struct Restore {
    int prev, dev;
    ~Restore() {
      if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
  } restore{prev, ctx->device};
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int nwg = tr_grid((int64_t)batch * max_pillars);
  rc = ctx->pfn_ws.ensure((size_t)kTrMaxWg * kTrStats * kTrC * sizeof(float));
  if (rc) return rc;
  float *part = static_cast<float *>(ctx->pfn_ws.ptr);
  hipLaunchKernelGGL(k_pfn_train_bwd, dim3(nwg), dim3(kTrWaves * 64), 0, st, pillars_dev, pfn_params_dev,
                     mean_dev, invstd_dev, grad_out_dev, part, batch, max_pillars,
                     max_points_per_pillar);
  hipLaunchKernelGGL(k_pfn_train_reduce, dim3(kTrBwd), dim3(1024), 0, st, part, nwg, kTrBwd, sums_dev);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}
