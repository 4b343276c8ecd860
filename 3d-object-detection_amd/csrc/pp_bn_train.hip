// pp_bn_train.hip -- the ReLU -> BatchNorm2d tail of the backbone blocks in TRAINING mode.
//
// /root/reference model/model.py:76-84,105-109: every block is Conv2d -> ReLU -> BatchNorm2d.
// In train() PyTorch-ROCm runs the tail as a ReLU kernel + MIOpen's BatchNorm (forward 160 us,
// backward 180 us on a 64 MB activation) + a ReLU-backward kernel, and keeps both the conv
// output and the ReLU output for the backward.  Here:
//
//   forward   pass 1  per-channel sum / sum of squares of r = max(z, 0)      (k_rbn_stats)
//             pass 2  y = s*r + t with s = gamma*invstd, t = beta - mean*s; the first slice
//                     of every channel also updates the running statistics   (k_rbn_apply)
//   backward  pass 1  dbeta = sum dy, dgamma = sum dy*xhat                  (k_rbn_bwd_stats)
//             pass 2  dz = [z > 0] * s * (dy - dbeta/M - xhat*dgamma/M)     (k_rbn_bwd_apply)
//
// Only z is kept for the backward.  Grid = (channels, slices): a workgroup owns one slice of
// one channel's planes; partial sums go to a [C][slices][2] f64 scratch, and the apply kernels
// reduce their channel's row themselves (no extra launch, no atomics, deterministic).
// HBM-bound: 12 B (forward) + 20 B (backward) per element.

#include "pp_common.h"

namespace pp {

constexpr int kRbnThreads = 256;
constexpr int kRbnMaxSplit = 64;

struct RbnGeom {
  int B, C, nsplit;
  int64_t HW;
  bool vec;  // HW % 4 == 0 and 16-byte aligned tensors: float4 accesses
};

__device__ __forceinline__ void rbn_range(const RbnGeom &g, int s, int64_t &lo, int64_t &hi) {
  lo = g.HW * s / g.nsplit;
  hi = g.HW * (s + 1) / g.nsplit;
  if (g.vec) {
    lo &= ~(int64_t)3;
    hi = (s + 1 == g.nsplit) ? g.HW : (hi & ~(int64_t)3);
  }
}

// block-wide sum of two doubles; result valid in every thread
__device__ __forceinline__ void rbn_block_sum(double &a, double &b, double (*s_red)[2]) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    a += __shfl_xor(a, d, 64);
    b += __shfl_xor(b, d, 64);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_red[w][0] = a;
    s_red[w][1] = b;
  }
  __syncthreads();
  a = 0.0;
  b = 0.0;
#pragma unroll
  for (int i = 0; i < kRbnThreads / 64; ++i) {
    a += s_red[i][0];
    b += s_red[i][1];
  }
  __syncthreads();
}

// sum over the slices of channel c of the two partials; result valid in every thread
__device__ __forceinline__ void rbn_channel_sums(const double *__restrict__ part, int c, int nsplit,
                                                 double &a, double &b, double (*s_red)[2]) {
  a = 0.0;
  b = 0.0;
  if ((int)threadIdx.x < nsplit) {
    a = part[((int64_t)c * nsplit + threadIdx.x) * 2];
    b = part[((int64_t)c * nsplit + threadIdx.x) * 2 + 1];
  }
  rbn_block_sum(a, b, s_red);
}

template <typename F>
__device__ __forceinline__ void rbn_foreach(const RbnGeom &g, int c, int s, F &&f) {
  int64_t lo, hi;
  rbn_range(g, s, lo, hi);
  for (int b = 0; b < g.B; ++b) {
    const int64_t base = ((int64_t)b * g.C + c) * g.HW;
    if (g.vec) {
      for (int64_t i = lo + (int64_t)threadIdx.x * 4; i < hi; i += kRbnThreads * 4) f(base + i, 4);
    } else {
      for (int64_t i = lo + threadIdx.x; i < hi; i += kRbnThreads) f(base + i, 1);
    }
  }
}

__global__ __launch_bounds__(kRbnThreads) void k_rbn_stats(const float *__restrict__ z,
                                                           double *__restrict__ part, RbnGeom g) {
  __shared__ double s_red[kRbnThreads / 64][2];
  const int c = blockIdx.x, s = blockIdx.y;
  float sum = 0.0f, sq = 0.0f;
  rbn_foreach(g, c, s, [&](int64_t o, int n) {
    if (n == 4) {
      const float4 v = *reinterpret_cast<const float4 *>(z + o);
      const float r0 = fmaxf(v.x, 0.0f), r1 = fmaxf(v.y, 0.0f), r2 = fmaxf(v.z, 0.0f),
                  r3 = fmaxf(v.w, 0.0f);
      sum += (r0 + r1) + (r2 + r3);
      sq = fmaf(r0, r0, fmaf(r1, r1, fmaf(r2, r2, fmaf(r3, r3, sq))));
    } else {
      const float r = fmaxf(z[o], 0.0f);
      sum += r;
      sq = fmaf(r, r, sq);
    }
  });
  double a = sum, b = sq;
  rbn_block_sum(a, b, s_red);
  if (threadIdx.x == 0) {
    part[((int64_t)c * g.nsplit + s) * 2] = a;
    part[((int64_t)c * g.nsplit + s) * 2 + 1] = b;
  }
}

__global__ __launch_bounds__(kRbnThreads) void k_rbn_apply(
    const float *__restrict__ z, float *__restrict__ y, const double *__restrict__ part,
    const float *__restrict__ gamma, const float *__restrict__ beta, float *running_mean,
    float *running_var, float *__restrict__ mean_out, float *__restrict__ invstd_out, double eps,
    double momentum, RbnGeom g) {
  __shared__ double s_red[kRbnThreads / 64][2];
  const int c = blockIdx.x, s = blockIdx.y;
  double a, b;
  rbn_channel_sums(part, c, g.nsplit, a, b, s_red);
  const double M = (double)g.B * (double)g.HW;
  const double mean = a / M;
  double var = b / M - mean * mean;  // biased: what BatchNorm normalises with
  var = var > 0.0 ? var : 0.0;
  const double invstd = 1.0 / sqrt(var + eps);
  const float sc = (float)((double)gamma[c] * invstd);
  const float sh = (float)((double)beta[c] - mean * (double)gamma[c] * invstd);
  if (s == 0 && threadIdx.x == 0) {
    mean_out[c] = (float)mean;
    invstd_out[c] = (float)invstd;
    if (running_mean) {
      const double unbiased = M > 1.0 ? var * (M / (M - 1.0)) : var;
      running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
      running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
  }
  rbn_foreach(g, c, s, [&](int64_t o, int n) {
    if (n == 4) {
      float4 v = *reinterpret_cast<const float4 *>(z + o);
      v.x = fmaf(fmaxf(v.x, 0.0f), sc, sh);
      v.y = fmaf(fmaxf(v.y, 0.0f), sc, sh);
      v.z = fmaf(fmaxf(v.z, 0.0f), sc, sh);
      v.w = fmaf(fmaxf(v.w, 0.0f), sc, sh);
      *reinterpret_cast<float4 *>(y + o) = v;
    } else {
      y[o] = fmaf(fmaxf(z[o], 0.0f), sc, sh);
    }
  });
}

__global__ __launch_bounds__(kRbnThreads) void k_rbn_bwd_stats(const float *__restrict__ z,
                                                               const float *__restrict__ dy,
                                                               const float *__restrict__ mean,
                                                               const float *__restrict__ invstd,
                                                               double *__restrict__ part, RbnGeom g) {
  __shared__ double s_red[kRbnThreads / 64][2];
  const int c = blockIdx.x, s = blockIdx.y;
  const float mu = mean[c], is = invstd[c];
  float sd = 0.0f, sdx = 0.0f;
  rbn_foreach(g, c, s, [&](int64_t o, int n) {
    if (n == 4) {
      const float4 v = *reinterpret_cast<const float4 *>(z + o);
      const float4 d = *reinterpret_cast<const float4 *>(dy + o);
      sd += (d.x + d.y) + (d.z + d.w);
      sdx = fmaf(d.x, (fmaxf(v.x, 0.0f) - mu) * is, sdx);
      sdx = fmaf(d.y, (fmaxf(v.y, 0.0f) - mu) * is, sdx);
      sdx = fmaf(d.z, (fmaxf(v.z, 0.0f) - mu) * is, sdx);
      sdx = fmaf(d.w, (fmaxf(v.w, 0.0f) - mu) * is, sdx);
    } else {
      const float d = dy[o];
      sd += d;
      sdx = fmaf(d, (fmaxf(z[o], 0.0f) - mu) * is, sdx);
    }
  });
  double a = sd, b = sdx;
  rbn_block_sum(a, b, s_red);
  if (threadIdx.x == 0) {
    part[((int64_t)c * g.nsplit + s) * 2] = a;
    part[((int64_t)c * g.nsplit + s) * 2 + 1] = b;
  }
}

__global__ __launch_bounds__(kRbnThreads) void k_rbn_bwd_apply(
    const float *__restrict__ z, const float *__restrict__ dy, const double *__restrict__ part,
    const float *__restrict__ gamma, const float *__restrict__ mean,
    const float *__restrict__ invstd, float *__restrict__ dz, float *__restrict__ dgamma,
    float *__restrict__ dbeta, RbnGeom g) {
  __shared__ double s_red[kRbnThreads / 64][2];
  const int c = blockIdx.x, s = blockIdx.y;
  double a, b;
  rbn_channel_sums(part, c, g.nsplit, a, b, s_red);  // a = dbeta, b = dgamma
  if (s == 0 && threadIdx.x == 0) {
    dbeta[c] = (float)a;
    dgamma[c] = (float)b;
  }
  const double M = (double)g.B * (double)g.HW;
  const float mu = mean[c], is = invstd[c];
  const float sc = gamma[c] * is;
  const float k1 = (float)(a / M), k2 = (float)(b / M);
  auto one = [&](float zv, float d) {
    const float xh = (fmaxf(zv, 0.0f) - mu) * is;
    return zv > 0.0f ? sc * (d - k1 - xh * k2) : 0.0f;
  };
  rbn_foreach(g, c, s, [&](int64_t o, int n) {
    if (n == 4) {
      const float4 v = *reinterpret_cast<const float4 *>(z + o);
      const float4 d = *reinterpret_cast<const float4 *>(dy + o);
      *reinterpret_cast<float4 *>(dz + o) = make_float4(one(v.x, d.x), one(v.y, d.y), one(v.z, d.z), one(v.w, d.w));
    } else {
      dz[o] = one(z[o], dy[o]);
    }
  });
}

static int rbn_setup(const char *who, pp_ctx_t *ctx, int64_t batch, int channels, int64_t hw,
                     std::initializer_list<const void *> tensors, RbnGeom *g, double **part) {
  if (!ctx) {
    set_error("%s: ctx is NULL", who);
    return PP_ERR_VALUE;
  }
  bool aligned = true;
  for (const void *t : tensors) {
    if (!t) {
      set_error("%s: NULL argument", who);
      return PP_ERR_VALUE;
    }
    aligned = aligned && ((reinterpret_cast<uintptr_t>(t) & 15) == 0);
  }
  if (batch < 1 || batch > (1 << 20) || channels < 1 || channels > 65535 || hw < 1 ||
      batch * hw > (1ll << 40)) {
    set_error("%s: bad sizes (batch=%lld channels=%d hw=%lld)", who, (long long)batch, channels,
              (long long)hw);
    return PP_ERR_VALUE;
  }
  g->B = (int)batch;
  g->C = channels;
  g->HW = hw;
  g->vec = aligned && (hw % 4 == 0);
  // >= 2048 workgroups over the chip, a slice no smaller than one pass of the workgroup
  int ns = (2048 + channels - 1) / channels;
  const int64_t per_pass = (int64_t)kRbnThreads * (g->vec ? 4 : 1);
  ns = (int)std::min<int64_t>(ns, std::max<int64_t>(1, hw / per_pass));
  g->nsplit = std::max(1, std::min(ns, kRbnMaxSplit));
  int rc = ctx->pfn_ws.ensure((size_t)channels * kRbnMaxSplit * 2 * sizeof(double) + 4096);
  if (rc) return rc;
  *part = static_cast<double *>(ctx->pfn_ws.ptr);
  return PP_OK;
}

struct RbnDevice {
  int prev = -1, dev;
  explicit RbnDevice(int d) : dev(d) {
    (void)hipGetDevice(&prev);
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~RbnDevice() {
    if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
  }
};

}  // namespace pp

using namespace pp;

extern "C" int pp_relu_bn_train_fwd_dev(pp_ctx_t *ctx, void *stream_, const float *z_dev, int64_t batch,
                                        int channels, int64_t hw, const float *gamma_dev,
                                        const float *beta_dev, double eps, double momentum,
                                        float *running_mean_dev, float *running_var_dev, float *y_dev,
                                        float *mean_out_dev, float *invstd_out_dev) {
  RbnGeom g;
  double *part = nullptr;
  if (!gamma_dev || !beta_dev || !mean_out_dev || !invstd_out_dev ||
      ((running_mean_dev == nullptr) != (running_var_dev == nullptr))) {
    set_error("pp_relu_bn_train_fwd_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  int rc = rbn_setup("pp_relu_bn_train_fwd_dev", ctx, batch, channels, hw, {z_dev, y_dev}, &g, &part);
  if (rc) return rc;
  RbnDevice guard(ctx->device);
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const dim3 grid((unsigned)channels, (unsigned)g.nsplit);
  hipLaunchKernelGGL(k_rbn_stats, grid, dim3(kRbnThreads), 0, st, z_dev, part, g);
  hipLaunchKernelGGL(k_rbn_apply, grid, dim3(kRbnThreads), 0, st, z_dev, y_dev, part, gamma_dev, beta_dev,
                     running_mean_dev, running_var_dev, mean_out_dev, invstd_out_dev, eps, momentum, g);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}

extern "C" int pp_relu_bn_train_bwd_dev(pp_ctx_t *ctx, void *stream_, const float *z_dev,
                                        const float *dy_dev, int64_t batch, int channels, int64_t hw,
                                        const float *gamma_dev, const float *mean_dev,
                                        const float *invstd_dev, float *dz_dev, float *dgamma_dev,
                                        float *dbeta_dev) {
  RbnGeom g;
  double *part = nullptr;
  if (!gamma_dev || !mean_dev || !invstd_dev || !dgamma_dev || !dbeta_dev) {
    set_error("pp_relu_bn_train_bwd_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  int rc = rbn_setup("pp_relu_bn_train_bwd_dev", ctx, batch, channels, hw, {z_dev, dy_dev, dz_dev}, &g,
                     &part);
  if (rc) return rc;
  RbnDevice guard(ctx->device);
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const dim3 grid((unsigned)channels, (unsigned)g.nsplit);
  hipLaunchKernelGGL(k_rbn_bwd_stats, grid, dim3(kRbnThreads), 0, st, z_dev, dy_dev, mean_dev, invstd_dev,
                     part, g);
  hipLaunchKernelGGL(k_rbn_bwd_apply, grid, dim3(kRbnThreads), 0, st, z_dev, dy_dev, part, gamma_dev,
                     mean_dev, invstd_dev, dz_dev, dgamma_dev, dbeta_dev, g);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}
