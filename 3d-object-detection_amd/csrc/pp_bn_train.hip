// pp_bn_train.hip -- the ReLU -> BatchNorm2d tail of the backbone blocks in TRAINING mode.
//
// /root/reference model/model.py:76-84,105-109: every block is Conv2d -> ReLU -> BatchNorm2d.
// In train() PyTorch-ROCm runs the tail as a ReLU kernel + MIOpen's BatchNorm (forward 160 us,
// backward 180 us on a 64 MB activation) + a ReLU-backward kernel, and keeps both the conv
// output and the ReLU output for the backward.  Here:
//
//   forward   pass 1  per-channel sum / sum of squares of r = max(z + b, 0), taken about the
//                     channel's first activation (shifted data: no cancellation)  (k_rbn_stats)
//             pass 2  y = s*r + t with s = gamma*invstd, t = beta - mean*s; the first slice
//                     of every channel also updates the running statistics   (k_rbn_apply)
//   backward  pass 1  dbeta = sum dy, dgamma = sum dy*xhat (+ three sums for db)  (k_rbn_bwd_stats)
//             pass 2  dz = [z > 0] * s * (dy - dbeta/M - xhat*dgamma/M)     (k_rbn_bwd_apply)
//
// z may be the convolution WITHOUT its bias: the kernels add the per-channel bias b on the fly
// (one elementwise kernel less forward) and the backward returns db = sum dz from three extra
// per-channel sums of its first pass (one reduction kernel less per layer).  dy may be a
// channel slice of a wider NCHW tensor (torch.cat's gradient), read in place.
//
// Only z is kept for the backward.  Grid = (channels, slices): a workgroup owns one slice of
// one channel's planes; partial sums go to a [C][slices][2 or 5] f64 scratch, and the apply
// kernels reduce their channel's row themselves (no extra launch, no atomics, deterministic).
// HBM-bound: 12 B (forward) + 20 B (backward) per element.

#include "pp_common.h"

namespace pp {

constexpr int kRbnThreads = 256;
constexpr int kRbnMaxSplit = 64;

struct RbnGeom {
  int B, C, nsplit;
  int64_t HW;
  int64_t dy_extra;  // dy only: (batch stride of dy) - C*HW, for a channel slice of a wider tensor
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

// block-wide sums of NP doubles; results valid in every thread
template <int NP>
__device__ __forceinline__ void rbn_block_sum(double (&v)[NP], double (*s_red)[NP]) {
#pragma unroll
  for (int k = 0; k < NP; ++k)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v[k] += __shfl_xor(v[k], d, 64);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < NP; ++k) s_red[w][k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    v[k] = 0.0;
#pragma unroll
    for (int i = 0; i < kRbnThreads / 64; ++i) v[k] += s_red[i][k];
  }
  __syncthreads();
}

// sums over the slices of channel c of the NP partials; results valid in every thread
template <int NP>
__device__ __forceinline__ void rbn_channel_sums(const double *__restrict__ part, int c, int nsplit,
                                                 double (&v)[NP], double (*s_red)[NP]) {
#pragma unroll
  for (int k = 0; k < NP; ++k)
    v[k] = ((int)threadIdx.x < nsplit) ? part[((int64_t)c * nsplit + threadIdx.x) * NP + k] : 0.0;
  rbn_block_sum<NP>(v, s_red);
}

template <int NP>
__device__ __forceinline__ void rbn_store_partials(double *__restrict__ part, int c, int s, int nsplit,
                                                   const double (&v)[NP]) {
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NP; ++k) part[((int64_t)c * nsplit + s) * NP + k] = v[k];
  }
}

// The statistics are summed about a provisional value of the channel (shifted data): the
// activation of its first element.  On the sparse BEV canvas most of a plane sits at exactly
// that value, and E[r^2] - E[r]^2 would otherwise cancel.
__device__ __forceinline__ float rbn_shift(const float *__restrict__ z, float bc, int c, const RbnGeom &g) {
  return fmaxf(z[(int64_t)c * g.HW] + bc, 0.0f);
}

// f(offset in z / y / dz, offset in dy, elements)
template <typename F>
__device__ __forceinline__ void rbn_foreach(const RbnGeom &g, int c, int s, F &&f) {
  int64_t lo, hi;
  rbn_range(g, s, lo, hi);
  for (int b = 0; b < g.B; ++b) {
    const int64_t base = ((int64_t)b * g.C + c) * g.HW;
    const int64_t dbase = base + (int64_t)b * g.dy_extra;
    if (g.vec) {
      for (int64_t i = lo + (int64_t)threadIdx.x * 4; i < hi; i += kRbnThreads * 4) f(base + i, dbase + i, 4);
    } else {
      for (int64_t i = lo + threadIdx.x; i < hi; i += kRbnThreads) f(base + i, dbase + i, 1);
    }
  }
}

__global__ __launch_bounds__(kRbnThreads) void k_rbn_stats(const float *__restrict__ z,
                                                           const float *__restrict__ bias,
                                                           double *__restrict__ part, RbnGeom g) {
  __shared__ double s_red[kRbnThreads / 64][2];
  const int c = blockIdx.x, s = blockIdx.y;
  const float bc = bias ? bias[c] : 0.0f;
  const float c0 = rbn_shift(z, bc, c, g);
  float sum = 0.0f, sq = 0.0f;
  rbn_foreach(g, c, s, [&](int64_t o, int64_t od, int n) {
    if (n == 4) {
      const float4 v = *reinterpret_cast<const float4 *>(z + o);
      const float r0 = fmaxf(v.x + bc, 0.0f) - c0, r1 = fmaxf(v.y + bc, 0.0f) - c0,
                  r2 = fmaxf(v.z + bc, 0.0f) - c0, r3 = fmaxf(v.w + bc, 0.0f) - c0;
      sum += (r0 + r1) + (r2 + r3);
      sq = fmaf(r0, r0, fmaf(r1, r1, fmaf(r2, r2, fmaf(r3, r3, sq))));
    } else {
      const float r = fmaxf(z[o] + bc, 0.0f) - c0;
      sum += r;
      sq = fmaf(r, r, sq);
    }
  });
  double v[2] = {sum, sq};
  rbn_block_sum<2>(v, s_red);
  rbn_store_partials<2>(part, c, s, g.nsplit, v);
}

__global__ __launch_bounds__(kRbnThreads) void k_rbn_apply(
    const float *__restrict__ z, const float *__restrict__ bias, float *__restrict__ y,
    const double *__restrict__ part, const float *__restrict__ gamma, const float *__restrict__ beta,
    float *running_mean,
    float *running_var, float *__restrict__ mean_out, float *__restrict__ invstd_out, double eps,
    double momentum, RbnGeom g) {
  __shared__ double s_red[kRbnThreads / 64][2];
  const int c = blockIdx.x, s = blockIdx.y;
  const float bc = bias ? bias[c] : 0.0f;
  double v[2];
  rbn_channel_sums<2>(part, c, g.nsplit, v, s_red);
  const double M = (double)g.B * (double)g.HW;
  const double dm = v[0] / M;  // sums are about rbn_shift()
  const double mean = (double)rbn_shift(z, bc, c, g) + dm;
  double var = v[1] / M - dm * dm;  // biased: what BatchNorm normalises with
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
  rbn_foreach(g, c, s, [&](int64_t o, int64_t od, int n) {
    if (n == 4) {
      float4 q = *reinterpret_cast<const float4 *>(z + o);
      q.x = fmaf(fmaxf(q.x + bc, 0.0f), sc, sh);
      q.y = fmaf(fmaxf(q.y + bc, 0.0f), sc, sh);
      q.z = fmaf(fmaxf(q.z + bc, 0.0f), sc, sh);
      q.w = fmaf(fmaxf(q.w + bc, 0.0f), sc, sh);
      *reinterpret_cast<float4 *>(y + o) = q;
    } else {
      y[o] = fmaf(fmaxf(z[o] + bc, 0.0f), sc, sh);
    }
  });
}

// partials: sum dy, sum dy*xhat, and for the conv bias gradient sum_{z>0} dy, #{z>0},
// sum_{z>0} xhat (db = sum dz = s*(sum_{z>0} dy - k1*cnt - k2*sum_{z>0} xhat))
__global__ __launch_bounds__(kRbnThreads) void k_rbn_bwd_stats(const float *__restrict__ z,
                                                               const float *__restrict__ bias,
                                                               const float *__restrict__ dy,
                                                               const float *__restrict__ mean,
                                                               const float *__restrict__ invstd,
                                                               double *__restrict__ part, RbnGeom g) {
  __shared__ double s_red[kRbnThreads / 64][5];
  const int c = blockIdx.x, s = blockIdx.y;
  const float bc = bias ? bias[c] : 0.0f;
  const float mu = mean[c], is = invstd[c];
  float sd = 0.0f, sdx = 0.0f, md = 0.0f, mc = 0.0f, mx = 0.0f;
  auto one = [&](float zv, float d) {
    zv += bc;
    const float xh = (fmaxf(zv, 0.0f) - mu) * is;
    const float m = zv > 0.0f ? 1.0f : 0.0f;
    sd += d;
    sdx = fmaf(d, xh, sdx);
    md = fmaf(m, d, md);
    mc += m;
    mx = fmaf(m, xh, mx);
  };
  rbn_foreach(g, c, s, [&](int64_t o, int64_t od, int n) {
    if (n == 4) {
      const float4 v = *reinterpret_cast<const float4 *>(z + o);
      const float4 d = *reinterpret_cast<const float4 *>(dy + od);
      one(v.x, d.x);
      one(v.y, d.y);
      one(v.z, d.z);
      one(v.w, d.w);
    } else {
      one(z[o], dy[od]);
    }
  });
  double v[5] = {sd, sdx, md, mc, mx};
  rbn_block_sum<5>(v, s_red);
  rbn_store_partials<5>(part, c, s, g.nsplit, v);
}

__global__ __launch_bounds__(kRbnThreads) void k_rbn_bwd_apply(
    const float *__restrict__ z, const float *__restrict__ bias, const float *__restrict__ dy,
    const double *__restrict__ part, const float *__restrict__ gamma, const float *__restrict__ mean,
    const float *__restrict__ invstd, float *__restrict__ dz, float *__restrict__ dgamma,
    float *__restrict__ dbeta, float *__restrict__ dbias, RbnGeom g) {
  __shared__ double s_red[kRbnThreads / 64][5];
  const int c = blockIdx.x, s = blockIdx.y;
  const float bc = bias ? bias[c] : 0.0f;
  double v[5];
  rbn_channel_sums<5>(part, c, g.nsplit, v, s_red);  // v[0] = dbeta, v[1] = dgamma
  const double M = (double)g.B * (double)g.HW;
  const float mu = mean[c], is = invstd[c];
  const float sc = gamma[c] * is;
  if (s == 0 && threadIdx.x == 0) {
    dbeta[c] = (float)v[0];
    dgamma[c] = (float)v[1];
    if (dbias) dbias[c] = (float)((double)sc * (v[2] - v[0] / M * v[3] - v[1] / M * v[4]));
  }
  const float k1 = (float)(v[0] / M), k2 = (float)(v[1] / M);
  auto one = [&](float zv, float d) {
    zv += bc;
    const float xh = (fmaxf(zv, 0.0f) - mu) * is;
    return zv > 0.0f ? sc * (d - k1 - xh * k2) : 0.0f;
  };
  rbn_foreach(g, c, s, [&](int64_t o, int64_t od, int n) {
    if (n == 4) {
      const float4 q = *reinterpret_cast<const float4 *>(z + o);
      const float4 d = *reinterpret_cast<const float4 *>(dy + od);
      *reinterpret_cast<float4 *>(dz + o) = make_float4(one(q.x, d.x), one(q.y, d.y), one(q.z, d.z), one(q.w, d.w));
    } else {
      dz[o] = one(z[o], dy[od]);
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
  g->dy_extra = 0;
  g->vec = aligned && (hw % 4 == 0);
  // >= 2048 workgroups over the chip, a slice no smaller than one pass of the workgroup
  int ns = (2048 + channels - 1) / channels;
  const int64_t per_pass = (int64_t)kRbnThreads * (g->vec ? 4 : 1);
  ns = (int)std::min<int64_t>(ns, std::max<int64_t>(1, hw / per_pass));
  g->nsplit = std::max(1, std::min(ns, kRbnMaxSplit));
  int rc = ctx->pfn_ws.ensure((size_t)channels * kRbnMaxSplit * 5 * sizeof(double) + 4096);
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

extern "C" int pp_relu_bn_train_fwd_dev(pp_ctx_t *ctx, void *stream_, const float *z_dev,
                                        const float *conv_bias_dev, int64_t batch, int channels,
                                        int64_t hw, const float *gamma_dev,
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
  hipLaunchKernelGGL(k_rbn_stats, grid, dim3(kRbnThreads), 0, st, z_dev, conv_bias_dev, part, g);
  hipLaunchKernelGGL(k_rbn_apply, grid, dim3(kRbnThreads), 0, st, z_dev, conv_bias_dev, y_dev, part, gamma_dev, beta_dev,
                     running_mean_dev, running_var_dev, mean_out_dev, invstd_out_dev, eps, momentum, g);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}

extern "C" int pp_relu_bn_train_bwd_dev(pp_ctx_t *ctx, void *stream_, const float *z_dev,
                                        const float *conv_bias_dev, const float *dy_dev,
                                        int64_t dy_batch_stride, int64_t batch, int channels,
                                        int64_t hw, const float *gamma_dev,
                                        const float *mean_dev, const float *invstd_dev, float *dz_dev,
                                        float *dgamma_dev, float *dbeta_dev, float *dbias_dev) {
  RbnGeom g;
  double *part = nullptr;
  if (!gamma_dev || !mean_dev || !invstd_dev || !dgamma_dev || !dbeta_dev) {
    set_error("pp_relu_bn_train_bwd_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  int rc = rbn_setup("pp_relu_bn_train_bwd_dev", ctx, batch, channels, hw, {z_dev, dy_dev, dz_dev}, &g,
                     &part);
  if (rc) return rc;
  if (dy_batch_stride == 0) dy_batch_stride = (int64_t)channels * hw;
  if (dy_batch_stride < (int64_t)channels * hw) {
    set_error("pp_relu_bn_train_bwd_dev: dy_batch_stride %lld < channels*hw", (long long)dy_batch_stride);
    return PP_ERR_VALUE;
  }
  g.dy_extra = dy_batch_stride - (int64_t)channels * hw;
  if (g.dy_extra % 4 != 0) g.vec = false;  // float4 alignment of the later batches of dy
  RbnDevice guard(ctx->device);
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const dim3 grid((unsigned)channels, (unsigned)g.nsplit);
  hipLaunchKernelGGL(k_rbn_bwd_stats, grid, dim3(kRbnThreads), 0, st, z_dev, conv_bias_dev, dy_dev, mean_dev,
                     invstd_dev, part, g);
  hipLaunchKernelGGL(k_rbn_bwd_apply, grid, dim3(kRbnThreads), 0, st, z_dev, conv_bias_dev, dy_dev, part,
                     gamma_dev, mean_dev, invstd_dev, dz_dev, dgamma_dev, dbeta_dev, dbias_dev, g);
  PP_HIP_TRY(hipGetLastError());
  return PP_OK;
}
