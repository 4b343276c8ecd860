// pp_epilogue.hip -- the elementwise / data-movement kernels around the network (inference):
//   k_bias_relu_bn[_nhwc]  the backbone's Conv2d -> ReLU -> BatchNorm2d tail in one pass
//   k_subtract_mean        pillar -= data_mean (data/dataset.py:102-105)
//   k_scatter_canvas       PPScatter.forward (model/model.py:53-62)
//
// Every block of the reference backbone is Conv2d -> ReLU -> BatchNorm2d
// (/root/reference model/model.py:76-84, 105-109).  In eval mode PyTorch-ROCm runs
// that as MIOpen conv + a bias kernel + a ReLU kernel + a BatchNorm kernel: three
// extra read+write passes over every activation tensor.  This kernel does
// y = max(x + b_c, 0) * s_c + t_c in place, in one pass (s = gamma/sqrt(var+eps),
// t = beta - mean*s): HBM-bound, 8 bytes of traffic per element.

#include "pp_common.h"

namespace pp {

// x: [B][C][hw] contiguous.  y: plane (b,c) lives at y + (b*y_batch_stride + c)*hw, which
// is x itself for the in-place form and a channel slice of a wider tensor otherwise (the
// up blocks write straight into the concatenated [B,6C,h,w] output, model/model.py:140).
__global__ __launch_bounds__(256) void k_bias_relu_bn(const float *__restrict__ x,
                                                      float *__restrict__ y, int C, int64_t hw,
                                                      int64_t y_batch_stride,
                                                      const float *__restrict__ prm) {
  const int64_t plane = blockIdx.y;  // b*C + c
  const int64_t bi = plane / C;
  const int c = (int)(plane - bi * C);
  const float b = prm[c * 3 + 0], s = prm[c * 3 + 1], t = prm[c * 3 + 2];
  const float *p = x + plane * hw;
  float *q = y + (bi * y_batch_stride + c) * hw;
  const int64_t n4 = hw >> 2;
  const bool aligned = (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(q)) & 15) == 0);
  if (aligned) {
    const float4 *p4 = reinterpret_cast<const float4 *>(p);
    float4 *q4 = reinterpret_cast<float4 *>(q);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      float4 v = p4[i];
      v.x = fmaxf(v.x + b, 0.0f) * s + t;
      v.y = fmaxf(v.y + b, 0.0f) * s + t;
      v.z = fmaxf(v.z + b, 0.0f) * s + t;
      v.w = fmaxf(v.w + b, 0.0f) * s + t;
      q4[i] = v;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw;
         i += (int64_t)gridDim.x * 256)
      q[i] = fmaxf(p[i] + b, 0.0f) * s + t;
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256)
      q[i] = fmaxf(p[i] + b, 0.0f) * s + t;
  }
}

// NHWC form: x [pixels][C] contiguous, y rows of y_stride floats (a channel slice of the
// concatenated output, or x itself).  One float4 = 4 consecutive channels per lane; when C/4
// divides the grid stride a lane keeps its channel group, so its 12 table values sit in
// registers for the whole loop.
template <bool kFixedLane>
__global__ __launch_bounds__(256) void k_bias_relu_bn_nhwc(const float4 *__restrict__ x,
                                                           float *__restrict__ y, int c4,
                                                           int64_t n4, int64_t y_stride,
                                                           const float *__restrict__ prm) {
  const int64_t step = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float4 b, s, t;
  auto load_table = [&](int g) {
    const float *q = prm + (int64_t)g * 12;
    b = make_float4(q[0], q[3], q[6], q[9]);
    s = make_float4(q[1], q[4], q[7], q[10]);
    t = make_float4(q[2], q[5], q[8], q[11]);
  };
  if (kFixedLane) {
    // c4 divides the grid stride: the lane keeps its channel group AND its pixel advances by a constant -- ONE
    // 64-bit division per thread instead of one per 16 bytes (no measurable change end to end: the kernel is a
    // read-modify-write stream at ~3.8 TB/s of traffic either way)
    int64_t pix = i / c4;
    const int g = (int)(i - pix * c4);
    const int64_t pstep = step / c4;
    load_table(g);
    float *yp = y + pix * y_stride + (int64_t)g * 4;
    const int64_t ystep = pstep * y_stride;
    for (; i < n4; i += step, yp += ystep) {
      float4 v = x[i];
      v.x = fmaxf(v.x + b.x, 0.0f) * s.x + t.x;
      v.y = fmaxf(v.y + b.y, 0.0f) * s.y + t.y;
      v.z = fmaxf(v.z + b.z, 0.0f) * s.z + t.z;
      v.w = fmaxf(v.w + b.w, 0.0f) * s.w + t.w;
      *reinterpret_cast<float4 *>(yp) = v;
    }
    return;
  }
  for (; i < n4; i += step) {
    const int64_t pix = i / c4;
    const int g = (int)(i - pix * c4);
    load_table(g);
    float4 v = x[i];
    v.x = fmaxf(v.x + b.x, 0.0f) * s.x + t.x;
    v.y = fmaxf(v.y + b.y, 0.0f) * s.y + t.y;
    v.z = fmaxf(v.z + b.z, 0.0f) * s.z + t.z;
    v.w = fmaxf(v.w + b.w, 0.0f) * s.w + t.w;
    *reinterpret_cast<float4 *>(y + pix * y_stride + (int64_t)g * 4) = v;
  }
}

// pillar -= data_mean (data/dataset.py:102-105): the optional per-element dataset mean of
// the [9,P,N] tensor (pillar_means.pkl, make_means.py), the same for every sweep of the batch
__global__ __launch_bounds__(256) void k_subtract_mean(float *__restrict__ x,
                                                       const float *__restrict__ mean, int64_t n,
                                                       bool vec) {
  float *xb = x + (int64_t)blockIdx.y * n;
  if (vec) {
    float4 *x4 = reinterpret_cast<float4 *>(xb);
    const float4 *m4 = reinterpret_cast<const float4 *>(mean);
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      float4 v = x4[i];
      const float4 m = m4[i];
      v.x -= m.x;
      v.y -= m.y;
      v.z -= m.z;
      v.w -= m.w;
      x4[i] = v;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
      xb[i] -= mean[i];
  }
}

// PPScatter.forward (model/model.py:53-62) on the feature net's output: out[b,:,row,col] =
// x[b,:,p] for the flagged pillars.  One wave moves a 64-channel x 64-pillar tile: coalesced
// 256-byte row reads of x[b][c][p0..p0+63], an LDS transpose, and for a channels-last canvas one
// 256-byte pixel per pillar (64 lanes = 64 channels); NCHW canvases get 4-byte scattered stores.
__global__ __launch_bounds__(256) void k_scatter_canvas(const float *__restrict__ x,
                                                        const long long *__restrict__ idx,
                                                        float *__restrict__ canvas, int C, int P,
                                                        int H, int W, int nhwc) {
  __shared__ float s_t[4][64][65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z;
  const int p0 = (blockIdx.x * 4 + wave) * 64;
  const int c0 = blockIdx.y * 64;
  if (p0 >= P) return;  // whole wave; no workgroup barrier below
  float(*t)[65] = s_t[wave];
  const int np = min(64, P - p0), nc = min(64, C - c0);
  // 16 row loads in flight at a time (one after the other they are a 64-step latency chain)
  const float *xr = x + ((int64_t)b * C + c0) * P + p0 + lane;
#pragma unroll 1
  for (int cb = 0; cb < nc; cb += 16) {
    float rr[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) rr[k] = (cb + k < nc && lane < np) ? xr[(int64_t)(cb + k) * P] : 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (cb + k < nc) t[cb + k][lane] = rr[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // every lane resolves ONE pillar's pixel (three 8-byte loads, all 64 in flight together:
  // read one after the other they were a 64-step latency chain, 50 us per launch)
  int64_t pix = -1;
  if (lane < np) {
    const long long *io = idx + ((int64_t)b * P + p0 + lane) * 3;
    const long long flag = io[0], col = io[1], row = io[2];  // {1, canvas_x, canvas_y}, pillars.cpp:390-392
    if (flag != 0 && row >= 0 && row < H && col >= 0 && col < W) pix = row * W + col;
  }
  for (int q = 0; q < np; ++q) {
    const int64_t pq = __shfl(pix, q, 64);  // wave-uniform
    if (pq < 0) continue;
    if (lane < nc) {
      const float v = t[lane][q];
      if (nhwc)
        canvas[((int64_t)b * H * W + pq) * C + c0 + lane] = v;
      else
        canvas[((int64_t)b * C + c0 + lane) * H * W + pq] = v;
    }
  }
}

}  // namespace pp

using namespace pp;

extern "C" int pp_scatter_canvas_dev(pp_ctx_t *ctx, void *stream_, const float *features_dev,
                                     const int64_t *indices_dev, int batch, int channels,
                                     int max_pillars, float *canvas_dev, int canvas_h, int canvas_w,
                                     int channels_last) {
  if (!ctx || !features_dev || !indices_dev || !canvas_dev) {
    set_error("pp_scatter_canvas_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  if (batch < 1 || batch > 65535 || channels < 1 || channels > 65535 * 64 || max_pillars < 1 ||
      canvas_h < 1 || canvas_w < 1) {
    set_error("pp_scatter_canvas_dev: bad sizes (batch=%d channels=%d P=%d canvas %dx%d)", batch,
              channels, max_pillars, canvas_h, canvas_w);
    return PP_ERR_VALUE;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != ctx->device) (void)hipSetDevice(ctx->device);
  hipStream_t st = static_cast<hipStream_t>(stream_);
  int rc = PP_OK;
  hipError_t e = hipMemsetAsync(canvas_dev, 0,
                                (size_t)batch * channels * canvas_h * canvas_w * sizeof(float), st);
  if (e == hipSuccess) {
    const dim3 grid((unsigned)((max_pillars + 255) / 256), (unsigned)((channels + 63) / 64), (unsigned)batch);
    hipLaunchKernelGGL(k_scatter_canvas, grid, dim3(256), 0, st, features_dev,
                       reinterpret_cast<const long long *>(indices_dev), canvas_dev, channels, max_pillars,
                       canvas_h, canvas_w, channels_last ? 1 : 0);
    e = hipGetLastError();
  }
  if (prev >= 0 && prev != ctx->device) (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    set_error("pp_scatter_canvas_dev failed: %s", hipGetErrorString(e));
    rc = PP_ERR_HIP;
  }
  return rc;
}

extern "C" int pp_subtract_mean_dev(pp_ctx_t *ctx, void *stream_, float *pillars_dev, int batch,
                                    int64_t elems_per_sweep, const float *mean_dev) {
  if (!ctx || !pillars_dev || !mean_dev) {
    set_error("pp_subtract_mean_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  if (batch < 1 || batch > 65535 || elems_per_sweep < 1) {
    set_error("pp_subtract_mean_dev: bad sizes (batch=%d elems=%lld)", batch, (long long)elems_per_sweep);
    return PP_ERR_VALUE;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != ctx->device) (void)hipSetDevice(ctx->device);
  const bool vec = (elems_per_sweep % 4 == 0) &&
                   (((reinterpret_cast<uintptr_t>(pillars_dev) | reinterpret_cast<uintptr_t>(mean_dev)) & 15) == 0);
  const int64_t work = vec ? elems_per_sweep / 4 : elems_per_sweep;
  const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((work + 255) / 256, 4096));
  hipLaunchKernelGGL(k_subtract_mean, dim3(gx, (unsigned)batch), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), pillars_dev, mean_dev, elems_per_sweep, vec);
  hipError_t e = hipGetLastError();
  if (prev >= 0 && prev != ctx->device) (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    set_error("k_subtract_mean launch failed: %s", hipGetErrorString(e));
    return PP_ERR_HIP;
  }
  return PP_OK;
}

extern "C" int pp_bias_relu_bn_nhwc_dev(pp_ctx_t *ctx, void *stream_, float *x_dev, int64_t pixels,
                                        int channels, const float *params_dev, float *y_dev,
                                        int64_t y_channels, int64_t y_channel_offset) {
  if (!y_dev) {  // in place
    y_dev = x_dev;
    y_channels = channels;
    y_channel_offset = 0;
  }
  if (!ctx || !x_dev || !params_dev) {
    set_error("pp_bias_relu_bn_nhwc_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  if (pixels < 1 || channels < 4 || (channels & 3) || (y_channels & 3) || (y_channel_offset & 3) ||
      y_channel_offset < 0 || y_channel_offset + channels > y_channels ||
      ((reinterpret_cast<uintptr_t>(x_dev) | reinterpret_cast<uintptr_t>(y_dev)) & 15)) {
    set_error("pp_bias_relu_bn_nhwc_dev: need channels, y_channels, y_channel_offset multiples of 4, "
              "the slice inside y, 16-byte aligned tensors (pixels=%lld channels=%d y_channels=%lld "
              "offset=%lld)", (long long)pixels, channels, (long long)y_channels,
              (long long)y_channel_offset);
    return PP_ERR_VALUE;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != ctx->device) (void)hipSetDevice(ctx->device);
  const int c4 = channels / 4;
  const int64_t n4 = pixels * c4;
  // 256 CUs x 8 workgroups in flight; a grid stride that is a multiple of 256 keeps the lane's
  // channel group fixed whenever c4 divides 256
  const unsigned blocks = (unsigned)std::min<int64_t>((n4 + 255) / 256, 2048);
  float *y = y_dev + y_channel_offset;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (256 % c4 == 0)
    hipLaunchKernelGGL(k_bias_relu_bn_nhwc<true>, dim3(blocks), dim3(256), 0, st,
                       reinterpret_cast<const float4 *>(x_dev), y, c4, n4, y_channels, params_dev);
  else
    hipLaunchKernelGGL(k_bias_relu_bn_nhwc<false>, dim3(blocks), dim3(256), 0, st,
                       reinterpret_cast<const float4 *>(x_dev), y, c4, n4, y_channels, params_dev);
  hipError_t e = hipGetLastError();
  if (prev >= 0 && prev != ctx->device) (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    set_error("k_bias_relu_bn_nhwc launch failed: %s", hipGetErrorString(e));
    return PP_ERR_HIP;
  }
  return PP_OK;
}

extern "C" int pp_bias_relu_bn_dev(pp_ctx_t *ctx, void *stream_, float *x_dev, int64_t batch,
                                   int channels, int64_t hw, const float *params_dev,
                                   float *y_dev, int64_t y_channels, int64_t y_channel_offset) {
  if (!y_dev) {  // in place
    y_dev = x_dev;
    y_channels = channels;
    y_channel_offset = 0;
  }
  if (y_channels < channels || y_channel_offset < 0 || y_channel_offset + channels > y_channels) {
    set_error("pp_bias_relu_bn_dev: channel slice [%lld,%lld) outside [0,%lld)",
              (long long)y_channel_offset, (long long)(y_channel_offset + channels),
              (long long)y_channels);
    return PP_ERR_VALUE;
  }
  if (!ctx || !x_dev || !params_dev) {
    set_error("pp_bias_relu_bn_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  if (batch < 1 || channels < 1 || hw < 1 || batch * channels > 65535) {
    set_error("pp_bias_relu_bn_dev: need batch*channels in [1,65535], hw >= 1");
    return PP_ERR_VALUE;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != ctx->device) (void)hipSetDevice(ctx->device);
  const unsigned gx = (unsigned)std::min<int64_t>(((hw >> 2) + 255) / 256 + 1, 64);
  hipLaunchKernelGGL(k_bias_relu_bn, dim3(gx, (unsigned)(batch * channels)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), x_dev, y_dev + y_channel_offset * hw, channels,
                     hw, y_channels, params_dev);
  hipError_t e = hipGetLastError();
  if (prev >= 0 && prev != ctx->device) (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    set_error("k_bias_relu_bn launch failed: %s", hipGetErrorString(e));
    return PP_ERR_HIP;
  }
  return PP_OK;
}
