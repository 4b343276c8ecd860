// pp_pfn.hip -- PPFeatureNet (inference) on the DENSE pillar tensor.
//
// /root/reference model/model.py:31-40: y[b,c,p] = max_n BN_c(ReLU(bias_c + sum_d W[c,d]
// x[b,d,p,n])) over all N slots (zero-padded ones included).  PyTorch-ROCm runs it as a 1x1
// convolution that writes a [B,64,P,N] intermediate (307 MB per sweep at P=12000, N=100),
// ReLU, BatchNorm and a max reduction that reads it back: > 2 GB of HBM traffic for 173 MB
// of input.  Here one wave owns four consecutive pillars, stages their nine feature rows
// in LDS and every lane carries one output channel: the input is read once (coalesced
// rows), the intermediate never exists, the output is one 16-byte store per lane.
// HBM-bound: 4*9*P*N bytes in, 4*64*P bytes out per sweep.
//
// The arithmetic is the fused voxelizer's (pp_voxelize.hip, kModePfn: one fmaf chain z from
// the bias in feature order, r = ReLU(z), then s >= 0 ? s*max(r) + t : s*min(r) + t), so the
// two paths give bit-identical features.  Two exact rearrangements make it cheaper here:
// ReLU is monotone, so max_n ReLU(z_n) = ReLU(max_n z_n) and min likewise; and a lane whose
// BatchNorm scale is negative negates its weights and bias (fma(-w, x, -a) = -fma(w, x, a)
// exactly), so min_n z_n = -max_n(-z_n): ONE running max per lane, no per-point ReLU.  The
// chain runs on point pairs with packed f32 FMAs (v_pk_fma_f32, twice the scalar FMA rate):
// the kernel is VALU-bound (9 MACs x 64 channels per 4-byte input), not HBM-bound.

#include "pp_common.h"

namespace pp {

constexpr int kPfnC = 64;        // output channels = lanes
constexpr int kPfnWaves = 4;     // waves per workgroup
constexpr int kPfnKW = 4;        // pillars per wave
constexpr int kPfnChunk = 256;   // points staged per pass

__device__ __forceinline__ void wave_sync() {
  // LDS operations of one wave execute in program order; this only stops the compiler
  // from moving LDS accesses across a cross-lane hand-off.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

typedef float v2f __attribute__((ext_vector_type(2)));

struct PfnLane {
  float w[9], bias, scale, shift;  // w, bias already negated when scale < 0
};

// z for two points at once; x[d] = {point a, point b} of feature row d
__device__ __forceinline__ v2f pfn_pair(const PfnLane &A, const v2f x[9]) {
  v2f z = {A.bias, A.bias};
#pragma unroll
  for (int d = 0; d < 9; ++d) {
    const v2f w = {A.w[d], A.w[d]};
    z = __builtin_elementwise_fma(w, x[d], z);
  }
  return z;
}

__device__ __forceinline__ float pfn_point(const PfnLane &A, float x0, float x1, float x2, float x3,
                                           float x4, float x5, float x6, float x7, float x8) {
  float r = A.bias;
  r = fmaf(A.w[0], x0, r);
  r = fmaf(A.w[1], x1, r);
  r = fmaf(A.w[2], x2, r);
  r = fmaf(A.w[3], x3, r);
  r = fmaf(A.w[4], x4, r);
  r = fmaf(A.w[5], x5, r);
  r = fmaf(A.w[6], x6, r);
  r = fmaf(A.w[7], x7, r);
  r = fmaf(A.w[8], x8, r);
  return r;
}

// the lane's result from its running maximum of (sign-folded) z
__device__ __forceinline__ float pfn_result(const PfnLane &A, float m) {
  const float r = fmaxf(A.scale >= 0.0f ? m : -m, 0.0f);
  return fmaf(r, A.scale, A.shift);
}

// kVec: N % 4 == 0 and N <= kPfnChunk -- one float4 per lane and feature row covers a
// pillar, and the next pillar's rows are in flight while this one is reduced.
template <bool kVec>
__global__ __launch_bounds__(kPfnWaves * 64) void k_pfn_dense(const float *__restrict__ x,
                                                              const float *__restrict__ prm,
                                                              float *__restrict__ out, int P, int N) {
  __shared__ __attribute__((aligned(16))) float s_x[kPfnWaves][9][kPfnChunk];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int p0 = (blockIdx.x * kPfnWaves + wave) * kPfnKW;
  if (p0 >= P) return;  // whole wave; no workgroup barrier below
  const int kw = min(kPfnKW, P - p0);
  float(*sx)[kPfnChunk] = s_x[wave];
  const int64_t plane = (int64_t)P * N;
  const float *xb = x + (int64_t)b * 9 * plane;
  float yv[kPfnKW];

  auto load_lane = [&](int c) {
    PfnLane A;
    const float *q = prm + c * 12;
#pragma unroll
    for (int d = 0; d < 9; ++d) A.w[d] = q[d];
    A.bias = q[9];
    A.scale = q[10];
    A.shift = q[11];
    if (A.scale < 0.0f) {
#pragma unroll
      for (int d = 0; d < 9; ++d) A.w[d] = -A.w[d];
      A.bias = -A.bias;
    }
    return A;
  };

  if (kVec) {
    // A broadcast LDS read costs the CU's one LDS pipe as much as any other, and with one
    // channel per lane every float read feeds a single FMA: the LDS pipe, shared by the four
    // SIMDs, was the bound.  So a lane carries FOUR channels (cg, cg+16, cg+32, cg+48) for a
    // QUARTER of the points (float4 groups j = g, g+4, ...): a quarter of the LDS reads for
    // the same FMAs, and the four partial maxima meet in two cross-lane steps per pillar.
    const int cg = lane & 15, g = lane >> 4;
    PfnLane A[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) A[i] = load_lane(cg + 16 * i);
    const int nq = N >> 2;  // float4 per row, <= 64
    float4 r[9];
    auto fetch = [&](int k) {
      const float *row = xb + (int64_t)(p0 + k) * N;
#pragma unroll
      for (int d = 0; d < 9; ++d)
        r[d] = (lane < nq) ? reinterpret_cast<const float4 *>(row + d * plane)[lane]
                           : make_float4(0, 0, 0, 0);
    };
    fetch(0);
#pragma unroll
    for (int k = 0; k < kPfnKW; ++k) {
      if (k < kw) {
        if (lane < nq) {
#pragma unroll
          for (int d = 0; d < 9; ++d) reinterpret_cast<float4 *>(sx[d])[lane] = r[d];
        }
        wave_sync();
        if (k + 1 < kw) fetch(k + 1);
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int j = g; j < nq; j += 4) {
          v2f lo[9], hi[9];
#pragma unroll
          for (int d = 0; d < 9; ++d) {
            const float4 v = reinterpret_cast<const float4 *>(sx[d])[j];
            lo[d] = v2f{v.x, v.y};
            hi[d] = v2f{v.z, v.w};
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const v2f za = pfn_pair(A[i], lo), zb = pfn_pair(A[i], hi);
            m[i] = fmaxf(fmaxf(m[i], fmaxf(za.x, za.y)), fmaxf(zb.x, zb.y));
          }
        }
        float y = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float mi = fmaxf(m[i], __shfl_xor(m[i], 16, 64));
          mi = fmaxf(mi, __shfl_xor(mi, 32, 64));
          const float yi = pfn_result(A[i], mi);
          y = (i == g) ? yi : y;  // group g stores channel cg + 16*g
        }
        yv[k] = y;
        wave_sync();  // every lane is done reading before the next overwrite
      } else {
        yv[k] = 0.0f;
      }
    }
    float *o = out + ((int64_t)b * kPfnC + cg + 16 * g) * P + p0;
    if (kw == kPfnKW && (P & 3) == 0) {
      *reinterpret_cast<float4 *>(o) = make_float4(yv[0], yv[1], yv[2], yv[3]);
    } else {
#pragma unroll
      for (int k = 0; k < kPfnKW; ++k)
        if (k < kw) o[k] = yv[k];
    }
    return;
  } else {
    const PfnLane A = load_lane(lane);
#pragma unroll 1
    for (int k = 0; k < kPfnKW; ++k) {
      yv[k] = 0.0f;
      if (k >= kw) continue;
      const float *row = xb + (int64_t)(p0 + k) * N;
      float m = -INFINITY;
      for (int n0 = 0; n0 < N; n0 += kPfnChunk) {
        const int cn = min(kPfnChunk, N - n0);
        for (int i = lane; i < cn; i += 64) {
#pragma unroll
          for (int d = 0; d < 9; ++d) sx[d][i] = row[d * plane + n0 + i];
        }
        wave_sync();
        for (int j = 0; j < cn; ++j) {
          m = fmaxf(m, pfn_point(A, sx[0][j], sx[1][j], sx[2][j], sx[3][j], sx[4][j], sx[5][j],
                                 sx[6][j], sx[7][j], sx[8][j]));
        }
        wave_sync();
      }
      yv[k] = pfn_result(A, m);
    }
  }
  float *o = out + ((int64_t)b * kPfnC + lane) * P + p0;
  if (kw == kPfnKW && (P & 3) == 0) {
    *reinterpret_cast<float4 *>(o) = make_float4(yv[0], yv[1], yv[2], yv[3]);
  } else {
#pragma unroll
    for (int k = 0; k < kPfnKW; ++k)
      if (k < kw) o[k] = yv[k];
  }
}

}  // namespace pp

using namespace pp;

extern "C" int pp_pfn_dense_dev(pp_ctx_t *ctx, void *stream_, const float *pillars_dev, int batch,
                                int max_pillars, int max_points_per_pillar,
                                const float *pfn_params_dev, int channels, float *features_dev) {
  if (!ctx || !pillars_dev || !pfn_params_dev || !features_dev) {
    set_error("pp_pfn_dense_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  if (channels != kPfnC) {
    set_error("the feature-net kernel is built for %d output channels (got %d)", kPfnC, channels);
    return PP_ERR_VALUE;
  }
  const int P = max_pillars, N = max_points_per_pillar;
  if (batch < 1 || batch > 65535 || P < 1 || N < 1 || (int64_t)P * N > (1ll << 31)) {
    set_error("pp_pfn_dense_dev: bad sizes (batch=%d P=%d N=%d)", batch, P, N);
    return PP_ERR_VALUE;
  }
  if ((reinterpret_cast<uintptr_t>(pillars_dev) | reinterpret_cast<uintptr_t>(features_dev)) & 15) {
    set_error("pp_pfn_dense_dev: tensors must be 16-byte aligned");
    return PP_ERR_VALUE;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != ctx->device) (void)hipSetDevice(ctx->device);
  const dim3 grid((unsigned)((P + kPfnWaves * kPfnKW - 1) / (kPfnWaves * kPfnKW)), (unsigned)batch);
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if ((N & 3) == 0 && N <= kPfnChunk)
    hipLaunchKernelGGL(k_pfn_dense<true>, grid, dim3(kPfnWaves * 64), 0, st, pillars_dev,
                       pfn_params_dev, features_dev, P, N);
  else
    hipLaunchKernelGGL(k_pfn_dense<false>, grid, dim3(kPfnWaves * 64), 0, st, pillars_dev,
                       pfn_params_dev, features_dev, P, N);
  hipError_t e = hipGetLastError();
  if (prev >= 0 && prev != ctx->device) (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    set_error("k_pfn_dense launch failed: %s", hipGetErrorString(e));
    return PP_ERR_HIP;
  }
  return PP_OK;
}
