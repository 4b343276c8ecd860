// pp_pfn.hip -- PPFeatureNet (inference) on the DENSE pillar tensor.
//
// /root/reference model/model.py:31-40: y[b,c,p] = max_n BN_c(ReLU(bias_c + sum_d W[c,d]
// x[b,d,p,n])) over all N slots (zero-padded ones included).  PyTorch-ROCm runs it as a 1x1
// convolution that writes a [B,64,P,N] intermediate (307 MB per sweep at P=12000, N=100),
// ReLU, BatchNorm and a max reduction that reads it back: > 2 GB of HBM traffic for 173 MB
// of input.  Here one wave owns four consecutive pillars: the input is read once, the
// intermediate never exists, the output is one 16-byte store per lane
// (4*9*P*N bytes in, 4*64*P bytes out per sweep).
//
// The arithmetic is the fused voxelizer's (pp_voxelize.hip, kModePfn: one fmaf chain z from
// the bias in feature order, r = ReLU(z), then s >= 0 ? s*max(r) + t : s*min(r) + t), so the
// two paths give bit-identical features.  Two exact rearrangements make it cheaper here:
// ReLU is monotone, so max_n ReLU(z_n) = ReLU(max_n z_n) and min likewise; and a lane whose
// BatchNorm scale is negative negates its weights and bias (fma(-w, x, -a) = -fma(w, x, a)
// exactly), so min_n z_n = -max_n(-z_n): ONE running max per lane, no per-point ReLU.
// 9 MACs x 64 channels per 4-byte input make this compute-bound, not HBM-bound (reading the
// 173 MB of a 4-sweep step takes 42 us on this chip): as a VALU kernel (v_fma_f32 or
// v_pk_fma_f32 alike, one or four channels per lane) it ran at ~55 TF/s = 98 us per step; the
// f32 MFMA form below is the same fmaf chain bit for bit and measures 86 us -- its MFMAs
// (130 per wave, 41 us in total) reach 45 % of the pipe (profiles/r01/NOTES.md).

#include "pp_common.h"

namespace pp {

constexpr int kPfnC = 64;        // output channels = lanes
constexpr int kPfnWaves = 4;     // waves per workgroup
constexpr int kPfnKW = 4;        // pillars per wave
constexpr int kPfnChunk = 256;   // points staged per pass (generic kernel)
constexpr int kMfmaKW = 4;       // pillars per wave in the MFMA kernel (8 and 16 measured slower)
constexpr int kPfnDepth = 8;     // operand tiles in flight per wave (MFMA kernel)

__device__ __forceinline__ void wave_sync() {
  // LDS operations of one wave execute in program order; this only stops the compiler
  // from moving LDS accesses across a cross-lane hand-off.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct PfnLane {
  float w[9], bias, scale, shift;  // w, bias already negated when scale < 0
};

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

typedef float f32x16 __attribute__((ext_vector_type(16)));

// N % 4 == 0: the feature net as a GEMM on the f32 matrix cores.  The wave's kw*N points are
// the rows (32 per tile: consecutive pillars are contiguous in every feature row, so a tile
// may straddle pillars), the 64 channels two column tiles, K = the nine features + one zero
// term in front: v_mfma_f32_32x32x2_f32 computes D = fma(a_k1, b_k1, fma(a_k0, b_k0, C))
// -- bit for bit the fmaf chain above with C = bias.  Operands come straight from global
// memory: lane l holds A[row l&31][k = l>>5], i.e. one float of one feature row per k-step,
// 32 consecutive floats per half-wave; no LDS.  amdgpu_waves_per_eu(2) caps the wave at 256
// registers so that the compiler takes the MFMA form with accumulators in arch VGPRs (the
// maxima read them directly; the AGPR form cost 32 v_accvgpr_read per tile).  The accumulator has the channel on the lane
// and 16 rows (four groups of four consecutive points) in registers: N % 4 == 0 puts every
// group inside one pillar, so the running maxima are in-lane, and the two half-waves meet
// once at the end.  Rows past the wave's last point repeat that point (no effect on a max).
__global__ __launch_bounds__(kPfnWaves * 64) __attribute__((amdgpu_waves_per_eu(2))) void k_pfn_dense_mfma(
    const float *__restrict__ x, const float *__restrict__ prm, float *__restrict__ out, int P, int N) {
  // everything that steers control flow is wave-uniform: say so (readfirstlane), or the
  // compiler predicates the whole tile loop with exec masks -- that junk, not the MFMAs,
  // was 3/4 of the first version's issue slots
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y;
  const int p0 = (blockIdx.x * kPfnWaves + wave) * kMfmaKW;
  if (p0 >= P) return;
  const int kw = min(kMfmaKW, P - p0);
  const int r = lane & 31, h = lane >> 5;
  const int64_t plane = (int64_t)P * N;
  const int npts = kw * N;
  const int tiles = (npts + 31) >> 5;

  // B operand (weights, sign-folded) and C (bias) for the two channel tiles
  float bw[2][5], cb[2], sc[2], sh[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const float *q = prm + (r + 32 * nt) * 12;
    sc[nt] = q[10];
    sh[nt] = q[11];
    const float sign = sc[nt] < 0.0f ? -1.0f : 1.0f;
    cb[nt] = sign * q[9];
#pragma unroll
    for (int st = 0; st < 5; ++st) {
      const int f = 2 * st + h - 1;  // k-steps: [zero, d0] [d1, d2] [d3, d4] [d5, d6] [d7, d8]
      bw[nt][st] = (f >= 0) ? sign * q[f] : 0.0f;
    }
  }
  // A operand through a raw buffer: byte offset of (feature f, point q) = f*plane*4 + (p0*N+q)*4.
  // The zero term (f = -1) gets a negative offset: out of range, and the buffer returns 0.
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(x + (int64_t)b * 9 * plane), 0, (int)(36u * (unsigned)P * (unsigned)N), 0x00020000);
  unsigned foff[5];
#pragma unroll
  for (int st = 0; st < 5; ++st)
    foff[st] = (unsigned)(((int64_t)(2 * st + h - 1) * plane + (int64_t)p0 * N) * 4);
  auto load_tile = [&](int t, float a[5]) {
    const unsigned q4 = (unsigned)min(t * 32 + r, npts - 1) * 4u;
#pragma unroll
    for (int st = 0; st < 5; ++st)
      a[st] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, foff[st] + q4, 0, 0));
  };
  // which pillar a row of the wave belongs to (rows past the last point repeat it, so they
  // count for the last pillar)
  auto pillar_of = [&](int row) {
    int pid = 0;
#pragma unroll
    for (int k = 1; k < kMfmaKW; ++k) pid += (row >= k * N) ? 1 : 0;
    return min(pid, kw - 1);
  };
  float m[2][kMfmaKW];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int k = 0; k < kMfmaKW; ++k) m[nt][k] = -INFINITY;

  // kPfnDepth tiles of operands in flight per wave: a tile is 10 MFMAs (~0.3 us), far less
  // than an HBM round trip
  float a[kPfnDepth][5];
#pragma unroll
  for (int i = 0; i < kPfnDepth; ++i)
    if (i < tiles) load_tile(i, a[i]);
  for (int t0 = 0; t0 < tiles; t0 += kPfnDepth) {
#pragma unroll
    for (int i = 0; i < kPfnDepth; ++i) {
      const int t = t0 + i;
      if (t >= tiles) break;  // wave-uniform
      f32x16 acc[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[nt][v] = cb[nt];
#pragma unroll
        for (int st = 0; st < 5; ++st)
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][st], bw[nt][st], acc[nt], 0, 0, 0);
      }
      if (t + kPfnDepth < tiles) load_tile(t + kPfnDepth, a[i]);
      const int k_lo = pillar_of(t * 32), k_hi = pillar_of(t * 32 + 31);
      if (k_lo == k_hi) {
        // the common case (10 of 13 tiles at N = 100): all 16 registers feed one maximum
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          float g = acc[nt][0];
#pragma unroll
          for (int v = 1; v < 16; ++v) g = fmaxf(g, acc[nt][v]);
#pragma unroll
          for (int k = 0; k < kMfmaKW; ++k)
            if (k_lo == k) m[nt][k] = fmaxf(m[nt][k], g);  // k_lo is wave-uniform
        }
      } else {
#pragma unroll
        for (int vg = 0; vg < 4; ++vg) {
          // rows vg*8 + h*4 .. +3 of the tile: inside one pillar because N % 4 == 0
          const int pid = pillar_of(t * 32 + vg * 8 + h * 4);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const float g = fmaxf(fmaxf(acc[nt][4 * vg], acc[nt][4 * vg + 1]),
                                  fmaxf(acc[nt][4 * vg + 2], acc[nt][4 * vg + 3]));
#pragma unroll
            for (int k = 0; k < kMfmaKW; ++k) m[nt][k] = (pid == k) ? fmaxf(m[nt][k], g) : m[nt][k];
          }
        }
      }
    }
  }
  // half h stores channel tile h
  float yv[kMfmaKW];
#pragma unroll
  for (int k = 0; k < kMfmaKW; ++k) {
    float y[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float mm = fmaxf(m[nt][k], __shfl_xor(m[nt][k], 32, 64));
      const float rr = fmaxf(sc[nt] >= 0.0f ? mm : -mm, 0.0f);
      y[nt] = fmaf(rr, sc[nt], sh[nt]);
    }
    yv[k] = h ? y[1] : y[0];
  }
  float *o = out + ((int64_t)b * kPfnC + r + 32 * h) * P + p0;
  if (kw == kMfmaKW && (P & 3) == 0) {
#pragma unroll
    for (int k = 0; k < kMfmaKW; k += 4)
      *reinterpret_cast<float4 *>(o + k) = make_float4(yv[k], yv[k + 1], yv[k + 2], yv[k + 3]);
  } else {
#pragma unroll
    for (int k = 0; k < kMfmaKW; ++k)
      if (k < kw) o[k] = yv[k];
  }
}

// Any N: lanes = channels, feature rows staged in LDS, scalar fmaf chain.
__global__ __launch_bounds__(kPfnWaves * 64) void k_pfn_dense(const float *__restrict__ x,
                                                              const float *__restrict__ prm,
                                                              float *__restrict__ out, int P, int N) {
  __shared__ float s_x[kPfnWaves][9][kPfnChunk];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int p0 = (blockIdx.x * kPfnWaves + wave) * kPfnKW;
  if (p0 >= P) return;  // whole wave; no workgroup barrier below
  const int kw = min(kPfnKW, P - p0);
  float(*sx)[kPfnChunk] = s_x[wave];
  const int64_t plane = (int64_t)P * N;
  const float *xb = x + (int64_t)b * 9 * plane;
  PfnLane A;
  {
    const float *q = prm + lane * 12;
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
  }
#pragma unroll 1
  for (int k = 0; k < kw; ++k) {
    const float *row = xb + (int64_t)(p0 + k) * N;
    float m = -INFINITY;
    for (int n0 = 0; n0 < N; n0 += kPfnChunk) {
      const int cn = min(kPfnChunk, N - n0);
      for (int i = lane; i < cn; i += 64) {
#pragma unroll
        for (int d = 0; d < 9; ++d) sx[d][i] = row[d * plane + n0 + i];
      }
      wave_sync();
      for (int j = 0; j < cn; ++j)
        m = fmaxf(m, pfn_point(A, sx[0][j], sx[1][j], sx[2][j], sx[3][j], sx[4][j], sx[5][j],
                               sx[6][j], sx[7][j], sx[8][j]));
      wave_sync();
    }
    out[((int64_t)b * kPfnC + lane) * P + p0 + k] = pfn_result(A, m);
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
  const dim3 grid_mfma((unsigned)((P + kPfnWaves * kMfmaKW - 1) / (kPfnWaves * kMfmaKW)), (unsigned)batch);
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if ((N & 3) == 0 && 36ll * P * N < (1ll << 31))  // the MFMA kernel's buffer descriptor is 32-bit
    hipLaunchKernelGGL(k_pfn_dense_mfma, grid_mfma, dim3(kPfnWaves * 64), 0, st, pillars_dev,
                       pfn_params_dev, features_dev, P, N);
  else
    hipLaunchKernelGGL(k_pfn_dense, grid, dim3(kPfnWaves * 64), 0, st, pillars_dev, pfn_params_dev,
                       features_dev, P, N);
  hipError_t e = hipGetLastError();
  if (prev >= 0 && prev != ctx->device) (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    set_error("k_pfn_dense launch failed: %s", hipGetErrorString(e));
    return PP_ERR_HIP;
  }
  return PP_OK;
}
