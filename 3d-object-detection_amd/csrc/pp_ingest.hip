// pp_ingest.hip -- lidar sweep ingest pre-pass.
//
// Replaces the per-sweep point preparation of PPDataset.__getitem__
// (/root/reference data/dataset.py:65-82): LidarPointCloud.from_file (f32 rows,
// first four columns x,y,z,intensity), LidarPointCloud.transform (4x4 rigid
// transform applied in f64, stored back as f32), LidarPointCloud.remove_close
// (drop points with |x| < r AND |y| < r) and the hstack aggregation of sweeps.
// The lyft_dataset_sdk point-cloud class is an absent third-party dependency;
// its published behaviour is restated (recalled).
//
// Removed points are not compacted away: their x is set to NaN, which the
// voxelizer's half-open range test drops (pp_voxelize.hip: point_cell), so the
// surviving points keep their input order and no scan is needed.

#include "pp_common.h"

namespace pp {

struct Xform {
  double m[12];  // rows 0..2 of the 4x4 matrix, row-major
};

__global__ __launch_bounds__(256) void k_ingest(const float *__restrict__ raw, int64_t n,
                                                int raw_cols, Xform t, double radius,
                                                float4 *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float *p = raw + i * raw_cols;
  const double x = p[0], y = p[1], z = p[2];
  // transf_matrix.dot([x, y, z, 1]) in f64, rounded to f32 once (the SDK stores the
  // result back into its float32 point array)
  const double tx = ((t.m[0] * x + t.m[1] * y) + t.m[2] * z) + t.m[3];
  const double ty = ((t.m[4] * x + t.m[5] * y) + t.m[6] * z) + t.m[7];
  const double tz = ((t.m[8] * x + t.m[9] * y) + t.m[10] * z) + t.m[11];
  float4 o = make_float4((float)tx, (float)ty, (float)tz, p[3]);
  // remove_close: both |x| and |y| (of the stored f32 values) inside the radius
  if (fabsf(o.x) < (float)radius && fabsf(o.y) < (float)radius) o.x = __int_as_float(0x7FC00000);
  out[i] = o;
}

// All sweeps of a sample in one launch (dataset.py:65-82 loops over num_sweeps files): the
// launch's points are the sweeps back to back, a thread finds its sweep in the prefix table.
constexpr int kMaxSweeps = PP_MAX_INGEST_SWEEPS;
struct SweepBatch {
  const float *raw[kMaxSweeps];
  int64_t first[kMaxSweeps + 1];  // output row of each sweep's first point
  Xform t[kMaxSweeps];
  int nsweeps, raw_cols;
};

__global__ __launch_bounds__(256) void k_ingest_sweeps(SweepBatch sb, double radius, float4 *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= sb.first[sb.nsweeps]) return;
  int s = 0;
#pragma unroll 1
  while (s + 1 < sb.nsweeps && i >= sb.first[s + 1]) ++s;
  const float *p = sb.raw[s] + (i - sb.first[s]) * sb.raw_cols;
  const double *m = sb.t[s].m;
  const double x = p[0], y = p[1], z = p[2];
  const double tx = ((m[0] * x + m[1] * y) + m[2] * z) + m[3];  // as in k_ingest
  const double ty = ((m[4] * x + m[5] * y) + m[6] * z) + m[7];
  const double tz = ((m[8] * x + m[9] * y) + m[10] * z) + m[11];
  float4 o = make_float4((float)tx, (float)ty, (float)tz, p[3]);
  if (fabsf(o.x) < (float)radius && fabsf(o.y) < (float)radius) o.x = __int_as_float(0x7FC00000);
  out[i] = o;
}

}  // namespace pp

using namespace pp;

extern "C" int pp_ingest_sweeps_dev(pp_ctx_t *ctx, void *stream_, int32_t n_sweeps, const float *const *raw_dev,
                                    const int64_t *n_points, int raw_cols, const double *transforms_rowmajor4x4,
                                    double min_dist, float *points_out_dev) {
  if (!ctx || !raw_dev || !n_points || !transforms_rowmajor4x4 || !points_out_dev) {
    set_error("pp_ingest_sweeps_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  if (n_sweeps < 1 || n_sweeps > kMaxSweeps || raw_cols < 4 || raw_cols > 64) {
    set_error("pp_ingest_sweeps_dev: need 1 <= n_sweeps <= %d and 4 <= raw_cols <= 64", kMaxSweeps);
    return PP_ERR_VALUE;
  }
  if (reinterpret_cast<uintptr_t>(points_out_dev) & 15) {
    set_error("pp_ingest_sweeps_dev: output must be 16-byte aligned");
    return PP_ERR_VALUE;
  }
  SweepBatch sb;
  sb.nsweeps = n_sweeps;
  sb.raw_cols = raw_cols;
  sb.first[0] = 0;
  for (int s = 0; s < n_sweeps; ++s) {
    if (n_points[s] < 0 || (n_points[s] > 0 && !raw_dev[s])) {
      set_error("pp_ingest_sweeps_dev: sweep %d: bad size or NULL rows", s);
      return PP_ERR_VALUE;
    }
    sb.raw[s] = raw_dev[s];
    sb.first[s + 1] = sb.first[s] + n_points[s];
    for (int k = 0; k < 12; ++k) sb.t[s].m[k] = transforms_rowmajor4x4[s * 16 + k];
  }
  for (int s = n_sweeps; s < kMaxSweeps; ++s) {
    sb.raw[s] = nullptr;
    sb.first[s + 1] = sb.first[n_sweeps];
  }
  const int64_t total = sb.first[n_sweeps];
  if (total == 0) return PP_OK;
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != ctx->device) (void)hipSetDevice(ctx->device);
  hipLaunchKernelGGL(k_ingest_sweeps, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), sb, min_dist, reinterpret_cast<float4 *>(points_out_dev));
  hipError_t e = hipGetLastError();
  if (prev >= 0 && prev != ctx->device) (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    set_error("k_ingest_sweeps launch failed: %s", hipGetErrorString(e));
    return PP_ERR_HIP;
  }
  return PP_OK;
}

extern "C" int pp_ingest_dev(pp_ctx_t *ctx, void *stream_, const float *raw_dev, int64_t n_points,
                             int raw_cols, const double *transform_rowmajor4x4, double min_dist,
                             float *points_out_dev) {
  if (!ctx || !transform_rowmajor4x4 || (n_points > 0 && (!raw_dev || !points_out_dev))) {
    set_error("pp_ingest_dev: NULL argument");
    return PP_ERR_VALUE;
  }
  if (n_points < 0 || raw_cols < 4 || raw_cols > 64) {
    set_error("pp_ingest_dev: need n_points >= 0 and 4 <= raw_cols <= 64");
    return PP_ERR_VALUE;
  }
  if (reinterpret_cast<uintptr_t>(points_out_dev) & 15) {
    set_error("pp_ingest_dev: output must be 16-byte aligned");
    return PP_ERR_VALUE;
  }
  if (n_points == 0) return PP_OK;
  Xform t;
  for (int k = 0; k < 12; ++k) t.m[k] = transform_rowmajor4x4[k];
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != ctx->device) (void)hipSetDevice(ctx->device);
  hipLaunchKernelGGL(k_ingest, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), raw_dev, n_points, raw_cols, t, min_dist,
                     reinterpret_cast<float4 *>(points_out_dev));
  hipError_t e = hipGetLastError();
  if (prev >= 0 && prev != ctx->device) (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    set_error("k_ingest launch failed: %s", hipGetErrorString(e));
    return PP_ERR_HIP;
  }
  return PP_OK;
}
