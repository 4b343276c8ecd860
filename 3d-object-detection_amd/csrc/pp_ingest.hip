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

}  // namespace pp

using namespace pp;

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
