// vox_lab -- development harness for the voxelizer kernels (not product, not a test).
// Links a variant build of pp_runtime.hip + pp_voxelize.hip; times the device entry point
// with HIP events and, in a -DPP_STAMPS build, prints k_tile's in-kernel phase stamps.
//   vox_lab <points.bin (f32 [B][n][4])> <B> <n> <half> <step> <P> <N> [iters] [order]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pp_hip.h"

extern "C" int pp_debug_stamps(pp_ctx_t *ctx, unsigned long long *host, int cap) __attribute__((weak));

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                   \
      return 1;                                                                \
    }                                                                          \
  } while (0)

int main(int argc, char **argv) {
  if (argc < 8) {
    fprintf(stderr, "usage: vox_lab pts.bin B n half step P N [iters] [order]\n");
    return 2;
  }
  const char *path = argv[1];
  const int B = atoi(argv[2]), n = atoi(argv[3]);
  const double half = atof(argv[4]), step = atof(argv[5]);
  const int P = atoi(argv[6]), N = atoi(argv[7]);
  const int iters = argc > 8 ? atoi(argv[8]) : 200;
  const int order = argc > 9 ? atoi(argv[9]) : 0;
  std::vector<float> pts((size_t)B * n * 4);
  FILE *f = fopen(path, "rb");
  if (!f || fread(pts.data(), 4, pts.size(), f) != pts.size()) {
    fprintf(stderr, "cannot read %zu floats from %s\n", pts.size(), path);
    return 1;
  }
  fclose(f);
  pp_ctx_t *ctx = nullptr;
  if (pp_ctx_create(0, &ctx)) {
    fprintf(stderr, "%s\n", pp_last_error());
    return 1;
  }
  pp_voxel_params_t prm{};
  prm.max_points_per_pillar = N;
  prm.max_pillars = P;
  prm.x_step = prm.y_step = step;
  prm.x_min = prm.y_min = -half;
  prm.x_max = prm.y_max = half;
  prm.z_min = -10;
  prm.z_max = 10;
  prm.canvas_height = 2 * half / step;
  prm.order = order;
  float *dpts, *dout;
  int64_t *didx;
  int32_t *dcnt;
  CK(hipMalloc(&dpts, pts.size() * 4));
  CK(hipMalloc(&dout, (size_t)B * 9 * P * N * 4));
  CK(hipMalloc(&didx, (size_t)B * P * 3 * 8));
  CK(hipMalloc(&dcnt, (size_t)B * 8));
  CK(hipMemcpy(dpts, pts.data(), pts.size() * 4, hipMemcpyHostToDevice));
  std::vector<int32_t> np(B, n);
  hipStream_t s;
  CK(hipStreamCreate(&s));
  for (int i = 0; i < 20; ++i)
    if (pp_voxelize_dev(ctx, s, dpts, n, np.data(), B, &prm, dout, didx, dcnt)) {
      fprintf(stderr, "%s\n", pp_last_error());
      return 1;
    }
  CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) pp_voxelize_dev(ctx, s, dpts, n, np.data(), B, &prm, dout, didx, dcnt);
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<int32_t> cnt(B * 2);
  CK(hipMemcpy(cnt.data(), dcnt, B * 8, hipMemcpyDeviceToHost));
  const double bytes = (double)B * (16.0 * n + 36.0 * P * N + 24.0 * P);
  const double us = ms * 1e3 / iters;
  printf("B=%d n=%d P=%d N=%d grid=%g/%g: %.2f us/step  %.0f sweeps/s  pipeline %.2f TB/s (%.1f%% of 8)  cells=%d pts=%d\n",
         B, n, P, N, half, step, us, B / (us * 1e-6), bytes / (us * 1e-6) / 1e12,
         bytes / (us * 1e-6) / 8e12 * 100, cnt[0], cnt[1]);
  if (pp_debug_stamps) {
    std::vector<unsigned long long> st(1 << 20);
    const int got = pp_debug_stamps(ctx, st.data(), (int)st.size());
    // rows of 8 stamps (100 MHz ticks) per wave: relative to the earliest first stamp
    if (got > 0) {
      const int rows = got / 8;
      unsigned long long t0 = ~0ull;
      for (int r = 0; r < rows; ++r)
        if (st[r * 8]) t0 = std::min(t0, st[r * 8]);
      double acc[8] = {0};
      double mx[8] = {0};
      int nk[8] = {0};
      int cntr = 0;
      for (int r = 0; r < rows; ++r) {
        if (!st[r * 8]) continue;
        ++cntr;
        for (int k = 0; k < 8; ++k) {
          if (st[r * 8 + k] < t0) continue;  // phase not reached by this wave
          const double v = (double)(st[r * 8 + k] - t0) * 0.01;
          acc[k] += v;
          mx[k] = std::max(mx[k], v);
          ++nk[k];
        }
      }
      printf("stamps (us since first wave start; mean / max over %d waves):\n", cntr);
      for (int k = 0; k < 8; ++k) printf("  phase %d: %.2f / %.2f  (%d waves)\n", k, acc[k] / std::max(nk[k], 1), mx[k], nk[k]);
    }
  }
  pp_ctx_destroy(ctx);
  return 0;
}
