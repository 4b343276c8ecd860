// tools/lab/fill_pattern.cpp: store-only kernels at the C5 output shape (B x 9 x P x N f32) -- a linear fill
// against k_emit's pattern (a wave writes its 4 pillars' 1600 bytes in each of the 9 feature planes).
// build: hipcc -O3 --offload-arch=gfx950 tools/lab/fill_pattern.cpp -o tools/lab/_build/fill_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_linear(float4 *out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = make_float4(0, 0, 0, 0);
}

// blocked: workgroup b streams one contiguous range, 4 KB per trip
__global__ __launch_bounds__(256) void k_blocked(float4 *out, size_t n4) {
  const size_t per = (n4 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) out[i] = make_float4(0, 0, 0, 0);
}
// blocked, U stores per thread in flight per trip
template <int U>
__global__ __launch_bounds__(256) void k_blocked_u(float4 *out, size_t n4) {
  const size_t per = (n4 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256 * U) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * 256 < hi) out[i + u * 256] = make_float4(0, 0, 0, 0);
  }
}

// KW pillars per wave, N floats each: the wave's slab in plane d starts at ((b*9 + d)*P + p0)*N
typedef unsigned v4u __attribute__((__vector_size__(16)));
// the same with the buffer-store cache policy bits: 0 plain, 1 sc0, 2 nt, 16 sc1, 3 sc0 nt, 18 sc1 nt
template <int AUX>
__global__ __launch_bounds__(256) void k_pattern_aux(float *out, int P, int N) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
  const int p0 = (blockIdx.x * 4 + w) * 4;
  if (p0 >= P) return;
  const int n4 = 4 * N / 4;
  const v4u z = {0, 0, 0, 0};
  for (int d = 0; d < 9; ++d) {
    float *dst = out + (((size_t)b * 9 + d) * P + p0) * N;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, n4 * 16, 0x00020000);
    for (int i = lane; i < n4; i += 64) __builtin_amdgcn_raw_buffer_store_b128(z, rs, i * 16, 0, AUX);
  }
}

template <int KW>
__global__ __launch_bounds__(256) void k_pattern(float *out, int P, int N, int delay) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.y;
  const int p0 = (blockIdx.x * 4 + w) * KW;
  if (p0 >= P) return;
  if (delay) {  // a dependent load chain in front of the stores, like k_emit's prologue
    volatile float *q = out;
    float s = 0;
    for (int k = 0; k < delay; ++k) s += q[(size_t)(lane + k * 64 + (int)s) & 1023];
    if (s == 12345.f) out[0] = s;
  }
  const int n4 = KW * N / 4;
  for (int d = 0; d < 9; ++d) {
    float4 *dst = reinterpret_cast<float4 *>(out + (((size_t)b * 9 + d) * P + p0) * N);
    for (int i = lane; i < n4; i += 64) dst[i] = make_float4(0, 0, 0, 0);
  }
}

// k_emit's pattern from a PERSISTENT grid: workgroup w writes the slabs of pillar groups w, w + nwg, ... (a wave = 4
// pillars x 9 planes per trip): fewer concurrent writers, a narrower write window per plane
__global__ __launch_bounds__(256) void k_pattern_persistent(float *out, int P, int N, int B) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int groups = (P + 15) / 16;
  const v4u z = {0, 0, 0, 0};
  for (int g = blockIdx.x; g < groups * B; g += gridDim.x) {
    const int b = g / groups, p0 = ((g - b * groups) * 4 + w) * 4;
    if (p0 >= P) continue;
    const int n4 = 4 * N / 4;
    for (int d = 0; d < 9; ++d) {
      float *dst = out + (((size_t)b * 9 + d) * P + p0) * N;
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, n4 * 16, 0x00020000);
      for (int i = lane; i < n4; i += 64) __builtin_amdgcn_raw_buffer_store_b128(z, rs, i * 16, 0, 0);
    }
  }
}
// plane-major persistent: the grid walks plane 0 of all pillars, then plane 1, ...: ONE write window
__global__ __launch_bounds__(256) void k_plane_major(float *out, int P, int N, int B, int chunk_pillars) {
  const int chunks = (P + chunk_pillars - 1) / chunk_pillars;
  const long total = (long)B * 9 * chunks;
  for (long c = blockIdx.x; c < total; c += gridDim.x) {
    const int ch = (int)(c % chunks), d = (int)((c / chunks) % 9), b = (int)(c / ((long)chunks * 9));
    const int p0 = ch * chunk_pillars, np = min(chunk_pillars, P - p0);
    float4 *dst = reinterpret_cast<float4 *>(out + (((size_t)b * 9 + d) * P + p0) * N);
    const int n4 = np * N / 4;
    for (int i = threadIdx.x; i < n4; i += 256) dst[i] = make_float4(0, 0, 0, 0);
  }
}


// TWO ROLES in one launch: `zwg` workgroups stream zeros linearly over the whole output, skipping every 128-byte
// line that holds a pillar's head group (nh 16-byte groups at the start of each pillar's row); the other workgroups
// are k_emit-shaped (a wave = 4 pillars) and write only those live lines, whole, behind `delay` dependent loads.
// Every line is written exactly once.  Is "a linear fill + a sparse live pass" faster than k_emit's dense pattern?
__global__ __launch_bounds__(256) void k_two_roles(float *out, int P, int N, int B, int zwg, int nh, int delay) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const v4u z = {0, 0, 0, 0};
  const unsigned n4row = (unsigned)N / 4u;                 // 16-byte groups per pillar row
  const size_t gplane = (size_t)P * n4row;                 // groups per plane (a multiple of 8: planes are line-aligned)
  if ((int)blockIdx.x < zwg) {
    // 32-bit arithmetic only: plane by plane, the grid striding over the plane's groups
    const unsigned gp = (unsigned)gplane, magic = 0xFFFFFFFFu / n4row + 1u;
    float4 *o4 = reinterpret_cast<float4 *>(out);
    for (int pl = 0; pl < B * 9; ++pl) {
      float4 *op = o4 + (size_t)pl * gplane;
      for (unsigned g = blockIdx.x * 256u + threadIdx.x; g < gp; g += (unsigned)zwg * 256u) {
        const unsigned p = __umulhi(g, magic), n4 = g - p * n4row;
        const bool head = n4 < (unsigned)nh;               // (a real kernel reads the pillar's count here)
        const unsigned long long m = __ballot(head);
        const bool live_line = ((m >> (lane & ~7)) & 0xFFull) != 0;   // 8 consecutive lanes = one line
        if (!live_line) op[g] = make_float4(0, 0, 0, 0);
      }
    }
    return;
  }
  if (delay < 0) return;  // zero role alone
  const int id = (int)blockIdx.x - zwg, groups = (P + 15) / 16;
  const int b = id / groups, p0 = ((id - b * groups) * 4 + w) * 4;
  if (p0 >= P || b >= B) return;
  if (delay) {
    volatile float *q = out;
    float s = 0;
    for (int k = 0; k < delay; ++k) s += q[(size_t)(lane + k * 64 + (int)s) & 1023];
    if (s == 12345.f) out[0] = s;
  }
  // the live lines of the wave's 4 pillars in the 9 planes: lines that hold groups [p*n4row, p*n4row + nh)
  for (int t = lane; t < 9 * 4 * 8 * 2; t += 64) {         // (plane, pillar, up to 2 lines, 8 groups per line)
    const int gi = t & 7, li = (t >> 3) & 1, k = (t >> 4) & 3, d = t >> 6;
    const int p = p0 + k;
    if (p >= P) continue;
    const size_t g0 = (size_t)p * n4row, l0 = g0 >> 3, l1 = (g0 + nh - 1) >> 3;
    const size_t line = l0 + li;
    if (line > l1) continue;
    // a line shared with the previous pillar's head is written by that pillar's lanes as well: identical zeros
    float4 *o4 = reinterpret_cast<float4 *>(out) + ((size_t)b * 9 + d) * gplane;
    o4[line * 8 + gi] = make_float4(0, 0, 0, 0);
  }
  (void)z;
}

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4, P = argc > 2 ? atoi(argv[2]) : 30000, N = argc > 4 ? atoi(argv[4]) : 100;
  const bool brief = argc > 5 && atoi(argv[5]);  // only the linear fill and k_emit's pattern
  const size_t n = (size_t)B * 9 * P * N;
  float *out;
  CK(hipMalloc(&out, n * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const bool cold = argc > 3 && atoi(argv[3]);  // 1 GiB of other traffic between the launches
  float *junk = nullptr, *junk2 = nullptr;
  const size_t jn = 256u << 20;
  if (cold) {
    CK(hipMalloc(&junk, jn * 4));
    CK(hipMalloc(&junk2, jn * 4));
  }
  auto run = [&](const char *name, auto launch) {
    if (cold) {
      float tot = 0;
      for (int i = 0; i < 20; ++i) {
        CK(hipMemcpyAsync(junk2, junk, jn * 4, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 4) tot += ms;
      }
      printf("%-28s %7.2f us  %.2f TB/s (cold)\n", name, tot / 16 * 1e3, n * 4 / (tot / 16 * 1e-3) / 1e12);
      return;
    }
    for (int i = 0; i < 10; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < 100; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %7.2f us  %.2f TB/s\n", name, ms * 10, n * 4 / (ms * 1e-5) / 1e12);
  };
  run("linear, 256 WGs", [&] { hipLaunchKernelGGL(k_linear, dim3(256), dim3(256), 0, 0, (float4 *)out, n / 4); });
  if (brief) {
    run("linear, 1024 WGs", [&] { hipLaunchKernelGGL(k_linear, dim3(1024), dim3(256), 0, 0, (float4 *)out, n / 4); });
    run("pattern KW=4", [&] { hipLaunchKernelGGL(k_pattern<4>, dim3((P + 15) / 16, B), dim3(256), 0, 0, out, P, N, 0); });
    run("pattern KW=4 + 1 load", [&] { hipLaunchKernelGGL(k_pattern<4>, dim3((P + 15) / 16, B), dim3(256), 0, 0, out, P, N, 1); });
    run("pattern KW=4 + 2 loads", [&] { hipLaunchKernelGGL(k_pattern<4>, dim3((P + 15) / 16, B), dim3(256), 0, 0, out, P, N, 2); });
    // the same with the workgroups per CU capped by a dynamic LDS allocation (30 KB: 5, like k_step; 40 KB: 4; 53 KB: 3)
    for (int lds : {20000, 26000, 30720, 40000, 53000}) {
      char nm[64];
      snprintf(nm, sizeof nm, "pattern + 2 loads, %d B LDS", lds);
      run(nm, [&] { hipLaunchKernelGGL(k_pattern<4>, dim3((P + 15) / 16, B), dim3(256), lds, 0, out, P, N, 2); });
    }
    run("pattern KW=4 + 3 loads", [&] { hipLaunchKernelGGL(k_pattern<4>, dim3((P + 15) / 16, B), dim3(256), 0, 0, out, P, N, 3); });
    run("zero role alone, 256 WGs", [&] { hipLaunchKernelGGL(k_two_roles, dim3(256), dim3(256), 0, 0, out, P, N, B, 256, 1, -1); });
    run("zero role alone, 512 WGs", [&] { hipLaunchKernelGGL(k_two_roles, dim3(512), dim3(256), 0, 0, out, P, N, B, 512, 1, -1); });
    run("live role alone", [&] { hipLaunchKernelGGL(k_two_roles, dim3((P + 15) / 16 * B), dim3(256), 0, 0, out, P, N, B, 0, 1, 2); });
    for (int zwg : {192, 256, 320, 384, 512})
      for (int nh : {1}) {
        char nm[64];
        snprintf(nm, sizeof nm, "two roles: %d zero WGs, nh=%d", zwg, nh);
        run(nm, [&] { hipLaunchKernelGGL(k_two_roles, dim3(zwg + (P + 15) / 16 * B), dim3(256), 0, 0, out, P, N, B, zwg, nh, 2); });
      }
    run("pattern KW=8", [&] { hipLaunchKernelGGL(k_pattern<8>, dim3((P + 31) / 32, B), dim3(256), 0, 0, out, P, N, 0); });
    run("pattern KW=8 + 2 loads", [&] { hipLaunchKernelGGL(k_pattern<8>, dim3((P + 31) / 32, B), dim3(256), 0, 0, out, P, N, 2); });
    return 0;
  }
  run("linear, 512 WGs", [&] { hipLaunchKernelGGL(k_linear, dim3(512), dim3(256), 0, 0, (float4 *)out, n / 4); });
  run("linear, 1024 WGs", [&] { hipLaunchKernelGGL(k_linear, dim3(1024), dim3(256), 0, 0, (float4 *)out, n / 4); });
  run("linear, 2048 WGs", [&] { hipLaunchKernelGGL(k_linear, dim3(2048), dim3(256), 0, 0, (float4 *)out, n / 4); });
  run("linear, 8192 WGs", [&] { hipLaunchKernelGGL(k_linear, dim3(8192), dim3(256), 0, 0, (float4 *)out, n / 4); });
  for (int wg : {256, 512, 1024, 2048, 4096}) {
    char nm[64];
    snprintf(nm, sizeof nm, "blocked, %d WGs", wg);
    run(nm, [&] { hipLaunchKernelGGL(k_blocked, dim3(wg), dim3(256), 0, 0, (float4 *)out, n / 4); });
    snprintf(nm, sizeof nm, "blocked x4, %d WGs", wg);
    run(nm, [&] { hipLaunchKernelGGL(k_blocked_u<4>, dim3(wg), dim3(256), 0, 0, (float4 *)out, n / 4); });
  }
  run("pattern KW=4", [&] { hipLaunchKernelGGL(k_pattern<4>, dim3((P + 15) / 16, B), dim3(256), 0, 0, out, P, N, 0); });
  run("pattern KW=4 + 2 loads", [&] { hipLaunchKernelGGL(k_pattern<4>, dim3((P + 15) / 16, B), dim3(256), 0, 0, out, P, N, 2); });
  run("pattern KW=4 buffer plain", [&] { hipLaunchKernelGGL(k_pattern_aux<0>, dim3((P + 15) / 16, B), dim3(256), 0, 0, out, P, N); });
  run("pattern KW=4 buffer sc0", [&] { hipLaunchKernelGGL(k_pattern_aux<1>, dim3((P + 15) / 16, B), dim3(256), 0, 0, out, P, N); });
  run("pattern KW=4 buffer nt", [&] { hipLaunchKernelGGL(k_pattern_aux<2>, dim3((P + 15) / 16, B), dim3(256), 0, 0, out, P, N); });
  run("pattern KW=4 buffer sc1", [&] { hipLaunchKernelGGL(k_pattern_aux<16>, dim3((P + 15) / 16, B), dim3(256), 0, 0, out, P, N); });
  run("pattern KW=4 buffer sc0 nt", [&] { hipLaunchKernelGGL(k_pattern_aux<3>, dim3((P + 15) / 16, B), dim3(256), 0, 0, out, P, N); });
  run("pattern KW=4 buffer sc1 nt", [&] { hipLaunchKernelGGL(k_pattern_aux<18>, dim3((P + 15) / 16, B), dim3(256), 0, 0, out, P, N); });
  run("pattern KW=4 buffer sc0 sc1", [&] { hipLaunchKernelGGL(k_pattern_aux<17>, dim3((P + 15) / 16, B), dim3(256), 0, 0, out, P, N); });
  for (int wg : {256, 512, 1024, 1280, 2048}) {
    char nm[64];
    snprintf(nm, sizeof nm, "pattern persistent, %d WGs", wg);
    run(nm, [&] { hipLaunchKernelGGL(k_pattern_persistent, dim3(wg), dim3(256), 0, 0, out, P, N, B); });
  }
  for (int wg : {256, 512, 1024}) {
    for (int cp : {64, 256, 1024}) {
      char nm[64];
      snprintf(nm, sizeof nm, "plane-major %d WGs x %d pillars", wg, cp);
      run(nm, [&] { hipLaunchKernelGGL(k_plane_major, dim3(wg), dim3(256), 0, 0, out, P, N, B, cp); });
    }
  }
  run("hipMemsetAsync", [&] { CK(hipMemsetAsync(out, 0, n * 4, 0)); });
  run("pattern KW=8", [&] { hipLaunchKernelGGL(k_pattern<8>, dim3((P + 31) / 32, B), dim3(256), 0, 0, out, P, N, 0); });
  run("pattern KW=16", [&] { hipLaunchKernelGGL(k_pattern<16>, dim3((P + 63) / 64, B), dim3(256), 0, 0, out, P, N, 0); });
  run("pattern KW=32", [&] { hipLaunchKernelGGL(k_pattern<32>, dim3((P + 127) / 128, B), dim3(256), 0, 0, out, P, N, 0); });
  return 0;
}
