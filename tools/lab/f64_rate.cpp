// tools/lab/f64_rate.cpp: issue rate of f64 VALU instructions on one SIMD (one wave per SIMD and four):
// cycles per v_fma_f64 / v_add_f64 / v_mul_f64 / v_rcp_f64 / a full IEEE division, independent chains; and (round 6) the
// LATENCY of a dependent v_mul_f64 -> v_add_f64 pair (the running mean's chain step), one chain and three side by side.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int OP>
__global__ void k(double *out, double a, double b, long long *cyc) {
  double x0 = a + threadIdx.x, x1 = a * 2, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N / 8; ++i) {
    if (OP == 0) { x0 = fma(x0, b, a); x1 = fma(x1, b, a); x2 = fma(x2, b, a); x3 = fma(x3, b, a); x4 = fma(x4, b, a); x5 = fma(x5, b, a); x6 = fma(x6, b, a); x7 = fma(x7, b, a); }
    if (OP == 1) { x0 += b; x1 += b; x2 += b; x3 += b; x4 += b; x5 += b; x6 += b; x7 += b; }
    if (OP == 2) { x0 *= b; x1 *= b; x2 *= b; x3 *= b; x4 *= b; x5 *= b; x6 *= b; x7 *= b; }
    if (OP == 3) { x0 = __builtin_amdgcn_rcp(x0); x1 = __builtin_amdgcn_rcp(x1); x2 = __builtin_amdgcn_rcp(x2); x3 = __builtin_amdgcn_rcp(x3); x4 = __builtin_amdgcn_rcp(x4); x5 = __builtin_amdgcn_rcp(x5); x6 = __builtin_amdgcn_rcp(x6); x7 = __builtin_amdgcn_rcp(x7); }
    if (OP == 4) { x0 = a / x0; x1 = a / x1; x2 = a / x2; x3 = a / x3; x4 = a / x4; x5 = a / x5; x6 = a / x6; x7 = a / x7; }
    if (OP == 6) { for (int q = 0; q < 8; ++q) { x0 = x0 * b; asm volatile("" : "+v"(x0)); x0 = x0 + a; asm volatile("" : "+v"(x0)); } }
    if (OP == 7) { for (int q = 0; q < 8; ++q) { x0 = x0 * b; x1 = x1 * b; x2 = x2 * b; asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2)); x0 = x0 + a; x1 = x1 + a; x2 = x2 + a; asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2)); } }
    if (OP == 8) {  // the dependent pair with VGPR operands (a different register pair per step, like the chain loop's)
      double b0 = b + threadIdx.x * 1e-9, b1 = b0 * b, b2 = b1 * b, b3 = b2 * b, a0 = a + threadIdx.x * 1e-9, a1 = a0 * a, a2 = a1 * a, a3 = a2 * a;
      asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      for (int q = 0; q < 2; ++q) {
        asm volatile("v_mul_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %5\n\tv_mul_f64 %0, %0, %2\n\tv_add_f64 %0, %0, %6\n\t"
                     "v_mul_f64 %0, %0, %3\n\tv_add_f64 %0, %0, %7\n\tv_mul_f64 %0, %0, %4\n\tv_add_f64 %0, %0, %8"
                     : "+v"(x0) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(a0), "v"(a1), "v"(a2), "v"(a3));
      }
    }
    if (OP == 5) { float f0 = (float)x0, f1 = (float)x1; for (int q = 0; q < 4; ++q) { f0 = fmaf(f0, (float)b, (float)a); f1 = fmaf(f1, (float)b, (float)a); } x0 = f0; x1 = f1; }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP>
void run(const char *name, int threads) {
  double *out; long long *cyc, h;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  k<OP><<<1, threads>>>(out, 1.000001, 0.999999, cyc);
  k<OP><<<1, threads>>>(out, 1.000001, 0.999999, cyc);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-10s %4d threads/CU: %6.2f clocks (s_memtime units) per wave-instruction%s\n", name, threads, (double)h / N, OP == 4 ? " (= per division)" : "");
}
int main() {
  for (int th : {64, 256, 1024}) {
    if (th == 64) { run<6>("dep mul+add (per 1/8 pair... x8 pairs per trip: multiply by 1: clocks per PAIR)", 64); run<7>("3 chains of dep mul+add side by side (clocks per pair-triple)", 64); run<8>("dep mul+add, VGPR operands (clocks per pair)", 64); run<0>("fma_f64", 64); run<1>("add_f64", 64); run<2>("mul_f64", 64); run<3>("rcp_f64", 64); run<4>("div_f64", 64); }
    if (th == 256) { run<0>("fma_f64", 256); run<1>("add_f64", 256); run<3>("rcp_f64", 256); run<4>("div_f64", 256); }
    if (th == 1024) { run<0>("fma_f64", 1024); run<1>("add_f64", 1024); run<3>("rcp_f64", 1024); run<4>("div_f64", 1024); }
  }
  return 0;
}
