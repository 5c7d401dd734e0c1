// microbenchmark: LDS f64 atomic-add vs bank distribution of distinct addresses (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
__global__ __launch_bounds__(256) void k(double *out, int iters, const int *slots)
{
  __shared__ double acc[4][2304];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int t = lane; t < 2304; t += 64) acc[wave][t] = 0;
  __syncthreads();
  double v = 1.0 + lane;
  int base[8];
  for (int r = 0; r < 8; ++r) base[r] = slots[r * 64 + lane] * 9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int q = 0; q < 9; ++q)
        __hip_atomic_fetch_add(&acc[wave][base[r] + q], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = acc[wave][lane] + v;
}
int main()
{
  double *out; hipMalloc(&out, 8 * 256 * 2048);
  int *d; hipMalloc(&d, 4 * 512);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 250, grid = 2048;
  std::mt19937 rng(1);
  auto run = [&](const char *name, std::vector<int> &s) {
    // bank statistics: max lanes per double-bank (32 double banks) per round
    double avgmax = 0;
    for (int r = 0; r < 8; ++r) { int c[32] = {0}; int m = 0; for (int l = 0; l < 64; ++l) m = std::max(m, ++c[(s[r*64+l]*9) & 31]); avgmax += m / 8.0; }
    double avgmax16 = 0;
    for (int r = 0; r < 8; ++r) { int m = 0; for (int h = 0; h < 2; ++h) { int c[32] = {0}; for (int l = 0; l < 32; ++l) m = std::max(m, ++c[(s[r*64+h*32+l]*9) & 31]); } avgmax16 += m / 8.0; }
    hipMemcpy(d, s.data(), 4 * 512, hipMemcpyHostToDevice);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, iters, d);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    double winstr = (double)grid * 4 * iters * 72;
    printf("%-44s max/bank(64) %.2f max/bank(half) %.2f  %8.3f ms  %6.1f CU-clk/wave-instr\n", name, avgmax, avgmax16, ms, ms * 1e-3 * 2.4e9 * 256 / winstr);
  };
  std::vector<int> s(512);
  for (int r = 0; r < 8; ++r) for (int l = 0; l < 64; ++l) s[r*64+l] = l;
  run("slots 0..63 (2 per double-bank)", s);
  for (int r = 0; r < 8; ++r) for (int l = 0; l < 64; ++l) s[r*64+l] = l * 4;
  run("slots 4l (8 per double-bank)", s);
  for (int r = 0; r < 8; ++r) for (int l = 0; l < 64; ++l) s[r*64+l] = l * 2;
  run("slots 2l (4 per double-bank)", s);
  for (int r = 0; r < 8; ++r) for (int l = 0; l < 64; ++l) s[r*64+l] = (l % 32) + 32 * (l / 32) * 3;
  run("slots l%32 + 96*(l/32) (2 per bank, halves)", s);
  for (int r = 0; r < 8; ++r) { std::vector<int> p(240); for (int i = 0; i < 240; ++i) p[i] = i; std::shuffle(p.begin(), p.end(), rng); for (int l = 0; l < 64; ++l) s[r*64+l] = p[l]; }
  run("random distinct of 240", s);
  for (int r = 0; r < 8; ++r) { std::vector<int> p(64); for (int i = 0; i < 64; ++i) p[i] = (i % 32) + 32 * (rng() % 7); std::shuffle(p.begin(), p.end(), rng); for (int l = 0; l < 64; ++l) s[r*64+l] = p[l]; }
  run("random, exactly 2 per double-bank", s);
  for (int r = 0; r < 8; ++r) { for (int h = 0; h < 2; ++h) { std::vector<int> p(32); for (int i = 0; i < 32; ++i) p[i] = i + 32 * (rng() % 7); std::shuffle(p.begin(), p.end(), rng); for (int l = 0; l < 32; ++l) s[r*64+h*32+l] = p[l]; } }
  run("random, 1 per double-bank per half-wave", s);
  for (int r = 0; r < 8; ++r) { for (int h = 0; h < 4; ++h) { std::vector<int> p(16); for (int i = 0; i < 16; ++i) p[i] = 2*i + (rng()&1) + 32 * (rng() % 7); std::shuffle(p.begin(), p.end(), rng); for (int l = 0; l < 16; ++l) s[r*64+h*16+l] = p[l]; } }
  run("random, quarter-waves cover 16 banks each", s);
  return 0;
}
