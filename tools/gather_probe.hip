// Probe: ceiling of random 128-B row gathers (the SpMM's access pattern) on MI355X.
// Not part of the product; build: hipcc --offload-arch=gfx950 -O3 -o gather_probe gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// each 8-lane group sums DEG gathered rows (F=32 floats) and writes one row
template <int DEG>
__global__ __launch_bounds__(256) void gather_sum(const float* __restrict__ T, const int* __restrict__ idx,
                                                  float* __restrict__ out, int ngroups) {
  int gid = (blockIdx.x * 256 + threadIdx.x) >> 3, sub = threadIdx.x & 7;
  if (gid >= ngroups) return;
  const int* ip = idx + (size_t)gid * DEG;
  f32x4 acc = {0, 0, 0, 0};
  int c[DEG];
#pragma unroll
  for (int u = 0; u < DEG; ++u) c[u] = ip[u];
#pragma unroll
  for (int u = 0; u < DEG; ++u) acc += *(const f32x4*)(T + (size_t)c[u] * 32 + 4 * sub);
  *(f32x4*)(out + (size_t)gid * 32 + 4 * sub) = acc;
}
__global__ void copy4(const f32x4* __restrict__ a, f32x4* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
template <typename F> float timeit(F f, int it = 20) {
  for (int i = 0; i < 3; ++i) f();
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a)); for (int i = 0; i < it; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / it * 1e3f;  // us
}
int main() {
  const int N = 170000, M = 1370000, DEG = 8, G = M / DEG;
  std::mt19937 rng(1);
  std::vector<int> uni(M), pl(M);
  for (int i = 0; i < M; ++i) uni[i] = rng() % N;
  // half power-law (v = N u^3), half uniform: the S-arxiv mix
  std::uniform_real_distribution<double> U(0, 1);
  for (int i = 0; i < M; ++i) { double u = U(rng); pl[i] = (i & 1) ? (int)(N * u * u * u) : (int)(rng() % N); }
  float *T, *out; int *di; f32x4 *ca, *cb;
  CK(hipMalloc(&T, (size_t)N * 32 * 4)); CK(hipMalloc(&out, (size_t)G * 32 * 4)); CK(hipMalloc(&di, (size_t)M * 4));
  CK(hipMemset(T, 0, (size_t)N * 32 * 4));
  size_t cn = (size_t)175 * 1000 * 1000 / 16; CK(hipMalloc(&ca, cn * 16)); CK(hipMalloc(&cb, cn * 16));
  float us = timeit([&] { copy4<<<2048, 256>>>(ca, cb, cn); });
  printf("copy 175MB->175MB: %.1f us  %.2f TB/s (r+w)\n", us, 2 * cn * 16 / us / 1e6);
  for (int pass = 0; pass < 2; ++pass) {
    CK(hipMemcpy(di, pass ? pl.data() : uni.data(), (size_t)M * 4, hipMemcpyHostToDevice));
    us = timeit([&] { gather_sum<DEG><<<(G * 8 + 255) / 256, 256>>>(T, di, out, G); });
    double gb = (double)M * 128 + (double)G * 128 + (double)M * 4;
    printf("%s gather deg8: %.1f us  gathered %.2f TB/s (%.0f MB)\n", pass ? "arxiv-mix" : "uniform  ", us, gb / us / 1e6, gb / 1e6);
  }
  // sorted indices within each group don't matter; try fully sorted idx (best locality)
  std::sort(uni.begin(), uni.end());
  CK(hipMemcpy(di, uni.data(), (size_t)M * 4, hipMemcpyHostToDevice));
  us = timeit([&] { gather_sum<DEG><<<(G * 8 + 255) / 256, 256>>>(T, di, out, G); });
  printf("sorted    gather deg8: %.1f us\n", us);
  return 0;
}
