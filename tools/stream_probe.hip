// Probe: (1) read rate of an [N, d] fp32 matrix under different lane->address layouts,
//        (2) sustained v_mfma_f32_16x16x4_f32 rate.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// LAYOUT 0: "A layout": lane (node=l&15, kq=l>>4) reads X[i0+node][16q+4kq..], q = 0..d/16-1
// LAYOUT 1: "L layout": lane (jj=l&15, t=l>>4) reads X[i0+4t+m][64J+4jj..], m=0..3, J over chunks
// LAYOUT 2: fully coalesced: lane l reads float4 #l of the 16-row tile treated as flat memory
template <int LAYOUT, int D>
__global__ __launch_bounds__(256) void read_tile(const float* __restrict__ X, float* __restrict__ out, int nblk) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4 acc = {0, 0, 0, 0};
  for (int blk = blockIdx.x * 4 + wave; blk < nblk; blk += gridDim.x * 4) {
    const float* base = X + (size_t)blk * 16 * D;
    constexpr int NV = 16 * D / 4 / 64;  // float4 per lane per tile
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      size_t off;
      if (LAYOUT == 0) off = (size_t)(lane & 15) * D + 16 * i + 4 * (lane >> 4);
      else if (LAYOUT == 1) { int J = i / 4, m = i % 4; off = (size_t)(4 * (lane >> 4) + m) * D + 64 * J + 4 * (lane & 15); }
      else off = (size_t)(i * 64 + lane) * 4;
      v[i] = *(const f32x4*)(base + off);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) acc += v[i];
  }
  if (acc[0] == 12345.f) out[threadIdx.x] = acc[1] + acc[2] + acc[3];
}

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
  f32x4 a0 = {0,0,0,0}, a1 = a0, a2 = a0, a3 = a0;
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
  }
  f32x4 s = a0 + a1 + a2 + a3;
  if (s[0] == 12345.f) out[threadIdx.x] = s[1];
}
template <typename F> float timeit(F f, int it = 20) {
  for (int i = 0; i < 3; ++i) f();
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a)); for (int i = 0; i < it; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / it * 1e3f;
}
int main() {
  const int N = 170000, D = 128, nblk = N / 16;
  float *X, *out; CK(hipMalloc(&X, (size_t)N * D * 4)); CK(hipMalloc(&out, 4096)); CK(hipMemset(X, 0, (size_t)N * D * 4));
  double mb = (double)N * D * 4 / 1e6;
  for (int grid : {768, 1536, 2656}) {
    float t0 = timeit([&] { read_tile<0, D><<<grid, 256>>>(X, out, nblk); });
    float t1 = timeit([&] { read_tile<1, D><<<grid, 256>>>(X, out, nblk); });
    float t2 = timeit([&] { read_tile<2, D><<<grid, 256>>>(X, out, nblk); });
    printf("grid %4d  read %.0f MB: A-layout %.1f us (%.2f TB/s) | L-layout %.1f us (%.2f TB/s) | flat %.1f us (%.2f TB/s)\n",
           grid, mb, t0, mb / t0 / 1e3 * 1e-3 * 1e3, t1, mb / t1 / 1e3 * 1e-3 * 1e3, t2, mb / t2 / 1e3 * 1e-3 * 1e3);
  }
  // d = 300 variant (HIV): 13346 nodes
  {
    const int N2 = 13346 / 16 * 16, nb2 = N2 / 16; float* X2; CK(hipMalloc(&X2, (size_t)N2 * 304 * 4));
    (void)nb2; CK(hipFree(X2));
  }
  for (int waves_per_simd : {1, 2, 4}) {
    int iters = 2000;
    float t = timeit([&] { mfma_loop<<<256 * waves_per_simd, 256>>>(out, iters); }, 5);
    double flops = (double)256 * waves_per_simd * 4 * iters * 4 * 2048;
    printf("mfma 16x16x4 f32, %d wave(s)/SIMD: %.1f us  %.1f TFLOP/s  (%.1f cycles/mfma/SIMD at 2.4 GHz)\n", waves_per_simd, t,
           flops / t / 1e6, t * 1e-6 * 2.4e9 / (iters * 4.0 * waves_per_simd));
  }
  return 0;
}
