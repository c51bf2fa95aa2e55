// Layout probe for v_mfma_f32_4x4x1_16b_f32: D_b[i][j] += A_b[i] * B_b[j], 16 blocks b.
// Expectation checked here: A lane l = A[b = l / 4][i = l % 4]; B lane l = B[b = l / 4][j = l % 4];
// D lane l, register v = D[b = l / 4][i = v][j = l % 4].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int l = threadIdx.x;
  const float a = 1.0f + l, b = 100.0f + l;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
    const float want = (1.0f + (4 * (l / 4) + v)) * (100.0f + l);
    if (h[l * 4 + v] != want) { if (bad < 5) printf("lane %d reg %d got %f want %f\n", l, v, h[l * 4 + v], want); ++bad; }
  }
  printf("mfma 4x4x1 layout %s (%d mismatches)\n", bad ? "DIFFERS" : "as expected", bad);
  return 0;
}
