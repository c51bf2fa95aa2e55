// Fused Adam step over a flat fp32 parameter range.
//
// Replaces, for the trainable set of the adapter fine-tuning loops (adapters + head + norm
// affine parameters, ~29 k floats), the ~10 multi-tensor launches of
//   torch.optim.Adam(...).step()   /root/reference/scripts/finetune_inductive_learning.py:379,67
//                                  /root/reference/scripts/finetune_transductive_learning.py:205,230
// with ONE launch per parameter group. Same update rule and the same operation order as
// torch's (non-amsgrad, non-capturable) Adam, in fp32:
//   g   = grad + weight_decay * p                       (L2 penalty, as torch.optim.Adam)
//   m   = m + (1 - beta1) * (g - m)                      (lerp)
//   v   = beta2 * v + (1 - beta2) * g * g
//   p   = p - (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
#include <cmath>

#include "common.h"

namespace gconv {

struct AdamArgs {
  float* p; const float* g; float* m; float* v;
  long long n;
  float one_minus_b1, b2, one_minus_b2, eps, wd, step_size, bc2_sqrt;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a) {
  if (a.wd != 0.f) g = g + a.wd * p;
  m = m + a.one_minus_b1 * (g - m);
  v = a.b2 * v + a.one_minus_b2 * g * g;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p = p - a.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_step_kernel(AdamArgs a) {
  const long long i4 = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) * 4;
  if (i4 + 3 < a.n && ((reinterpret_cast<uintptr_t>(a.p) | reinterpret_cast<uintptr_t>(a.g) |
                        reinterpret_cast<uintptr_t>(a.m) | reinterpret_cast<uintptr_t>(a.v)) & 15u) == 0) {
    f32x4 p = ldg4(a.p + i4), m = ldg4(a.m + i4), v = ldg4(a.v + i4);
    const f32x4 g = ldg4(a.g + i4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pk = p[k], mk = m[k], vk = v[k];
      adam_one(pk, g[k], mk, vk, a);
      p[k] = pk; m[k] = mk; v[k] = vk;
    }
    stg4(a.p + i4, p); stg4(a.m + i4, m); stg4(a.v + i4, v);
  } else {
    for (long long i = i4; i < a.n && i < i4 + 4; ++i) adam_one(a.p[i], a.g[i], a.m[i], a.v[i], a);
  }
}

}  // namespace gconv

using namespace gconv;

extern "C" gconv_status_t gconv_adam_step_f32(float* param, const float* grad, float* exp_avg,
                                              float* exp_avg_sq, int64_t count, float lr, float beta1,
                                              float beta2, float eps, float weight_decay, int64_t step,
                                              gconv_stream_t stream) {
  if (count < 0 || step < 1) return GCONV_E_INVALID_ARG;
  if (count == 0) return GCONV_OK;
  if (!param || !grad || !exp_avg || !exp_avg_sq) return GCONV_E_INVALID_ARG;
  if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f)) return GCONV_E_INVALID_ARG;
  // bias corrections in double on the host, as torch computes them from Python floats
  const double bc1 = 1.0 - std::pow(static_cast<double>(beta1), static_cast<double>(step));
  const double bc2 = 1.0 - std::pow(static_cast<double>(beta2), static_cast<double>(step));
  AdamArgs a{};
  a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = count;
  a.one_minus_b1 = static_cast<float>(1.0 - static_cast<double>(beta1));
  a.b2 = beta2;
  a.one_minus_b2 = static_cast<float>(1.0 - static_cast<double>(beta2));
  a.eps = eps; a.wd = weight_decay;
  a.step_size = static_cast<float>(static_cast<double>(lr) / bc1);
  a.bc2_sqrt = static_cast<float>(std::sqrt(bc2));
  const long long groups = (count + 3) / 4;
  const unsigned grid = static_cast<unsigned>((groups + 255) / 256);
  adam_step_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(a);
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}
