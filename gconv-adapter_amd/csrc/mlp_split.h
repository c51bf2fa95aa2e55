// fp32 -> scaled fp16 pair, shared by the kernels that PRODUCE an operand of the f16x2 MLP GEMMs
// (csrc/mlp_f16.hip): the row-split pass, the GEMM epilogue that writes the hidden activation, and the
// GIN aggregation that writes the MLP's input directly (csrc/gin.hip).
//
// An fp32 value x is carried as  x * 2^e = hi + lo  with hi = fp16(x 2^e), lo = fp16(x 2^e - hi): 11 + 11
// mantissa bits, a relative error of 2^-22 — provided the scaled values sit inside fp16's range (max
// 65504 = 2^16, normal above 2^-14). `e` is ONE exponent per matrix row (activations, gradients: a node's
// row; weights: one for the whole matrix), chosen so that the row's largest magnitude — or, for the hidden
// activation, an upper BOUND of it — lands in [2^13, 2^14): as high as fp16 allows with a factor 4 to spare.
// An entry's lo is a normal fp16 number while the entry is above 2^-17 of the row's scale (2^-3 scaled);
// smaller entries keep an absolute error of 2^-25 scaled = 2^-38 of the row's scale, far below fp32's own
// resolution of the row's sums. The headroom matters for the hidden activation, whose bound (row maximum x
// largest weight-row 1-norm + largest bias) overestimates the typical entry by one to three orders of
// magnitude. Powers of two: scaling and un-scaling are exact; a row of zeros takes the largest exponent
// (its contribution to any bound vanishes).
#pragma once

#include "common.h"

namespace gconv {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int kMlpBK = 32;                       // contraction step of the GEMM = padding unit of a row
__host__ __device__ inline int mlp_pad_k(int k) { return (k + kMlpBK - 1) / kMlpBK * kMlpBK; }

constexpr int kPairTopExp = 14;                  // scaled row maximum (or bound) in [2^13, 2^14)
constexpr float kPairTop = 16384.f;              // 2^kPairTopExp: what a row's scaled entries stay below

// exponent e with bound * 2^e in [2^13, 2^14) (the largest, 100, for bound == 0); clamped so that 2^e and
// 2^-e are finite
__device__ __forceinline__ int scale_exp(float bound) {
  if (!(bound > 0.f)) return 100;
  int e;
  (void)frexpf(bound, &e);                       // bound = f 2^e, f in [0.5, 1)
  e = kPairTopExp - e;
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}

__device__ __forceinline__ void split2(float xs, _Float16& hi, _Float16& lo) {   // xs: already scaled
  hi = static_cast<_Float16>(xs);
  lo = static_cast<_Float16>(xs - static_cast<float>(hi));
}

// 4 consecutive values of a row -> the two images (8 bytes each)
__device__ __forceinline__ void split2x4(f32x4 v, int e, f16x4& h, f16x4& l) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    _Float16 hi, lo;
    split2(ldexpf(v[k], e), hi, lo);
    h[k] = hi;
    l[k] = lo;
  }
}

// layout of a row pair [2][N][Kp] fp16 + int32 exponent per row
__host__ __device__ inline size_t mlp_pair_image_elems(int n, int f) { return static_cast<size_t>(n > 0 ? n : 1) * mlp_pad_k(f); }
__host__ __device__ inline size_t mlp_pair_exp_offset(int n, int f) {   // bytes
  return (2 * mlp_pair_image_elems(n, f) * 2 + 15) / 16 * 16;
}
__host__ __device__ inline size_t mlp_pair_bytes(int n, int f) {
  return mlp_pair_exp_offset(n, f) + (static_cast<size_t>(n > 0 ? n : 1) * 4 + 15) / 16 * 16;
}

}  // namespace gconv
