// The frozen GIN layer's MLP: Linear(d -> 2d) + ReLU + Linear(2d -> d) of
// /root/reference/src/layers/inductive/gin_conv.py:24-28 (`self.mlp`, called from `update`, :85),
// forward and input gradient (the layer is frozen under the adapter recipe, gnn.py:230-241: its
// weights never need a gradient).
//
// These are the only NON-skinny GEMMs next to the adapter path (K = 300 / 600, 9.4 GFLOP per
// direction on a 512-molecule batch): library fp32 GEMMs run them at 60-90 TFLOP/s on the fp32
// MFMA, whose rate is the vector rate. Here every fp32 operand is written as the sum of THREE bf16
// values (hi + mid + lo carry 8 + 8 + 8 mantissa bits: an fp32 value exactly, up to the last
// rounding) and the product as the six cross terms that matter,
//     a b ~= ah bh + ah bm + am bh + ah bl + al bh + am bm          (dropped: <= 2^-23 |a b|)
// on v_mfma_f32_16x16x32_bf16 — 16x the fp32 MFMA rate per product, so 6 products cost 3/8 of one
// fp32 product — accumulated in fp32. The weights are frozen: they are split ONCE
// (gconv_mlp_split_weights_bf16x3); the activations are split on the fly while a tile goes from
// global memory to LDS (eight VALU operations per element, amortised over the 64 output columns of
// the tile). fp32-accurate, not fp32-exact: the layer's parity bar (1e-5 against the fp64
// restatement, tests/test_gin.py) is the gate, `GCONV_TORCH_MLP=1` the A/B knob.
//
// One kernel, four uses per layer:
//   forward   H  = relu(A W1^T + b1)         A [N, d],  W1 [2d, d]
//             Y  = H W2^T + b2               H [N, 2d], W2 [d, 2d]
//   backward  dH = (dY W2) . [H > 0]         "weights" = W2^T, mask = H
//             dA = dH W1                     "weights" = W1^T
//
// Tiling: a workgroup (4 waves) computes 128 rows x 64 columns; a wave owns 32 rows x 64 columns
// (2 x 4 accumulator tiles of 16 x 16). Per k-step of 32: the A tile (128 x 32 fp32) is split and
// parked in LDS as three bf16 images, the three weight images (64 x 32 bf16 each) are copied;
// rows are padded to 80 bytes so that the 16-byte fragment reads of 16 consecutive rows hit 16
// different bank groups. The next k-step's global loads are in flight while the current one is
// multiplied.
#include "common.h"

namespace gconv {

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kBM = 128, kBN = 64, kBK = 32;
constexpr int kRowB = 80;                       // bytes per LDS row: 32 bf16 + 16 bytes of padding
constexpr int kAImg = kBM * kRowB, kBImg = kBN * kRowB;

__device__ __forceinline__ unsigned short bf16_rn(float x) {   // round to nearest even (v_cvt_pk_bf16_f32)
  const __bf16 b = static_cast<__bf16>(x);
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf16_up(unsigned short b) {
  return __builtin_bit_cast(float, static_cast<unsigned int>(b) << 16);
}
// x = hi + mid + lo (+ at most one rounding of the last term)
__device__ __forceinline__ void split3(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
  hi = bf16_rn(x);
  const float r1 = x - bf16_up(hi);
  mid = bf16_rn(r1);
  const float r2 = r1 - bf16_up(mid);
  lo = bf16_rn(r2);
}

// workgroup barrier for LDS hand-offs: __syncthreads() would also drain the global loads of the
// NEXT k-step, which are meant to stay in flight across it
__device__ __forceinline__ void lds_only_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

}  // namespace

// W [rows, cols] fp32 (or its transpose) -> three bf16 images [3][Mp][Kp], Mp = roundup(M, 64),
// Kp = roundup(K, 32), zero padded; (M, K) = (rows, cols), or (cols, rows) when transposed.
__global__ void mlp_split_weights_kernel(const float* __restrict__ w, int rows, int cols, int transpose,
                                         int M, int K, int Mp, int Kp, unsigned short* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Mp * Kp) return;
  const int m = idx / Kp, k = idx % Kp;
  float v = 0.f;
  if (m < M && k < K) v = transpose ? w[static_cast<size_t>(k) * cols + m] : w[static_cast<size_t>(m) * cols + k];
  unsigned short hi, mid, lo;
  split3(v, hi, mid, lo);
  const size_t img = static_cast<size_t>(Mp) * Kp;
  out[idx] = hi;
  out[img + idx] = mid;
  out[2 * img + idx] = lo;
  (void)rows;
}

// EPI: 0 none, 1 relu, 2 multiply by [mask > 0]
template <int EPI>
__global__ __launch_bounds__(256, 2) void mlp_gemm_bf16x3_kernel(const float* __restrict__ A,
                                                                 const unsigned short* __restrict__ W,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ mask,
                                                                 float* __restrict__ C, int N, int K, int M,
                                                                 int Mp, int Kp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* la = smem;                      // [3][kBM][kRowB]
  char* lb = smem + 3 * kAImg;          // [3][kBN][kRowB]
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
  const int row0 = blockIdx.x * kBM, col0 = blockIdx.y * kBN;
  const size_t wimg = static_cast<size_t>(Mp) * Kp;

  // staging roles: A tile = 128 rows x 8 float4 -> thread (r = tid / 8 + 32 j, q = tid % 8), j = 0..3;
  // weight tile = 64 rows x 4 x 16 bytes per image -> thread (r = tid / 4, q = tid % 4), 3 images
  const int ar = tid >> 3, aq = tid & 7;
  const int br = tid >> 2, bq = tid & 3;
  f32x4 av[4];
  bf16x8 bv[3];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = row0 + ar + 32 * j, k = k0 + 4 * aq;
      // (K is a multiple of 4: a float4 is inside the row or outside it)
      av[j] = (r < N && k < K) ? ldg4(A + static_cast<size_t>(r) * K + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int s = 0; s < 3; ++s)
      bv[s] = *reinterpret_cast<const bf16x8*>(W + s * wimg + static_cast<size_t>(col0 + br) * Kp + k0 + 8 * bq);
  };
  auto park_tiles = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x4 h, m, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned short hi, mid, lo;
        split3(av[j][e], hi, mid, lo);
        h[e] = static_cast<short>(hi); m[e] = static_cast<short>(mid); l[e] = static_cast<short>(lo);
      }
      const int off = (ar + 32 * j) * kRowB + 8 * aq;
      *reinterpret_cast<bf16x4*>(la + off) = h;
      *reinterpret_cast<bf16x4*>(la + kAImg + off) = m;
      *reinterpret_cast<bf16x4*>(la + 2 * kAImg + off) = l;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) *reinterpret_cast<bf16x8*>(lb + s * kBImg + br * kRowB + 16 * bq) = bv[s];
  };

  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: lane l reads row (l & 15), k-group (l >> 4) of a 16 x 32 image slice
  const int frow = lane & 15, fk = lane >> 4;
  const int a_off = (32 * wave + frow) * kRowB + 16 * fk;
  const int b_off = frow * kRowB + 16 * fk;

  load_tiles(0);
  for (int k0 = 0; k0 < Kp; k0 += kBK) {
    park_tiles();
    lds_only_barrier();
    if (k0 + kBK < Kp) load_tiles(k0 + kBK);   // in flight during the MFMAs below (and across the barrier)
    bf16x8 af[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < 3; ++s) af[i][s] = *reinterpret_cast<const bf16x8*>(la + s * kAImg + a_off + 16 * i * kRowB);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x8 bf[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) bf[s] = *reinterpret_cast<const bf16x8*>(lb + s * kBImg + b_off + 16 * j * kRowB);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        // smallest terms first: the accumulator sees them before the dominant product
        f32x4 c = acc[i][j];
        c = mfma_bf16(af[i][1], bf[1], c);   // mid * mid
        c = mfma_bf16(af[i][2], bf[0], c);   // lo  * hi
        c = mfma_bf16(af[i][0], bf[2], c);   // hi  * lo
        c = mfma_bf16(af[i][1], bf[0], c);   // mid * hi
        c = mfma_bf16(af[i][0], bf[1], c);   // hi  * mid
        c = mfma_bf16(af[i][0], bf[0], c);   // hi  * hi
        acc[i][j] = c;
      }
    }
    lds_only_barrier();   // the images are rewritten by the next step
  }

  // epilogue: C layout of the 16 x 16 tile: column = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = col0 + 16 * j + (lane & 15);
    const float bc = (bias && c < M) ? bias[c] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = row0 + 32 * wave + 16 * i + 4 * (lane >> 4) + v;
        if (r < N && c < M) {
          float y = acc[i][j][v] + bc;
          if (EPI == 1) y = y > 0.f ? y : 0.f;
          if (EPI == 2) y = mask[static_cast<size_t>(r) * M + c] > 0.f ? y : 0.f;
          C[static_cast<size_t>(r) * M + c] = y;
        }
      }
    }
  }
}

}  // namespace gconv

using namespace gconv;

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

extern "C" size_t gconv_mlp_split_bytes(int32_t out_features, int32_t in_features) {
  if (out_features <= 0 || in_features <= 0) return 0;
  return static_cast<size_t>(3) * round_up(out_features, kBN) * round_up(in_features, kBK) * sizeof(unsigned short);
}

extern "C" gconv_status_t gconv_mlp_split_weights_bf16x3(const float* w, int32_t rows, int32_t cols,
                                                         int32_t transpose, void* out,
                                                         gconv_stream_t stream_) {
  if (!w || !out || rows <= 0 || cols <= 0) return GCONV_E_INVALID_ARG;
  const int M = transpose ? cols : rows, K = transpose ? rows : cols;
  const int Mp = round_up(M, kBN), Kp = round_up(K, kBK);
  const int total = Mp * Kp;
  mlp_split_weights_kernel<<<(total + 255) / 256, 256, 0, static_cast<hipStream_t>(stream_)>>>(
      w, rows, cols, transpose ? 1 : 0, M, K, Mp, Kp, static_cast<unsigned short*>(out));
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

extern "C" gconv_status_t gconv_mlp_gemm_bf16x3_f32(const float* a, const void* w_split, const float* bias,
                                                    const float* mask, float* c, int32_t num_rows,
                                                    int32_t in_features, int32_t out_features,
                                                    int32_t epilogue, gconv_stream_t stream_) {
  if (!a || !w_split || !c || num_rows < 0 || in_features <= 0 || out_features <= 0) return GCONV_E_INVALID_ARG;
  if (epilogue < 0 || epilogue > 2 || (epilogue == 2 && !mask)) return GCONV_E_INVALID_ARG;
  if (in_features % 4 != 0) return GCONV_E_UNSUPPORTED;
  if (!aligned16(a) || !aligned16(w_split)) return GCONV_E_ALIGNMENT;
  if (num_rows == 0) return GCONV_OK;
  const int Mp = round_up(out_features, kBN), Kp = round_up(in_features, kBK);
  const dim3 grid((num_rows + kBM - 1) / kBM, Mp / kBN);
  const size_t lds = 3 * (kAImg + kBImg);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const unsigned short* w = static_cast<const unsigned short*>(w_split);
  static_assert(3 * (kAImg + kBImg) <= 48 * 1024, "dynamic LDS above 48 KB needs the opt-in attribute");
#define GCONV_MLP_LAUNCH(EPI_)                                                                                   \
  mlp_gemm_bf16x3_kernel<EPI_><<<grid, 256, lds, stream>>>(a, w, bias, mask, c, num_rows, in_features,           \
                                                           out_features, Mp, Kp)
  switch (epilogue) {
    case 0: GCONV_MLP_LAUNCH(0); break;
    case 1: GCONV_MLP_LAUNCH(1); break;
    default: GCONV_MLP_LAUNCH(2); break;
  }
#undef GCONV_MLP_LAUNCH
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}
