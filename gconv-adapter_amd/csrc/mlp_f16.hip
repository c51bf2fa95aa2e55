// The frozen GIN layer's MLP on the fp16 MFMA at fp32 accuracy (round 4):
// Linear(d -> 2d) + ReLU + Linear(2d -> d) of /root/reference/src/layers/inductive/gin_conv.py:24-28
// (`self.mlp`, called from `update`, :85), forward and input gradient of the frozen layer
// (src/models/inductive/gnn.py:230-241).
//
// Round 3 (csrc/mlp.hip) wrote every fp32 operand as THREE bf16 values and a product as six MFMAs. Here an
// operand is TWO fp16 values of a power-of-two SCALED copy (mlp_split.h: 11 + 11 mantissa bits; one exponent
// per activation row, one per weight matrix, chosen so that the values sit in fp16's normal range) and a
// product is THREE MFMAs,
//     a b ~= ah bh + ah bl + al bh                          (dropped: al bl <= 2^-22 |a b|)
// on v_mfma_f32_16x16x32_f16 with fp32 accumulation: half the matrix work of the bf16 triple, two thirds of
// its operand bytes (LDS staging, L2 traffic, the hidden activation's round trip), the same accuracy — both
// are dominated by the fp32 accumulation (emulation over K = 300, N(0, 1) data: max error 1.3e-6 here,
// 1.5e-6 for the bf16 triple, 3.7e-6 for a plain fp32 GEMM; parity gate: tests/test_gin.py at 1e-5 vs fp64
// and <= 2x torch's fp32 error). What the scaling needs:
//   weights   exponent from max |W| (scaled into [2^13, 2^14)); max row 1-norm and max |bias| are kept beside the
//             pair: they bound the hidden activation without looking at it
//   inputs    exponent from the row's max (row-split pass; or the producer: the GIN aggregation writes the pair)
//   hidden    |h[r, :]| <= 2^-e(r) 2^14 L1max + bmax — every column tile of row r derives the SAME exponent from
//             that bound; un-scaling happens in the epilogue in true units, so a GEMM's output row is exact in
//             the exponent bookkeeping whatever the tiles were
// Kernel shape as in round 3: 128 nodes x 80 features per workgroup of 4 waves, k-steps of 32, six (here: four)
// images by LDS-DMA into two XOR-swizzled LDS buffers, one barrier per k-step, XCD-aware tile order — but
// 26 KB per stage instead of 39: THREE workgroups per CU.
#include <cstdlib>

#include "mlp_split.h"

namespace gconv {

namespace {

constexpr int kBM = 128;                        // nodes per workgroup
constexpr int kBN = 80, kNT = kBN / 16;         // features per workgroup, feature tiles
constexpr int kBK = kMlpBK;
constexpr int kThreads = 256;
constexpr int kAChunks = 2 * kBM / 16, kBChunks = 2 * kBN / 16;   // 1 KB chunks (16 rows x 64 B) of a stage
constexpr int kBufB = (kAChunks + kBChunks) * 1024;
constexpr int kChunksPerWave = (kAChunks + kBChunks + 3) / 4;

__device__ __forceinline__ void lds_only_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void lds_dma16(const void* src, void* dst) {
  __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void*>(
                                       reinterpret_cast<uintptr_t>(src)),
                                   reinterpret_cast<__attribute__((address_space(3))) void*>(
                                       static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dst))),
                                   16, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_f16(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline int pad_m(int m) { return round_up(mlp_pad_k(m), kBN); }

}  // namespace

// tail of a weight pair (4 floats): 2^-wexp, max row 1-norm of the operand, max |bias|, wexp (int bits)
__global__ __launch_bounds__(1024) void mlp_weight_stats_kernel(const float* __restrict__ w,
                                                                const float* __restrict__ bias, int rows,
                                                                int cols, int transpose, int M, int K,
                                                                float* __restrict__ tail) {
  __shared__ float red[2][16];
  float amax = 0.f, l1 = 0.f;
  for (int m = threadIdx.x; m < M; m += blockDim.x) {   // operand row m: W[m, :] or, transposed, W[:, m]
    float s = 0.f;
    for (int k = 0; k < K; ++k) {
      const float v = fabsf(transpose ? w[static_cast<size_t>(k) * cols + m] : w[static_cast<size_t>(m) * cols + k]);
      s += v;
      amax = fmaxf(amax, v);
    }
    l1 = fmaxf(l1, s);
  }
  float bm = 0.f;
  if (bias)
    for (int m = threadIdx.x; m < M; m += blockDim.x) bm = fmaxf(bm, fabsf(bias[m]));
  l1 = fmaxf(l1, 0.f);
  for (int off = 32; off > 0; off >>= 1) {
    amax = fmaxf(amax, __shfl_xor(amax, off));
    l1 = fmaxf(l1, __shfl_xor(l1, off));
    bm = fmaxf(bm, __shfl_xor(bm, off));
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __shared__ float red3[16];
  if (lane == 0) { red[0][wave] = amax; red[1][wave] = l1; red3[wave] = bm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < static_cast<int>(blockDim.x >> 6); ++i) {
      amax = fmaxf(amax, red[0][i]); l1 = fmaxf(l1, red[1][i]); bm = fmaxf(bm, red3[i]);
    }
    const int e = scale_exp(amax);
    tail[0] = ldexpf(1.f, -e);
    tail[1] = l1;
    tail[2] = bm;
    tail[3] = __int_as_float(e);
  }
  (void)rows;
}

// W [rows, cols] fp32 (or its transpose) -> [2][Mp][Kp] fp16 of W 2^wexp, zero padded
__global__ void mlp_split_weights_f16_kernel(const float* __restrict__ w, int cols, int transpose, int M, int K,
                                             int Mp, int Kp, const float* __restrict__ tail,
                                             _Float16* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Mp * Kp) return;
  const int e = __float_as_int(tail[3]);
  const int m = idx / Kp, k = idx % Kp;
  float v = 0.f;
  if (m < M && k < K) v = transpose ? w[static_cast<size_t>(k) * cols + m] : w[static_cast<size_t>(m) * cols + k];
  _Float16 hi, lo;
  split2(ldexpf(v, e), hi, lo);
  out[idx] = hi;
  out[static_cast<size_t>(Mp) * Kp + idx] = lo;
}

// X [N, F] fp32 -> row pair: one wave per row (its max sets the row's exponent), NP x 64 float4 lanes
template <int NP>
__global__ __launch_bounds__(256) void mlp_split_rows_f16_kernel(const float* __restrict__ x, int N, int F, int Kp,
                                                                 _Float16* __restrict__ out,
                                                                 int32_t* __restrict__ rexp) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave_id();
  if (row >= N) return;
  const int nq = F >> 2, nqp = Kp >> 2;
  f32x4 v[NP];
  float m = 0.f;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int q = lane + 64 * p;
    v[p] = q < nq ? ldg4(x + static_cast<size_t>(row) * F + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) m = fmaxf(m, fabsf(v[p][k]));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  const int e = scale_exp(m);
  const size_t img = static_cast<size_t>(N) * Kp;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int q = lane + 64 * p;
    if (q < nqp) {   // (columns F .. Kp: zeros)
      f16x4 h, l;
      split2x4(v[p], e, h, l);
      const size_t o = static_cast<size_t>(row) * Kp + 4 * q;
      *reinterpret_cast<f16x4*>(out + o) = h;
      *reinterpret_cast<f16x4*>(out + img + o) = l;
    }
  }
  if (lane == 0) rexp[row] = e;
}

// C = epi(A W^T + bias) in TRUE units.  A2: row pair [2][N][Kp] + a_exp[N];  W2: [2][Mp][Kp] + wtail.
//   OUT_SPLIT false: C fp32 [N, M].   true: C as a row pair [2][N][Mq] + c_exp[N], Mq = pad_k(M)
//   EPI 0: none, 1: ReLU, 2: multiply by [mask_hi > 0] (mask_hi: the hi image of a ReLU output, [N][Mq])
// Staging and fragment layout: see csrc/mlp.hip (the same LDS-DMA double buffer with XOR-swizzled 64-byte
// rows; 2 images per operand instead of 3).
// S = LDS stages: S - 1 k-steps of operands are in flight while one is multiplied. A workgroup's k-steps are
// sequential round trips to L2 (1 - 2 us each against 0.2 us of MFMAs); with the 400 - 800 workgroups of a
// 512-molecule batch a CU holds one to three of them, so the depth of each workgroup's own pipeline, not the
// number of co-resident workgroups, is what keeps requests in flight (S = 3: 78 KB, two workgroups per CU).
template <int EPI, bool OUT_SPLIT, int S>
__global__ __launch_bounds__(kThreads, S >= 3 ? 2 : 3) void mlp_gemm_f16x2_kernel(const _Float16* __restrict__ A2,
                                                                     const int32_t* __restrict__ a_exp,
                                                                     const _Float16* __restrict__ W2,
                                                                     const float* __restrict__ wtail,
                                                                     const float* __restrict__ bias,
                                                                     const _Float16* __restrict__ mask_hi,
                                                                     float* __restrict__ C,
                                                                     _Float16* __restrict__ C2,
                                                                     int32_t* __restrict__ c_exp, int N, int Kp,
                                                                     int M, int Mp, int Mq) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
  // XCD-aware tile order: the column tiles of ONE node tile on consecutive workgroups of ONE XCD
  const int ncol = Mp / kBN;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rt = (slot / ncol) * 8 + xcd;               // node tile
  if (rt * kBM >= N) return;                            // (grid rounded up to 8 node tiles per group)
  const int row0 = rt * kBM, col0 = (slot % ncol) * kBN;
  const size_t aimg = static_cast<size_t>(N) * Kp, wimg = static_cast<size_t>(Mp) * Kp;

  const int crow = lane >> 2;
  const int cq = (lane & 3) ^ ((4 - (crow >> 2)) & 3);   // the 16-byte piece that belongs in this lane's LDS slot
  // every wave issues exactly kChunksPerWave DMA instructions per stage (the last two slots of waves 2 / 3
  // repeat their previous chunk: same bytes, same place), so that "all but the newest stage have landed"
  // is one vmcnt value for all waves
  auto dma = [&](int buf, int k0) {
    char* base = smem + buf * kBufB;
#pragma unroll
    for (int j = 0; j < kChunksPerWave; ++j) {
      int c = wave + 4 * j;
      if (c >= kAChunks + kBChunks) c -= 4;
      if (c < kAChunks) {                     // wave-uniform
        const int s = c / (kBM / 16), r16 = c % (kBM / 16);
        const int row = min(row0 + 16 * r16 + crow, N - 1);   // (rows past N: a valid row, never stored)
        lds_dma16(A2 + s * aimg + static_cast<size_t>(row) * Kp + k0 + 8 * cq, base + c * 1024);
      } else {
        const int cb = c - kAChunks, s = cb / (kBN / 16), r16 = cb % (kBN / 16);
        lds_dma16(W2 + s * wimg + static_cast<size_t>(col0 + 16 * r16 + crow) * Kp + k0 + 8 * cq, base + c * 1024);
      }
    }
  };

  f32x4 acc[2][kNT];     // [node tile][feature tile]: D[i = feature][j = node]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < kNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fslot = (lane >> 4) ^ ((4 - (frow >> 2)) & 3);
  const int f_off = frow * 64 + 16 * fslot;
  const int a_off = 2 * wave * 1024 + f_off;            // this wave's 32 nodes = chunks 2 wave, 2 wave + 1 of an image

  const int nk = Kp / kBK;
#pragma unroll
  for (int st = 0; st < S - 1; ++st)
    if (st < nk) dma(st, st * kBK);                      // (wave-uniform)
  // epilogue operands that do not depend on the products: requested now, used after the k loop
  int ea[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) ea[i] = a_exp[min(row0 + 32 * wave + 16 * i + (lane & 15), N - 1)];
  const float winv = wtail[0], l1max = wtail[1], bmax = wtail[2];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the first stages and these few words; then the count is the stages')
  int buf = 0, nbuf = (S - 1) % S;
  for (int k = 0; k < nk; ++k) {
    // this wave's pieces of step k have landed (newer stages may still be in flight) ...
    if (S >= 3 && k + 1 < nk && k > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kChunksPerWave * (S - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_only_barrier();                                  // ... and so have everybody else's; step k - 1 is read out
    if (k + S - 1 < nk) dma(nbuf, (k + S - 1) * kBK);    // (wave-uniform) into the buffer step k - 1 has just left
    const char* la = smem + buf * kBufB;
    const char* lb = la + kAChunks * 1024;
    f16x8 af[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        af[i][s] = *reinterpret_cast<const f16x8*>(la + s * (kBM / 16) * 1024 + a_off + i * 1024);
#pragma unroll
    for (int j = 0; j < kNT; ++j) {
      f16x8 wf[2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
        wf[s] = *reinterpret_cast<const f16x8*>(lb + (s * (kBN / 16) + j) * 1024 + f_off);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x4 c = acc[i][j];   // smallest terms first
        c = mfma_f16(wf[0], af[i][1], c);   // W hi * A lo
        c = mfma_f16(wf[1], af[i][0], c);   // W lo * A hi
        c = mfma_f16(wf[0], af[i][0], c);   // hi * hi
        acc[i][j] = c;
      }
    }
    buf = buf + 1 == S ? 0 : buf + 1;
    nbuf = nbuf + 1 == S ? 0 : nbuf + 1;
  }

  // epilogue: lane l holds features f0 .. f0 + 3 (f0 = col0 + 16 j + 4 (l >> 4)) of node row0 + 32 wave + 16 i + (l & 15)
  float sc[2];
  int eh[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    sc[i] = ldexpf(winv, -ea[i]);                        // products of scaled operands -> true units
    // the output row's exponent from a bound every column tile agrees on (file header)
    eh[i] = scale_exp(ldexpf(kPairTop * l1max, -ea[i]) + (EPI == 1 ? bmax : 0.f));
  }
#pragma unroll
  for (int j = 0; j < kNT; ++j) {
    const int f0 = col0 + 16 * j + 4 * (lane >> 4);
    f32x4 bc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int v = 0; v < 4; ++v) bc[v] = f0 + v < M ? bias[f0 + v] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = row0 + 32 * wave + 16 * i + (lane & 15);
      if (r >= N) continue;
      f32x4 y = acc[i][j] * sc[i] + bc;
      if (EPI == 1) {
#pragma unroll
        for (int v = 0; v < 4; ++v) y[v] = y[v] > 0.f ? y[v] : 0.f;
      }
      if (EPI == 2 && f0 < Mq) {
        const f16x4 mk = *reinterpret_cast<const f16x4*>(mask_hi + static_cast<size_t>(r) * Mq + f0);
#pragma unroll
        for (int v = 0; v < 4; ++v) y[v] = mk[v] > static_cast<_Float16>(0.f) ? y[v] : 0.f;
      }
      if (!OUT_SPLIT) {
        if (f0 + 3 < M) {
          stg4(C + static_cast<size_t>(r) * M + f0, y);
        } else {
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (f0 + v < M) C[static_cast<size_t>(r) * M + f0 + v] = y[v];
        }
      } else if (f0 < Mq) {   // (Mq is a multiple of 4: the 4 features are inside the padded row or outside it)
#pragma unroll
        for (int v = 0; v < 4; ++v) y[v] = f0 + v < M ? y[v] : 0.f;
        f16x4 h, l;
        split2x4(y, eh[i], h, l);
        const size_t img = static_cast<size_t>(N) * Mq, o = static_cast<size_t>(r) * Mq + f0;
        *reinterpret_cast<f16x4*>(C2 + o) = h;
        *reinterpret_cast<f16x4*>(C2 + img + o) = l;
        if (col0 == 0 && j == 0 && (lane >> 4) == 0) c_exp[r] = eh[i];
      }
    }
  }
}

}  // namespace gconv

using namespace gconv;

extern "C" size_t gconv_mlp_weights_pair_bytes(int32_t out_features, int32_t in_features) {
  if (out_features <= 0 || in_features <= 0) return 0;
  return static_cast<size_t>(2) * pad_m(out_features) * mlp_pad_k(in_features) * sizeof(_Float16) + 16;
}

extern "C" size_t gconv_mlp_rows_pair_bytes(int32_t num_rows, int32_t features) {
  if (num_rows < 0 || features <= 0) return 0;
  return mlp_pair_bytes(num_rows, features);
}

extern "C" gconv_status_t gconv_mlp_split_weights_f16x2(const float* w, const float* bias, int32_t rows,
                                                        int32_t cols, int32_t transpose, void* out,
                                                        gconv_stream_t stream_) {
  if (!w || !out || rows <= 0 || cols <= 0) return GCONV_E_INVALID_ARG;
  if (!aligned16(out)) return GCONV_E_ALIGNMENT;
  const int M = transpose ? cols : rows, K = transpose ? rows : cols;
  const int Mp = pad_m(M), Kp = mlp_pad_k(K);
  const int total = Mp * Kp;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  _Float16* o = static_cast<_Float16*>(out);
  float* tail = reinterpret_cast<float*>(o + static_cast<size_t>(2) * total);
  mlp_weight_stats_kernel<<<1, 1024, 0, stream>>>(w, bias, rows, cols, transpose ? 1 : 0, M, K, tail);
  GCONV_LAUNCH_CHECK();
  mlp_split_weights_f16_kernel<<<(total + 255) / 256, 256, 0, stream>>>(w, cols, transpose ? 1 : 0, M, K, Mp, Kp,
                                                                         tail, o);
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

extern "C" gconv_status_t gconv_mlp_split_rows_f16x2(const float* x, int32_t num_rows, int32_t features,
                                                     void* out, gconv_stream_t stream_) {
  if (!x || !out || num_rows < 0 || features <= 0) return GCONV_E_INVALID_ARG;
  if (features % 4 != 0 || features > 1024) return GCONV_E_UNSUPPORTED;
  if (!aligned16(x) || !aligned16(out)) return GCONV_E_ALIGNMENT;
  if (num_rows == 0) return GCONV_OK;
  const int Kp = mlp_pad_k(features);
  _Float16* o = static_cast<_Float16*>(out);
  int32_t* rexp = reinterpret_cast<int32_t*>(static_cast<char*>(out) + mlp_pair_exp_offset(num_rows, features));
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int grid = ceil_div(num_rows, 4);
  switch (ceil_div(Kp, 256)) {
    case 1: mlp_split_rows_f16_kernel<1><<<grid, 256, 0, stream>>>(x, num_rows, features, Kp, o, rexp); break;
    case 2: mlp_split_rows_f16_kernel<2><<<grid, 256, 0, stream>>>(x, num_rows, features, Kp, o, rexp); break;
    case 3: mlp_split_rows_f16_kernel<3><<<grid, 256, 0, stream>>>(x, num_rows, features, Kp, o, rexp); break;
    default: mlp_split_rows_f16_kernel<4><<<grid, 256, 0, stream>>>(x, num_rows, features, Kp, o, rexp); break;
  }
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

extern "C" gconv_status_t gconv_mlp_gemm_f16x2(const void* a_pair, const void* w_pair, const float* bias,
                                               const void* mask_pair, float* c_f32, void* c_pair,
                                               int32_t num_rows, int32_t in_features, int32_t out_features,
                                               int32_t epilogue, gconv_stream_t stream_) {
  if (!a_pair || !w_pair || num_rows < 0 || in_features <= 0 || out_features <= 0) return GCONV_E_INVALID_ARG;
  if (epilogue < 0 || epilogue > 2 || (epilogue == 2 && !mask_pair)) return GCONV_E_INVALID_ARG;
  if ((c_f32 == nullptr) == (c_pair == nullptr)) return GCONV_E_INVALID_ARG;   // exactly one output form
  if (!aligned16(a_pair) || !aligned16(w_pair) || (c_f32 && !aligned16(c_f32)) || (c_pair && !aligned16(c_pair)) ||
      (mask_pair && !aligned16(mask_pair)))
    return GCONV_E_ALIGNMENT;
  if (c_f32 && out_features % 4 != 0) return GCONV_E_UNSUPPORTED;
  if (num_rows == 0) return GCONV_OK;
  const int Kp = mlp_pad_k(in_features), Mp = pad_m(out_features), Mq = mlp_pad_k(out_features);
  const int row_tiles = (num_rows + kBM - 1) / kBM;
  const dim3 grid(((row_tiles + 7) / 8) * (Mp / kBN) * 8);   // 1-D: the kernel maps it to tiles XCD-aware
  // pipeline depth (kernel comment): three stages unless GCONV_MLP_STAGES=2 asks for round 3's two
  static const int stages = [] { const char* v = getenv("GCONV_MLP_STAGES"); return v && v[0] == '2' ? 2 : 3; }();
  const size_t lds = static_cast<size_t>(stages) * kBufB;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const _Float16* a = static_cast<const _Float16*>(a_pair);
  const int32_t* aexp = reinterpret_cast<const int32_t*>(static_cast<const char*>(a_pair) +
                                                         mlp_pair_exp_offset(num_rows, in_features));
  const _Float16* w = static_cast<const _Float16*>(w_pair);
  const float* wtail = reinterpret_cast<const float*>(w + static_cast<size_t>(2) * Mp * Kp);
  const _Float16* mk = static_cast<const _Float16*>(mask_pair);
  _Float16* c2 = static_cast<_Float16*>(c_pair);
  int32_t* cexp = c_pair ? reinterpret_cast<int32_t*>(static_cast<char*>(c_pair) +
                                                      mlp_pair_exp_offset(num_rows, out_features))
                         : nullptr;
#define GCONV_MLP_LAUNCH_S(EPI_, SPLIT_, S_)                                                                       \
  do {                                                                                                             \
    static bool opted_in = false; /* (dynamic LDS above the 48 KB default; once per kernel) */                      \
    if (!opted_in) {                                                                                               \
      GCONV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_gemm_f16x2_kernel<EPI_, SPLIT_, S_>),       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));           \
      opted_in = true;                                                                                             \
    }                                                                                                              \
    mlp_gemm_f16x2_kernel<EPI_, SPLIT_, S_><<<grid, kThreads, lds, stream>>>(a, aexp, w, wtail, bias, mk, c_f32,   \
                                                                            c2, cexp, num_rows, Kp, out_features, \
                                                                            Mp, Mq);                               \
  } while (0)
#define GCONV_MLP_LAUNCH(EPI_, SPLIT_)               \
  do {                                               \
    if (stages == 3) GCONV_MLP_LAUNCH_S(EPI_, SPLIT_, 3); \
    else GCONV_MLP_LAUNCH_S(EPI_, SPLIT_, 2);        \
  } while (0)
  if (c_pair) {
    switch (epilogue) {
      case 0: GCONV_MLP_LAUNCH(0, true); break;
      case 1: GCONV_MLP_LAUNCH(1, true); break;
      default: GCONV_MLP_LAUNCH(2, true); break;
    }
  } else {
    switch (epilogue) {
      case 0: GCONV_MLP_LAUNCH(0, false); break;
      case 1: GCONV_MLP_LAUNCH(1, false); break;
      default: GCONV_MLP_LAUNCH(2, false); break;
    }
  }
#undef GCONV_MLP_LAUNCH
#undef GCONV_MLP_LAUNCH_S
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}
