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
// fp32 product — accumulated in fp32. fp32-accurate, not fp32-exact: the layer's parity bar
// (1e-5 against the fp64 restatement, tests/test_gin.py) is the gate, `GCONV_TORCH_MLP=1` the knob.
//
// Splitting is done ONCE per value, never inside a GEMM's inner loop (the first version split the
// activation tile while staging it — ten VALU operations per element and k-step, repeated by every
// column tile):
//   weights  gconv_mlp_split_weights_bf16x3, once per frozen weight
//   inputs   gconv_mlp_split_rows_bf16x3: one pass over agg (forward) / dY (backward)
//   hidden   the first GEMM of each direction WRITES its output as a bf16 triple (epilogue: bias +
//            ReLU, resp. the ReLU mask read off the saved triple's hi image) — the hidden
//            activation never exists in fp32.
//   forward   H3 = split(relu(A3 W1^T + b1)) ;  Y  = H3 W2^T + b2
//   backward  G3 = split((dY3 W2) . [H > 0]) ;  dA = G3 W1
//
// One GEMM kernel: a workgroup (4 waves) computes 128 nodes x 80 features; a wave owns 32 nodes x
// 80 features (2 x 5 accumulator tiles). The MFMA's row operand is the WEIGHT fragment, its column
// operand the activation fragment: a lane then holds 4 consecutive features of one node — one
// 16-byte fp32 store, or one 8-byte store per bf16 image. Per k-step of 32 the six images (3 x 128
// x 32 and 3 x 80 x 32 bf16) go from global memory to LDS by LDS-DMA, double-buffered, XOR-swizzled
// rows of 64 bytes (see the kernel).
// Measured on a 512-molecule batch (tools/lab/mlp_bench.py, forward + input gradient of one layer):
// library GEMMs 321-326 us; split while staging 239; split once 266-275 (each node tile's images
// were re-read by 8 column tiles on 8 different XCDs: 324 MB per GEMM past L2 at 4.4 TB/s);
// XCD-aware tile order 226; a second k-step of look-ahead 239 (190 VGPRs: two workgroups per CU
// instead of three) — reverted; 160-feature tiles 234-249.
#include "common.h"

namespace gconv {

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kBM = 128;                        // nodes per workgroup
#ifndef GCONV_MLP_BN
#define GCONV_MLP_BN 80
#endif
constexpr int kBN = GCONV_MLP_BN, kNT = kBN / 16;   // features per workgroup, feature tiles
constexpr int kBK = 32;
constexpr int kThreads = 256;
// LDS: per buffer the three activation images (kBM rows x 64 B) then the three weight images (kBN
// rows x 64 B), in 1 KB chunks of 16 rows — what one LDS-DMA wave instruction writes
constexpr int kAChunks = 3 * kBM / 16, kBChunks = 3 * kBN / 16;
constexpr int kBufB = (kAChunks + kBChunks) * 1024;
constexpr int kChunksPerWave = (kAChunks + kBChunks + 3) / 4;

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

// 16 bytes per lane, global -> LDS without a register in between: lane i's piece lands at dst + 16 i
// (dst wave-uniform). Counted in vmcnt; visible to other waves after the issuer's wait + a barrier.
__device__ __forceinline__ void lds_dma16(const void* src, void* dst) {
  __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void*>(
                                       reinterpret_cast<uintptr_t>(src)),
                                   reinterpret_cast<__attribute__((address_space(3))) void*>(
                                       static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dst))),
                                   16, 0, 0);
}

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

}  // namespace

// padded sizes: K (the contraction) to a multiple of 32; feature counts to a multiple of the
// feature tile that also covers the 32-padding a bf16-triple OUTPUT needs as the next GEMM's contraction
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int pad_k(int k) { return round_up(k, kBK); }
static inline int pad_m(int m) { return round_up(pad_k(m), kBN); }

// W [rows, cols] fp32 (or its transpose) -> three bf16 images [3][Mp][Kp], zero padded;
// (M, K) = (rows, cols), or (cols, rows) when transposed.
__global__ void mlp_split_weights_kernel(const float* __restrict__ w, int cols, int transpose,
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
}

// X [N, K] fp32 -> three bf16 images [3][N][Kp] (columns K .. Kp zero); 4 values per thread
__global__ void mlp_split_rows_kernel(const float* __restrict__ x, int N, int K, int Kp,
                                      unsigned short* __restrict__ out) {
  const int q = Kp / 4;
  const size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<size_t>(N) * q) return;
  const int r = static_cast<int>(idx / q), k = 4 * static_cast<int>(idx % q);
  const f32x4 v = k < K ? ldg4(x + static_cast<size_t>(r) * K + k) : f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x4 h, m, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned short hi, mid, lo;
    split3(v[e], hi, mid, lo);
    h[e] = static_cast<short>(hi); m[e] = static_cast<short>(mid); l[e] = static_cast<short>(lo);
  }
  const size_t img = static_cast<size_t>(N) * Kp, o = static_cast<size_t>(r) * Kp + k;
  *reinterpret_cast<bf16x4*>(out + o) = h;
  *reinterpret_cast<bf16x4*>(out + img + o) = m;
  *reinterpret_cast<bf16x4*>(out + 2 * img + o) = l;
}

// C = epi(A W^T + bias).  A3: [3][N][Kp] bf16 images, W3: [3][Mp][Kp].
//   OUT_SPLIT false: C fp32 [N, M].   true: C as bf16 images [3][N][Mq], Mq = pad_k(M) (columns M .. Mq zero)
//   EPI 0: none, 1: ReLU, 2: multiply by [mask_hi > 0] (mask_hi: the hi image of a ReLU output, [N][Mq])
//
// Staging: the six images of a k-step go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no
// staging registers, no ds_write instructions), 1 KB = 16 rows x 64 bytes per wave instruction, into
// one of TWO LDS buffers: the DMA of step k + 1 runs under the MFMAs of step k and ONE barrier per
// k-step is enough (a buffer is rewritten two steps after it was read). Rows are 64 bytes, not
// padded: the 16-byte piece q of row r sits in slot q ^ ((4 - (r >> 2)) & 3) of its row, which
// makes the 16-byte fragment reads of every ds_read_b128 lane group (rows 0-3 / 12-15 at one
// k-group, rows 4-11 at the next) fall into 16 different bank groups — applied on the SOURCE side:
// lane i of a DMA instruction always writes LDS slot i, so it fetches the piece that belongs there.
// (Round-3 lab, tools/lab/prof_mlp.sh: with register staging + ds_write_b128 + 80-byte rows the
// kernel's skeleton — no global loads, no MFMAs — alone took 30-34 of 48-59 us per GEMM.)
template <int EPI, bool OUT_SPLIT>
__global__ __launch_bounds__(kThreads, 2) void mlp_gemm_bf16x3_kernel(const unsigned short* __restrict__ A3,
                                                                      const unsigned short* __restrict__ W3,
                                                                      const float* __restrict__ bias,
                                                                      const unsigned short* __restrict__ mask_hi,
                                                                      float* __restrict__ C,
                                                                      unsigned short* __restrict__ C3, int N,
                                                                      int Kp, int M, int Mp, int Mq) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
  // XCD-aware tile order. Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share
  // an L2): the column tiles of ONE node tile are given to consecutive workgroups of ONE XCD, so the
  // node tile's images are fetched into that L2 once and re-read from it by the other column tiles.
  // Placement is a speed matter only: any dispatch order computes the same tiles.
  const int ncol = Mp / kBN;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rt = (slot / ncol) * 8 + xcd;               // node tile
  if (rt * kBM >= N) return;                            // (grid rounded up to 8 node tiles per group)
  const int row0 = rt * kBM, col0 = (slot % ncol) * kBN;
  const size_t aimg = static_cast<size_t>(N) * Kp, wimg = static_cast<size_t>(Mp) * Kp;

  // DMA roles: chunk c of a k-step = 16 rows of one image (c < kAChunks: activations, else weights);
  // wave w moves the chunks w, w + 4, ...; lane i = (row i >> 2 of the chunk, LDS slot i & 3)
  const int crow = lane >> 2;
  const int cq = (lane & 3) ^ ((4 - (crow >> 2)) & 3);   // the piece that belongs in this lane's slot
  auto dma = [&](int buf, int k0) {
    char* base = smem + buf * kBufB;
#pragma unroll
    for (int j = 0; j < kChunksPerWave; ++j) {
      const int c = wave + 4 * j;
      if (c < kAChunks) {                     // wave-uniform
        const int s = c / (kBM / 16), r16 = c % (kBM / 16);
        const int row = min(row0 + 16 * r16 + crow, N - 1);   // (rows past N: a valid row, never stored)
        const unsigned short* src = A3 + s * aimg + static_cast<size_t>(row) * Kp + k0 + 8 * cq;
        lds_dma16(src, base + c * 1024);
      } else if (c < kAChunks + kBChunks) {
        const int cb = c - kAChunks, s = cb / (kBN / 16), r16 = cb % (kBN / 16);
        const unsigned short* src = W3 + s * wimg + static_cast<size_t>(col0 + 16 * r16 + crow) * Kp + k0 + 8 * cq;
        lds_dma16(src, base + c * 1024);
      }
    }
  };

  f32x4 acc[2][kNT];     // [node tile][feature tile]: D[i = feature][j = node]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < kNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: lane l reads row (l & 15), k-group (l >> 4) of a 16-row slice of an image
  const int frow = lane & 15;
  const int fslot = (lane >> 4) ^ ((4 - (frow >> 2)) & 3);
  const int f_off = frow * 64 + 16 * fslot;
  const int a_off = 2 * wave * 1024 + f_off;            // this wave's 32 nodes = chunks 2 wave, 2 wave + 1 of an image

  dma(0, 0);
  int buf = 0;
  for (int k0 = 0; k0 < Kp; k0 += kBK, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of step k0 have landed ...
    lds_only_barrier();                                  // ... and so have everybody else's; step k0 - 1 is read out
    if (k0 + kBK < Kp) dma(buf ^ 1, k0 + kBK);           // (wave-uniform) under the MFMAs below
    const char* la = smem + buf * kBufB;
    const char* lb = la + kAChunks * 1024;
    bf16x8 af[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < 3; ++s)
        af[i][s] = *reinterpret_cast<const bf16x8*>(la + s * (kBM / 16) * 1024 + a_off + i * 1024);
#pragma unroll
    for (int j = 0; j < kNT; ++j) {
      bf16x8 wf[3];
#pragma unroll
      for (int s = 0; s < 3; ++s)
        wf[s] = *reinterpret_cast<const bf16x8*>(lb + (s * (kBN / 16) + j) * 1024 + f_off);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        // smallest terms first: the accumulator sees them before the dominant product
        f32x4 c = acc[i][j];
        c = mfma_bf16(wf[1], af[i][1], c);   // mid * mid
        c = mfma_bf16(wf[0], af[i][2], c);   // W hi  * A lo
        c = mfma_bf16(wf[2], af[i][0], c);   // W lo  * A hi
        c = mfma_bf16(wf[0], af[i][1], c);   // W hi  * A mid
        c = mfma_bf16(wf[1], af[i][0], c);   // W mid * A hi
        c = mfma_bf16(wf[0], af[i][0], c);   // hi * hi
        acc[i][j] = c;
      }
    }
  }

  // epilogue: lane l holds features f0 .. f0 + 3 (f0 = col0 + 16 j + 4 (l >> 4)) of node row0 + 32 wave + 16 i + (l & 15)
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
      f32x4 y = acc[i][j] + bc;
      if (EPI == 1) {
#pragma unroll
        for (int v = 0; v < 4; ++v) y[v] = y[v] > 0.f ? y[v] : 0.f;
      }
      if (EPI == 2 && f0 < Mq) {
        const bf16x4 mk = *reinterpret_cast<const bf16x4*>(mask_hi + static_cast<size_t>(r) * Mq + f0);
#pragma unroll
        for (int v = 0; v < 4; ++v) y[v] = mk[v] > 0 ? y[v] : 0.f;   // (bf16 bits as int16: > 0 <=> positive value)
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
        bf16x4 h, m, l;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          unsigned short hi, mid, lo;
          split3(f0 + v < M ? y[v] : 0.f, hi, mid, lo);
          h[v] = static_cast<short>(hi); m[v] = static_cast<short>(mid); l[v] = static_cast<short>(lo);
        }
        const size_t img = static_cast<size_t>(N) * Mq, o = static_cast<size_t>(r) * Mq + f0;
        *reinterpret_cast<bf16x4*>(C3 + o) = h;
        *reinterpret_cast<bf16x4*>(C3 + img + o) = m;
        *reinterpret_cast<bf16x4*>(C3 + 2 * img + o) = l;
      }
    }
  }
}

}  // namespace gconv

using namespace gconv;

extern "C" size_t gconv_mlp_split_bytes(int32_t out_features, int32_t in_features) {
  if (out_features <= 0 || in_features <= 0) return 0;
  return static_cast<size_t>(3) * pad_m(out_features) * pad_k(in_features) * sizeof(unsigned short);
}

extern "C" size_t gconv_mlp_rows_split_bytes(int32_t num_rows, int32_t features) {
  if (num_rows < 0 || features <= 0) return 0;
  return static_cast<size_t>(3) * (num_rows > 0 ? num_rows : 1) * pad_k(features) * sizeof(unsigned short);
}

extern "C" gconv_status_t gconv_mlp_split_weights_bf16x3(const float* w, int32_t rows, int32_t cols,
                                                         int32_t transpose, void* out,
                                                         gconv_stream_t stream_) {
  if (!w || !out || rows <= 0 || cols <= 0) return GCONV_E_INVALID_ARG;
  const int M = transpose ? cols : rows, K = transpose ? rows : cols;
  const int Mp = pad_m(M), Kp = pad_k(K);
  const int total = Mp * Kp;
  mlp_split_weights_kernel<<<(total + 255) / 256, 256, 0, static_cast<hipStream_t>(stream_)>>>(
      w, cols, transpose ? 1 : 0, M, K, Mp, Kp, static_cast<unsigned short*>(out));
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

extern "C" gconv_status_t gconv_mlp_split_rows_bf16x3(const float* x, int32_t num_rows, int32_t features,
                                                      void* out, gconv_stream_t stream_) {
  if (!x || !out || num_rows < 0 || features <= 0) return GCONV_E_INVALID_ARG;
  if (features % 4 != 0) return GCONV_E_UNSUPPORTED;
  if (!aligned16(x) || !aligned16(out)) return GCONV_E_ALIGNMENT;
  if (num_rows == 0) return GCONV_OK;
  const int Kp = pad_k(features);
  const size_t total = static_cast<size_t>(num_rows) * (Kp / 4);
  mlp_split_rows_kernel<<<static_cast<unsigned>((total + 255) / 256), 256, 0, static_cast<hipStream_t>(stream_)>>>(
      x, num_rows, features, Kp, static_cast<unsigned short*>(out));
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

extern "C" gconv_status_t gconv_mlp_gemm_bf16x3(const void* a_split, const void* w_split, const float* bias,
                                                const void* mask_split, float* c_f32, void* c_split,
                                                int32_t num_rows, int32_t in_features, int32_t out_features,
                                                int32_t epilogue, gconv_stream_t stream_) {
  if (!a_split || !w_split || num_rows < 0 || in_features <= 0 || out_features <= 0) return GCONV_E_INVALID_ARG;
  if (epilogue < 0 || epilogue > 2 || (epilogue == 2 && !mask_split)) return GCONV_E_INVALID_ARG;
  if ((c_f32 == nullptr) == (c_split == nullptr)) return GCONV_E_INVALID_ARG;   // exactly one output form
  if (!aligned16(a_split) || !aligned16(w_split) || (c_f32 && !aligned16(c_f32)) || (c_split && !aligned16(c_split)) ||
      (mask_split && !aligned16(mask_split)))
    return GCONV_E_ALIGNMENT;
  if (c_f32 && out_features % 4 != 0) return GCONV_E_UNSUPPORTED;
  if (num_rows == 0) return GCONV_OK;
  const int Kp = pad_k(in_features), Mp = pad_m(out_features), Mq = pad_k(out_features);
  const int row_tiles = (num_rows + kBM - 1) / kBM;
  const dim3 grid(((row_tiles + 7) / 8) * (Mp / kBN) * 8);   // 1-D: the kernel maps it to tiles XCD-aware
  const size_t lds = 2 * kBufB;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const unsigned short* a = static_cast<const unsigned short*>(a_split);
  const unsigned short* w = static_cast<const unsigned short*>(w_split);
  const unsigned short* mk = static_cast<const unsigned short*>(mask_split);
  unsigned short* c3 = static_cast<unsigned short*>(c_split);
#define GCONV_MLP_LAUNCH(EPI_, SPLIT_)                                                                             \
  do {                                                                                                             \
    static DeviceOnce opted_in; /* (two 39 KB buffers of dynamic LDS: above the 48 KB default; per kernel and device) */ \
    if (opted_in.need()) {                                                                                         \
      GCONV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_gemm_bf16x3_kernel<EPI_, SPLIT_>),          \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));           \
      opted_in.mark();                                                                                             \
    }                                                                                                              \
    mlp_gemm_bf16x3_kernel<EPI_, SPLIT_><<<grid, kThreads, lds, stream>>>(a, w, bias, mk, c_f32, c3, num_rows, Kp, \
                                                                         out_features, Mp, Mq);                    \
  } while (0)
  if (c_split) {
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
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}
