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
// 26 KB per stage instead of 39 (three stages: two workgroups per CU). Shapes the W-stationary kernel at the
// end of this file takes (K = 300 / 600 of the d = 300 GIN MLP) do not come here.
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


// ---- the same product with the WEIGHTS stationary in registers (round 4, second half) -------------------------
// The tiled kernel above lives between barriers: ten (K = 300) or nineteen (K = 600) k-steps per workgroup, each
// a 0.2 us burst of MFMAs behind a 1 - 2 us LDS-DMA round trip with two steps in flight, and the 400 - 800
// workgroups of a 512-molecule batch fill the chip's 512 slots unevenly. Here a workgroup keeps ITS columns of W
// — 64 NT features x the whole K, as MFMA fragments — in registers (NT KS 8 VGPRs per lane: 160 at K = 320 with
// 32 features per wave, 152 at K = 608 with 16) and walks a contiguous range of 16-node row blocks: per block ONE
// barrier, 3 NT KS MFMAs per wave (60 / 57), an epilogue, and R - 1 blocks of A (20 / 38 KB each: every k-step
// of both images, the 16 row exponents and, for EPI 2, the block's piece of the ReLU mask) in flight by LDS-DMA.
// Four LOADER waves issue and await that DMA: vmcnt is per wave and counts stores too — the compute waves' output
// stores would otherwise sit in the queue the next block's loads are awaited on — and the LDS-DMA rate of a CU
// grows with the number of waves issuing it (DMA alone, 103 k rows, K = 600: 141 us with two loader waves, 103
// with four). The compute waves read everything through LDS and wait for no vector-memory instruction inside
// the loop. Column groups (Mp / (64 NT) = 5 for both shapes) x as many row ranges as fill the chip once = 240
// workgroups of equal length; the groups of a range sit on one XCD, so A is fetched from memory once and read
// from L2 five times. Pair outputs at NT = 2 leave as 16-byte stores (v_permlane16_swap pairs the two tiles'
// 4-feature pieces of neighbouring row groups: 64 contiguous bytes per node and instruction instead of 32).
// The arithmetic per output element is the tiled kernel's, product for product: the two agree bit for bit
// (tests/test_gin.py). The previous block's epilogue is part of the current block's instruction stream (buffer
// stores that drop what lies past the row / the matrix: no branches), so its VALU work sits in the MFMAs' issue
// shadows, and the A fragments are read two k-steps ahead of their MFMAs (GCONV_MLP_WS_SCHED; hipcc's own order
// keeps them one step ahead: K = 600 at 13104 rows 24.7 -> 20.0 us).
// Measured (tools/lab/ws_gemm.py, one GEMM, 13104 / 102952 rows): K = 300 with ReLU, pair out 35.7 -> 25.9 /
// 237 -> 154 us; K = 600, fp32 out 27.4 -> 20.0 / 204 -> 159 us. Leave-one-out builds (GCONV_MLP_WS_LAB) at
// 13104 rows, K = 300: DMA, barriers and LDS reads alone 12.0, with the MFMAs 19.9, with the epilogue's
// arithmetic 23.2, with its stores 25.6 us — the streams add up more than they overlap; the 5 x 16 MB of A
// through L2 at the 40 - 70 GB/s per CU that LDS-DMA delivers and the 31 MB of pair output are what the kernel
// waits for, not the 6 us of MFMAs.
#ifndef GCONV_MLP_WS_SCHED
#define GCONV_MLP_WS_SCHED 2   // A fragments this many k-steps ahead (0: the compiler's own order)
#endif
#ifndef GCONV_MLP_WS_LAB
#define GCONV_MLP_WS_LAB 0   // leave-one-out builds (results wrong by construction): 1 no MFMAs, 2 no epilogue,
#endif                       // 4 no stores, 8 no DMA after the prologue, 16 no LDS fragment reads
constexpr int kWsComputeWaves = 4, kWsLoaderWaves = 4;
constexpr int kWsThreads = 64 * (kWsComputeWaves + kWsLoaderWaves);
constexpr int ws_ring(int KS) { return KS <= 10 ? 6 : 4; }                   // row blocks of A in LDS
// one ring slot: the block's A chunks (16 rows x 64 B each: 2 images x KS k-steps), its 16 row exponents (one
// 256-byte dword DMA), and for EPI 2 the block's piece of the ReLU mask (16 rows x 64 NT features, fp16)
constexpr int ws_mask_chunks(int NT, int EPI) { return EPI == 2 ? 2 * NT : 0; }
constexpr int ws_slot_bytes(int KS, int NT, int EPI) { return (2 * KS + ws_mask_chunks(NT, EPI)) * 1024 + 256; }
constexpr int ws_items(int KS, int NT, int EPI) { return 2 * KS + 1 + ws_mask_chunks(NT, EPI); }
constexpr int ws_items_per_loader(int KS, int NT, int EPI) {
  return (ws_items(KS, NT, EPI) + kWsLoaderWaves - 1) / kWsLoaderWaves;
}
constexpr bool ws_supported(int KS, int NT, int EPI) {
  return ws_ring(KS) * ws_slot_bytes(KS, NT, EPI) <= 160 * 1024 &&
         ws_items_per_loader(KS, NT, EPI) * (ws_ring(KS) - 2) <= 63;          // (vmcnt is a 6-bit count)
}

template <int KS, int NT, int EPI, bool OUT_SPLIT>
__global__ __launch_bounds__(kWsThreads, 1) void mlp_gemm_f16x2_ws_kernel(const _Float16* __restrict__ A2,
                                                                         const int32_t* __restrict__ a_exp,
                                                                         const _Float16* __restrict__ W2,
                                                                         const float* __restrict__ wtail,
                                                                         const float* __restrict__ bias,
                                                                         const _Float16* __restrict__ mask_hi,
                                                                         float* __restrict__ C,
                                                                         _Float16* __restrict__ C2,
                                                                         int32_t* __restrict__ c_exp, int N, int M,
                                                                         int Mp, int Mq, int CG, int rr_per_xcd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int Kp = KS * kBK, R = ws_ring(KS), kSlotB = ws_slot_bytes(KS, NT, EPI);
  constexpr int kMaskOff = 2 * KS * 1024, kMaskChunks = ws_mask_chunks(NT, EPI), kExpOff = kMaskOff + kMaskChunks * 1024;
  constexpr int kItems = ws_items(KS, NT, EPI), kDma = ws_items_per_loader(KS, NT, EPI);
  static_assert(ws_supported(KS, NT, EPI), "ring does not fit LDS / vmcnt");
  const int lane = threadIdx.x & 63, wave = wave_id();
  // the column groups of one row range on consecutive workgroups of ONE XCD: A comes from memory once per range
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int cg = slot % CG, rr = xcd * rr_per_xcd + slot / CG, RR = 8 * rr_per_xcd;
  const int nb_total = (N + 15) >> 4;
  const int b0 = static_cast<int>(static_cast<long long>(rr) * nb_total / RR);
  const int nb = static_cast<int>(static_cast<long long>(rr + 1) * nb_total / RR) - b0;
  if (nb <= 0) return;                                     // (workgroup-uniform, before any barrier)
  const size_t aimg = static_cast<size_t>(N) * Kp;

  if (wave >= kWsComputeWaves) {   // ---- loader waves: items [lw kDma, (lw + 1) kDma) of every block ----------
    const int lw = wave - kWsComputeWaves;
    const int crow = lane >> 2;
    const int cq = (lane & 3) ^ ((4 - (crow >> 2)) & 3);   // the 16-byte piece that belongs in this lane's LDS slot
    auto dma = [&](int blk) {
      char* base = smem + (blk % R) * kSlotB;
      const int row0 = (b0 + blk) * 16;
      const int row = min(row0 + crow, N - 1);             // (rows past N: a valid row, never stored)
      const _Float16* src = A2 + static_cast<size_t>(row) * Kp + 8 * cq;
#pragma unroll
      for (int j = 0; j < kDma; ++j) {
        const int c = min(lw * kDma + j, kItems - 1);      // (the odd item out is issued twice: same bytes, same place)
        if (c < 2 * KS) {                                  // A: image c / KS, k-step c % KS
          lds_dma16(src + (c / KS) * aimg + (c % KS) * kBK, base + c * 1024);
        } else if (c < 2 * KS + kMaskChunks) {             // mask: LDS piece p = 64 m + lane <- (row p / (8 NT), 16-byte
          const int pp = (c - 2 * KS) * 64 + lane;         // piece (p % (8 NT)) ^ (row % (8 NT)) of the row's 128 NT bytes)
          const int mrow = pp / (8 * NT), piece = (pp % (8 * NT)) ^ (mrow % (8 * NT));
          const int col = min(cg * (64 * NT) + 8 * piece, Mq - 8);   // (past the padded row: any valid piece, never used)
          lds_dma16(mask_hi + static_cast<size_t>(min(row0 + mrow, N - 1)) * Mq + col, base + c * 1024);
        } else {                                           // the 16 row exponents (lanes 16 .. 63: repeats)
          __builtin_amdgcn_global_load_lds(
              reinterpret_cast<const __attribute__((address_space(1))) void*>(
                  reinterpret_cast<uintptr_t>(a_exp + min(row0 + (lane & 15), N - 1))),
              reinterpret_cast<__attribute__((address_space(3))) void*>(
                  static_cast<uint32_t>(reinterpret_cast<uintptr_t>(base + kExpOff))),
              4, 0, 0);
        }
      }
    };
    for (int j = 0; j < R - 1; ++j)
      if (j < nb) dma(j);
    for (int i = 0; i < nb; ++i) {
      // block i has landed (R - 2 newer ones may be in flight; towards the end fewer were issued: wait for all)
      if (i + R - 2 < nb) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma * (R - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                        // ... and block i - 1 is read out:
      if (i + R - 1 < nb && !(GCONV_MLP_WS_LAB & 8)) dma(i + R - 1);   // its slot takes block i + R - 1
    }
    return;
  }

  // ---- compute waves: features colbase .. colbase + 16 NT of every row block ---------------------------------
  // (their only vector-memory instructions inside the loop are the output stores: everything they read arrives
  // through LDS, so nothing in the loop waits for a store to be acknowledged)
  const int colbase = cg * (64 * NT) + wave * (16 * NT);
  const int frow = lane & 15, fgrp = lane >> 4;
  f16x8 wf[NT][KS][2];
  {
    const size_t wimg = static_cast<size_t>(Mp) * Kp;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          wf[t][k][s] = *reinterpret_cast<const f16x8*>(W2 + s * wimg + static_cast<size_t>(colbase + 16 * t + frow) * Kp +
                                                         k * kBK + 8 * fgrp);
  }
  const float winv = wtail[0], l1max = wtail[1], bmax = wtail[2];
  f32x4 bc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int f0 = colbase + 16 * t + 4 * fgrp;
#pragma unroll
    for (int v = 0; v < 4; ++v) bc[t][v] = bias && f0 + v < M ? bias[f0 + v] : 0.f;
  }
  // the fragments are in their registers before the loop starts (an opaque use: the compiler waits for the loads
  // HERE, once, instead of counting them down inside the loop, and cannot re-load them per block)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      asm volatile("" : "+v"(wf[t][k][0]), "+v"(wf[t][k][1]));
    }
    asm volatile("" : "+v"(bc[t]));
  }
  const int f_off = frow * 64 + 16 * (fgrp ^ ((4 - (frow >> 2)) & 3));
  // this lane's 8 bytes of the mask piece: features 16 (NT wave + t) + 4 fgrp .. + 3 of the group's 64 NT
  int mk_off[NT > 0 ? NT : 1];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int piece = (2 * (NT * wave + t)) + (fgrp >> 1);
    mk_off[t] = kMaskOff + (frow * (8 * NT) + (piece ^ (frow % (8 * NT)))) * 16 + 8 * (fgrp & 1);
  }
  // Outputs through buffer descriptors: a row past N, a feature past the padded row or "no block yet" is an
  // offset beyond the descriptor and the store is dropped — the epilogue has no branches, so it shares one basic
  // block with the NEXT block's MFMAs and the scheduler can put its VALU and store instructions into their
  // issue gaps (an MFMA holds the vector issue for 8 of its 16 cycles). One wave per SIMD computes: nothing
  // else would fill those gaps.
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const rsrc_t rc = OUT_SPLIT ? make_rsrc(C2, static_cast<uint32_t>(static_cast<size_t>(4) * N * Mq))
                              : make_rsrc(C, static_cast<uint32_t>(static_cast<size_t>(4) * N * M));
  const rsrc_t re = make_rsrc(c_exp, OUT_SPLIT ? 4u * N : 0u);
  const int img_b = 2 * N * Mq;                              // bytes of one fp16 image
  auto epilogue = [&](const f32x4 (&acc)[NT], int r, int ea, const f16x4 (&mk)[NT]) {
    // lane l holds features f0 .. f0 + 3 (f0 = colbase + 16 t + 4 (l >> 4)) of node r (r < 0: nothing yet)
    const bool live = static_cast<unsigned>(r) < static_cast<unsigned>(N);
    const float sc = ldexpf(winv, -ea);                    // products of scaled operands -> true units
    const int eh = scale_exp(ldexpf(kPairTop * l1max, -ea) + (EPI == 1 ? bmax : 0.f));
    f16x4 ph[NT], pl[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int f0 = colbase + 16 * t + 4 * fgrp;
      f32x4 y = acc[t] * sc + bc[t];
      if (EPI == 1) {
#pragma unroll
        for (int v = 0; v < 4; ++v) y[v] = y[v] > 0.f ? y[v] : 0.f;
      }
      if (EPI == 2) {
#pragma unroll
        for (int v = 0; v < 4; ++v) y[v] = mk[t][v] > static_cast<_Float16>(0.f) ? y[v] : 0.f;
      }
      if (!OUT_SPLIT) {   // (M is a multiple of 4: the 4 features are inside the row or outside it)
        int off = live && f0 < M && !(GCONV_MLP_WS_LAB & 4) ? (r * M + f0) * 4 : kBufOob;
        asm volatile("" : "+v"(off));   // (opaque: a select, not a branch around the store)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, y), rc, off, 0, 0);
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) y[v] = f0 + v < M ? y[v] : 0.f;
        split2x4(y, eh, ph[t], pl[t]);
        if (NT == 1) {    // (Mq is a multiple of 4)
          int off = live && f0 < Mq && !(GCONV_MLP_WS_LAB & 4) ? (r * Mq + f0) * 2 : kBufOob;
          asm volatile("" : "+v"(off));
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ph[t]), rc, off, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pl[t]), rc, off + img_b, 0, 0);
        }
      }
    }
    if (OUT_SPLIT && NT == 2) {
      // v_permlane16_swap: the odd 16-lane rows of the first operand <-> the even rows of the second. With
      // (tile 0, tile 1) as the operands, row groups 0 / 2 end up with features 4 g .. 4 g + 7 of tile 0 and row
      // groups 1 / 3 with features 4 (g - 1) .. 4 (g - 1) + 7 of tile 1: every store instruction writes 64
      // contiguous bytes per node instead of 32
      u32x4 out[2];
#pragma unroll
      for (int img = 0; img < 2; ++img) {
        const u32x2 a = __builtin_bit_cast(u32x2, img ? pl[0] : ph[0]);
        const u32x2 b = __builtin_bit_cast(u32x2, img ? pl[NT - 1] : ph[NT - 1]);
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const auto sw = __builtin_amdgcn_permlane16_swap(a[w], b[w], false, false);
          out[img][w] = sw[0];
          out[img][2 + w] = sw[1];
        }
      }
      const int f8 = colbase + ((fgrp & 1) ? 16 + 4 * (fgrp - 1) : 4 * fgrp);   // (Mq is a multiple of 8)
      int off = live && f8 < Mq && !(GCONV_MLP_WS_LAB & 4) ? (r * Mq + f8) * 2 : kBufOob;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_buffer_store_b128(out[0], rc, off, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(out[1], rc, off + img_b, 0, 0);
    }
    if (OUT_SPLIT) {
      int off = live && colbase == 0 && fgrp == 0 ? 4 * r : kBufOob;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_buffer_store_b32(eh, re, off, 0, 0);
    }
  };

  f32x4 pacc[NT];                                          // the previous block's sums, exponents and mask:
  f16x4 pmk[NT];                                           // its epilogue runs inside this block's MFMAs
  int pr = -1, pea = 0, slot_b = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    pacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    pmk[t] = f16x4{0, 0, 0, 0};
  }
  for (int i = 0; i < nb; ++i) {
    __builtin_amdgcn_s_barrier();                          // block i is in LDS
    const char* sb = smem + slot_b * kSlotB;
    const char* la = sb + f_off;
    slot_b = slot_b + 1 == R ? 0 : slot_b + 1;
    const int ea = *reinterpret_cast<const int*>(sb + kExpOff + 4 * frow);
    f16x4 mk[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
      mk[t] = EPI == 2 ? *reinterpret_cast<const f16x4*>(sb + mk_off[t]) : f16x4{0, 0, 0, 0};
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // A fragments kAhead k-steps ahead of their MFMAs; the fences keep LDS reads and MFMAs in this order and let
    // everything else (the previous block's epilogue) move into the MFMAs' issue shadows
    constexpr int kAhead = GCONV_MLP_WS_SCHED;
    constexpr int kFence = 0x2 | 0x4 | 0x10;               // VALU, SALU and vector memory may cross
    f16x8 ah[KS], al[KS];
#pragma unroll
    for (int k = 0; k < kAhead && k < KS; ++k) {
      ah[k] = *reinterpret_cast<const f16x8*>(la + k * 1024);
      al[k] = *reinterpret_cast<const f16x8*>(la + (KS + k) * 1024);
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      if (k + kAhead < KS) {
        ah[k + kAhead] = *reinterpret_cast<const f16x8*>(la + (k + kAhead) * 1024);
        al[k + kAhead] = *reinterpret_cast<const f16x8*>(la + (KS + k + kAhead) * 1024);
      }
      if (kAhead > 0) __builtin_amdgcn_sched_barrier(kFence);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x4 c = acc[t];   // smallest terms first (as in the tiled kernel)
#if GCONV_MLP_WS_LAB & 1
        c[0] += static_cast<float>(al[k][0]) + static_cast<float>(ah[k][1]) + static_cast<float>(wf[t][k][0][0]) +
                static_cast<float>(wf[t][k][1][0]);
#else
        c = mfma_f16(wf[t][k][0], al[k], c);   // W hi * A lo
        c = mfma_f16(wf[t][k][1], ah[k], c);   // W lo * A hi
        c = mfma_f16(wf[t][k][0], ah[k], c);   // hi * hi
#endif
        acc[t] = c;
      }
      if (kAhead > 0) __builtin_amdgcn_sched_barrier(kFence);
    }
#if !(GCONV_MLP_WS_LAB & 2)
    epilogue(pacc, pr, pea, pmk);
#endif
    pr = (b0 + i) * 16 + frow;
    pea = ea;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      pacc[t] = acc[t];
      pmk[t] = mk[t];
    }
  }
#if GCONV_MLP_WS_LAB & 2
  asm volatile("" ::"v"(pacc[0]), "v"(pacc[NT - 1]), "v"(pea), "v"(pmk[0]));
#else
  epilogue(pacc, pr, pea, pmk);
#endif
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
  // weights stationary in registers (kernel comment) for the two contraction lengths of the GIN MLP (d = 300:
  // K = 300 with 32 features per wave, K = 600 with 16) when the chip holds a row range per column group;
  // GCONV_MLP_WS=0: the tiled kernel for everything (A/B)
  static const bool ws_on = [] { const char* v = getenv("GCONV_MLP_WS"); return !(v && v[0] == '0'); }();
  const int KS = Kp / kBK, NT = KS == 10 ? 2 : 1;
  // (its stores go through buffer descriptors with 32-bit offsets, "dropped" = an offset of 2^30 or more: outputs
  //  of 1 GiB and beyond take the tiled kernel)
  const bool ws_fits = static_cast<size_t>(num_rows) * (c_pair ? Mq : out_features) * 4 <= (static_cast<size_t>(1) << 30);
  if (ws_on && ws_fits && (KS == 10 || (KS == 19 && epilogue != 2)) && Mp % (64 * NT) == 0) {
    const int cus = device_cus();   // (per device, common.h)
    const int CG = Mp / (64 * NT), rr_per_xcd = (cus / 8) / CG;
    if (rr_per_xcd >= 1) {
      const dim3 wgrid(8 * rr_per_xcd * CG);
#define GCONV_MLP_WS_LAUNCH_K(KS_, NT_, EPI_, SPLIT_)                                                              \
  do {                                                                                                             \
    constexpr int ws_lds = ws_ring(KS_) * ws_slot_bytes(KS_, NT_, EPI_);                                                   \
    static DeviceOnce opted_in; /* (dynamic LDS above the 48 KB default: per kernel AND per device) */              \
    if (opted_in.need()) {                                                                                         \
      GCONV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_gemm_f16x2_ws_kernel<KS_, NT_, EPI_, SPLIT_>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, ws_lds));                          \
      opted_in.mark();                                                                                             \
    }                                                                                                              \
    mlp_gemm_f16x2_ws_kernel<KS_, NT_, EPI_, SPLIT_><<<wgrid, kWsThreads, ws_lds, stream>>>(                       \
        a, aexp, w, wtail, bias, mk, c_f32, c2, cexp, num_rows, out_features, Mp, Mq, CG, rr_per_xcd);             \
  } while (0)
#define GCONV_MLP_WS_LAUNCH(EPI_, SPLIT_)                      \
  do {                                                         \
    if (KS == 10) GCONV_MLP_WS_LAUNCH_K(10, 2, EPI_, SPLIT_);  \
    else if (EPI_ != 2) GCONV_MLP_WS_LAUNCH_K(19, 1, (EPI_ == 2 ? 0 : EPI_), SPLIT_); \
  } while (0)
      if (c_pair) {
        switch (epilogue) {
          case 0: GCONV_MLP_WS_LAUNCH(0, true); break;
          case 1: GCONV_MLP_WS_LAUNCH(1, true); break;
          default: GCONV_MLP_WS_LAUNCH(2, true); break;
        }
      } else {
        switch (epilogue) {
          case 0: GCONV_MLP_WS_LAUNCH(0, false); break;
          case 1: GCONV_MLP_WS_LAUNCH(1, false); break;
          default: GCONV_MLP_WS_LAUNCH(2, false); break;
        }
      }
#undef GCONV_MLP_WS_LAUNCH
#undef GCONV_MLP_WS_LAUNCH_K
      GCONV_LAUNCH_CHECK();
      return GCONV_OK;
    }
  }
#define GCONV_MLP_LAUNCH_S(EPI_, SPLIT_, S_)                                                                       \
  do {                                                                                                             \
    static DeviceOnce opted_in; /* (dynamic LDS above the 48 KB default; once per kernel and device) */             \
    if (opted_in.need()) {                                                                                         \
      GCONV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_gemm_f16x2_kernel<EPI_, SPLIT_, S_>),       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));           \
      opted_in.mark();                                                                                             \
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
