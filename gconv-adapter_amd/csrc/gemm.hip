// K1 / K3 / B1 / B4 / B5 — skinny fp32 projections on v_mfma_f32_16x16x4_f32.
//
// Replaces GCNConv.lin (bias-free nn.Linear) of conv_down / conv_up
// (/root/reference/src/finetune/gconv_adapter.py:40-41), the residual + scalar
// tail (:94-95,105-106) and their autograd backward.
//
// All four streaming kernels are HBM-bound (arithmetic intensity <= r/2
// flop/B): the [N, d] operands are touched once per kernel with 16-byte lane
// accesses, everything of rank width stays in registers / LDS / L2.
//
// Two register layouts for a 16-node block of an [N, d] matrix:
//   "A layout"  (contraction over d; K1, B1): lane (node = l & 15, kq = l >> 4)
//       holds float4 X[i0 + node][16 q + 4 kq .. +3] for k-step q; the MFMA k
//       index is permuted (k = 16 q + 4 kq + t for MFMA t), the weight
//       fragment uses the same permutation.
//   "L layout"  (contraction over rank or over nodes; K3, B4): one wave owns a
//       64-column chunk; lane (jj = l & 15, t = l >> 4) holds, for m = 0..3,
//       float4 M[i0 + 4 t + m][64 J + 4 jj .. +3]  (256 B contiguous per row).
//       The same registers are (i) the MFMA B operand for products that
//       contract over nodes (dW^T = Z^T M; k-step m, MFMA c uses component c)
//       and (ii) the element-wise operand of products that output [node, col]
//       tiles (D register m of MFMA c is element (i0 + 4 t + m, 64 J + 4 jj + c)).
#include <cstdlib>

#include "gemm.h"

namespace gconv {

// ===========================================================================
// K1 / B1: out[N, Fp] = epilogue(X[N, d] * Wt[d, Fp])
// ===========================================================================
// NQ = compile-time number of 16-wide k-steps (d <= 16 NQ). The whole 16-node x d tile
// of a block lives in NQ float4 registers per lane, used as a ring: as soon as k-step q of
// the current block has fed its MFMAs, the same register is re-loaded with k-step q of the
// wave's NEXT block, so NQ loads stay in flight per wave under straight-line code (no
// branch between a load and its use: hipcc can then emit counted vmcnt waits).
// Lab variants of the K1 / B1 memory schedule (tools/lab/r05_k1.sh; 0 = product): bit 0 ring refills issued in
// pairs (both halves of a 128-byte line back to back), bit 1 non-temporal X loads, bit 2 non-temporal `out` stores,
// bit 3 no `out` stores at all (a bound; wrong results)
#ifndef GCONV_LAB_ROWSA
#define GCONV_LAB_ROWSA 0
#endif
__device__ __forceinline__ f32x4 rowsA_ldg4(const float* p) {
#if GCONV_LAB_ROWSA & 2
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
  return ldg4(p);
#endif
}

template <int NCB, int NQ, int DEPTH>
__global__ __launch_bounds__(256) void gemm_rowsA_kernel(RowsAArgs a) {
  extern __shared__ f32x4 wl[];  // [NQ][NCB][64] weight fragments
  const int lane = threadIdx.x & 63, wave = wave_id();

  // ---- stage Wt(k, c) into fragment order; global reads walk W in its storage order
  {
    float* wlf = reinterpret_cast<float*>(wl);
    constexpr int kpad = NQ * 16, cpad = NCB * 16;
    for (int idx = threadIdx.x; idx < kpad * cpad; idx += 256) {
      int k, c;
      if (a.mode == ROWSA_DOWN) { c = idx / kpad; k = idx % kpad; }   // W_down[c][k]: k fastest
      else                      { k = idx / cpad; c = idx % cpad; }   // W_up[k][c]:   c fastest
      float v = 0.f;
      if (k < a.d && c < a.r)
        v = (a.mode == ROWSA_DOWN) ? a.W[static_cast<size_t>(c) * a.d + k]
                                   : a.W[static_cast<size_t>(k) * a.r + c];
      const int q = k >> 4, kq = (k >> 2) & 3, t = k & 3, cb = c >> 4, ln = (c & 15) + 16 * kq;
      wlf[((q * NCB + cb) * 64 + ln) * 4 + t] = v;
    }
  }
  __syncthreads();

  const float sval = (a.mode == ROWSA_DP && a.scalar) ? a.scalar[0] : 1.f;
  const int node = lane & 15, kq = lane >> 4;
  const int nblk = ceil_div(a.n, 16);
  const int stride = gridDim.x * 4;
  const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
  float ds_acc = 0.f;

  // per-lane column offsets of the NQ k-steps (clamped into the row) and their validity
  int koff[NQ];
  bool kok[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    koff[q] = min(16 * q + 4 * kq, a.d - 4);
    kok[q] = 16 * q + 4 * kq < a.d;
  }

  // nullable operands: always-valid pointer + index 0, value selected afterwards (no branch)
  const bool has_rs = a.row_scale != nullptr, has_p = a.P != nullptr;
  const float* __restrict__ rsp = has_rs ? a.row_scale : a.X;
  const float* __restrict__ pp = has_p ? a.P : a.X;

  // DEPTH rings of NQ registers: ring dd holds the wave's block blk + dd * stride; once a k-step
  // has fed its MFMAs the slot is refilled for the block DEPTH strides ahead, so a load has
  // DEPTH block periods to arrive (one period is shorter than the memory latency under load).
  int blk = blockIdx.x * 4 + wave;
  f32x4 x[DEPTH][NQ];
  // Epilogue operands (row scale, P rows) travel ONE BLOCK AHEAD like the tiles: vmcnt retires in order
  // and stores count in it, so a small load issued after a block's `out` stores can only be waited for
  // by waiting for those stores' acknowledgements first — a write round trip (~2 us) per block and
  // wave, which is what these kernels spent most of their time on through round 2.
  float rs[DEPTH][4], pv[DEPTH][NCB][4];
  auto load_epi = [&](int block, float (&rsv)[4], float (&pvv)[NCB][4]) {
    const int i0 = block * 16;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int rowc = min(i0 + 4 * kq + m, a.n - 1);
      rsv[m] = rsp[has_rs ? rowc : 0];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int cc = min(16 * cb + node, a.fp - 1);
        pvv[cb][m] = pp[has_p ? static_cast<size_t>(rowc) * a.fp + cc : 0];
      }
    }
  };
#pragma unroll
  for (int dd = 0; dd < DEPTH; ++dd) {
    const int b0 = min(blk + dd * stride, nblk - 1);
    load_epi(b0, rs[dd], pv[dd]);
    const float* xb = a.X + static_cast<size_t>(min(b0 * 16 + node, a.n - 1)) * a.d;
#pragma unroll
    for (int q = 0; q < NQ; ++q) x[dd][q] = rowsA_ldg4(xb + koff[q]);
  }
  while (blk < nblk) {
#pragma unroll
    for (int dd = 0; dd < DEPTH; ++dd) {
      const int cur = blk + dd * stride;
      const bool live = cur < nblk;                // uniform; a dead block recomputes the last one, stores nothing
      const int i0 = min(cur, nblk - 1) * 16;
      // the block this ring holds next (clamped: redundant at the tail, never invalid)
      const int nb = min(cur + DEPTH * stride, nblk - 1);
      const float* __restrict__ xn = a.X + static_cast<size_t>(min(nb * 16 + node, a.n - 1)) * a.d;
      float rsn[4], pvn[NCB][4];
      load_epi(nb, rsn, pvn);                      // next block's epilogue operands: ahead of the ring refills
      __builtin_amdgcn_sched_barrier(0);  // keep those loads ahead of the ring refills
      f32x4 acc[NCB][2];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb][0] = acc[cb][1] = zero;
      // weight fragments are double-buffered in registers: the LDS reads of k-step q + 1 are
      // issued BEFORE the MFMAs of k-step q (with one buffer the read can only follow the last
      // MFMA that uses the old values, and its whole LDS latency is exposed once per k-step)
      f32x4 wq[2][NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) wq[0][cb] = wl[cb * 64 + lane];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const f32x4 xv = kok[q] ? x[dd][q] : zero;
#if GCONV_LAB_ROWSA & 1
        if (q & 1) {
          x[dd][q - 1] = rowsA_ldg4(xn + koff[q - 1]);
          x[dd][q] = rowsA_ldg4(xn + koff[q]);
        } else if (q == NQ - 1) {
          x[dd][q] = rowsA_ldg4(xn + koff[q]);
        }
#else
        x[dd][q] = rowsA_ldg4(xn + koff[q]);  // ring: refill this slot ...
#endif
        if (q + 1 < NQ) {
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) wq[(q + 1) & 1][cb] = wl[((q + 1) * NCB + cb) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);  // ... BEFORE this k-step's MFMAs, not after the burst
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const f32x4 w = wq[q & 1][cb];
          acc[cb][0] = mfma16(xv[0], w[0], acc[cb][0]);
          acc[cb][1] = mfma16(xv[1], w[1], acc[cb][1]);
          acc[cb][0] = mfma16(xv[2], w[2], acc[cb][0]);
          acc[cb][1] = mfma16(xv[3], w[3], acc[cb][1]);
        }
      }

      // D register m <-> (row i0 + 4 kq + m, column 16 cb + node)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const f32x4 res = acc[cb][0] + acc[cb][1];
        const int c = 16 * cb + node;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int row = i0 + 4 * kq + m;
          if (live && row < a.n && c < a.fp) {
            float v = res[m];
            ds_acc += v * (has_p ? pv[dd][cb][m] : 0.f);
            if (a.mode == ROWSA_DP) v *= sval;
#if GCONV_LAB_ROWSA & 8
            if (v == 123.456f)
#endif
#if GCONV_LAB_ROWSA & 4
            __builtin_nontemporal_store(v * (has_rs ? rs[dd][m] : 1.f), &a.out[static_cast<size_t>(row) * a.fp + c]);
#else
            a.out[static_cast<size_t>(row) * a.fp + c] = v * (has_rs ? rs[dd][m] : 1.f);
#endif
          }
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        rs[dd][m] = rsn[m];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) pv[dd][cb][m] = pvn[cb][m];
      }
    }
    blk += DEPTH * stride;
  }

  if (a.ds_partial) {  // block-wide sum in a fixed order
    __shared__ float red[4];
    for (int off = 32; off > 0; off >>= 1) ds_acc += __shfl_xor(ds_acc, off);
    if (lane == 0) red[wave] = ds_acc;
    __syncthreads();
    if (threadIdx.x == 0) a.ds_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// k-step counts with a compiled kernel: d <= 16 * NQ is served by the next one up
static int rowsA_nq(int d) {
  static const int kNq[] = {2, 4, 8, 12, 16, 19, 24, 32, 48, 64};
  const int need = ceil_div(d, 16);
  for (int v : kNq)
    if (need <= v) return v;
  return 0;
}

size_t rowsA_lds_bytes(int d, int fp) {
  const int nq = rowsA_nq(d), ncb = fp >= 16 ? fp / 16 : 1;
  return static_cast<size_t>(nq) * ncb * 64 * sizeof(f32x4);
}

int rowsA_grid(int n, int d, int fp) {
  const int nblk = ceil_div(n, 16);
  int g = ceil_div(nblk, 4);  // persistent: <= 2 workgroups per CU walk the node blocks
  static const int wgpc = std::getenv("GCONV_ROWSA_WGPC") ? std::atoi(std::getenv("GCONV_ROWSA_WGPC")) : 2;
  // (a weight table above 80 KB of LDS leaves room for one workgroup per CU)
  const int per_cu = rowsA_lds_bytes(d, fp) > 80 * 1024 ? 1 : wgpc;
  static const int lab_grid = std::getenv("GCONV_LAB_ROWSA_GRID") ? std::atoi(std::getenv("GCONV_LAB_ROWSA_GRID")) : 0;
  if (lab_grid > 0 && lab_grid <= 2048) return g < lab_grid ? g : lab_grid;   // (lab: any grid; ds partial buffers hold 2048)
  return g < 1 ? 1 : (g > per_cu * launch_cus() ? per_cu * launch_cus() : g);   // measured best at S-arxiv: 2 (27 us) vs 3-6 (28-29 us)
}

// Ring depth: 1. A second ring (two block periods for a load to arrive, 123 VGPRs) was measured
// in round 2 at S-arxiv: 27.2 -> 32.3 us, at 2 and at 3 workgroups per CU — the kernel is not
// short of bytes in flight. GCONV_ROWSA_DEPTH=2 selects it for d <= 192 (A/B knob).
template <int NCB>
static gconv_status_t launch_rowsA_ncb(const RowsAArgs& a, int nq, int grid, size_t lds,
                                       hipStream_t stream) {
  static const int depth_env = std::getenv("GCONV_ROWSA_DEPTH") ? std::atoi(std::getenv("GCONV_ROWSA_DEPTH")) : 1;
  switch (nq) {
#define GCONV_ROWSA_CASE(NQ_)                                                                      \
    case NQ_:                                                                                      \
      if (lds > 48 * 1024)   /* wide rows x large rank: the fragment table needs opting in */      \
        GCONV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rowsA_kernel<NCB, NQ_, 1>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds))); \
      if (depth_env >= 2 && NQ_ <= 12 && lds <= 48 * 1024)                                         \
        gemm_rowsA_kernel<NCB, NQ_, 2><<<grid, 256, lds, stream>>>(a);                             \
      else gemm_rowsA_kernel<NCB, NQ_, 1><<<grid, 256, lds, stream>>>(a);                          \
      break;
    GCONV_ROWSA_CASE(2) GCONV_ROWSA_CASE(4) GCONV_ROWSA_CASE(8) GCONV_ROWSA_CASE(12)
    GCONV_ROWSA_CASE(16) GCONV_ROWSA_CASE(19) GCONV_ROWSA_CASE(24) GCONV_ROWSA_CASE(32)
    GCONV_ROWSA_CASE(48) GCONV_ROWSA_CASE(64)
#undef GCONV_ROWSA_CASE
    default: return GCONV_E_UNSUPPORTED;
  }
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

gconv_status_t launch_rowsA(const RowsAArgs& a, hipStream_t stream) {
  if (a.n == 0) return GCONV_OK;
  const int nq = rowsA_nq(a.d);
  const size_t lds = rowsA_lds_bytes(a.d, a.fp);
  if (nq == 0 || lds > kRowsAMaxLds) return GCONV_E_UNSUPPORTED;
  const int grid = rowsA_grid(a.n, a.d, a.fp);
  switch (a.fp >= 16 ? a.fp / 16 : 1) {
    case 1: return launch_rowsA_ncb<1>(a, nq, grid, lds, stream);
    case 2: return launch_rowsA_ncb<2>(a, nq, grid, lds, stream);
    case 4: return launch_rowsA_ncb<4>(a, nq, grid, lds, stream);
  }
  return GCONV_E_UNSUPPORTED;
}

// ===========================================================================
// K3: out = s * (P W_up^T + b_up + X)
// ===========================================================================
template <int FP, int TB>
__global__ __launch_bounds__(TB) void up_fwd_kernel(UpArgs a, int nchunks, int nslots) {
  constexpr int KS = FP / 4;
  const int lane = threadIdx.x & 63, w = wave_id();
  const int J = w % nchunks, slot = w / nchunks;
  const int jj = lane & 15, t = lane >> 4;
  const int colbase = 64 * J + 4 * jj;
  const bool colvalid = colbase < a.d;
  const float sval = a.scalar ? a.scalar[0] : 1.f;

  // B fragments: wb[s][c] = W_up[colbase + c][rank_k(s, t)]
  float wb[KS][4];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = rank_k<FP>(s, t);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      wb[s][c] = (colvalid && k < a.r) ? a.W_up[static_cast<size_t>(colbase + c) * a.r + k] : 0.f;
  }
  f32x4 bu = f32x4{0.f, 0.f, 0.f, 0.f};
  if (colvalid && a.b_up) bu = ldg4(a.b_up + colbase);

  const int nblk = ceil_div(a.n, 16);
  const int stride = gridDim.x * nslots;
  int blk = blockIdx.x * nslots + slot;
  f32x4 xv[4], xn[4];
  float pa[KS], pn[KS];
#pragma unroll
  for (int m = 0; m < 4; ++m) xv[m] = xn[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (blk < nblk) {
    if (a.skip) load_tile_L(a.X, blk * 16, t, a.n, a.d, colbase, colvalid, xv);  // uniform branch
    load_rank_frag<FP>(a.P, min(blk * 16 + jj, a.n - 1), t, pa);
  }
  while (blk < nblk) {
    const int i0 = blk * 16, nxt = blk + stride;
    {  // next block's operands fly while this block computes. Unconditional with a clamped
       // block index: a conditional prefetch would force vmcnt(0) at every use.
      const int nc = min(nxt, nblk - 1);
      if (a.skip) load_tile_L(a.X, nc * 16, t, a.n, a.d, colbase, colvalid, xn);
      load_rank_frag<FP>(a.P, min(nc * 16 + jj, a.n - 1), t, pn);
    }
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = mfma16(pa[s], wb[s][c], acc[c]);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = i0 + 4 * t + m;
      if (colvalid && row < a.n) {
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = sval * (acc[c][m] + bu[c] + xv[m][c]);
        stg4_stream(a.out + static_cast<size_t>(row) * a.d + colbase, o);
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) xv[m] = xn[m];
#pragma unroll
    for (int s = 0; s < KS; ++s) pa[s] = pn[s];
    blk = nxt;
  }
}

template <int FP>
static gconv_status_t launch_up_fwd_fp(const UpArgs& a, int grid, int threads, int nchunks,
                                       int nslots, hipStream_t stream) {
  if (threads <= 256)
    up_fwd_kernel<FP, 256><<<grid, threads, 0, stream>>>(a, nchunks, nslots);
  else if (threads <= 512)
    up_fwd_kernel<FP, 512><<<grid, threads, 0, stream>>>(a, nchunks, nslots);
  else
    up_fwd_kernel<FP, 1024><<<grid, threads, 0, stream>>>(a, nchunks, nslots);
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

gconv_status_t launch_up_fwd(const UpArgs& a, hipStream_t stream) {
  if (a.n == 0) return GCONV_OK;
  const int nchunks = ceil_div(a.d, 64);
  if (nchunks > 16) return GCONV_E_UNSUPPORTED;
  const int nslots = chunk_slots(nchunks, 4);
  const int nblk = ceil_div(a.n, 16);
  // persistent: 2 workgroups per CU, each wave slot walks node blocks with a 1-deep prefetch
  int grid = ceil_div(nblk, nslots);
  if (grid > 2 * launch_cus()) grid = 2 * launch_cus();   // measured: 2 per CU 37 us, 4 per CU 41 us, 6 per CU 45 us
  const int threads = 64 * nchunks * nslots;
  switch (a.fp) {
    case 4:  return launch_up_fwd_fp<4>(a, grid, threads, nchunks, nslots, stream);
    case 8:  return launch_up_fwd_fp<8>(a, grid, threads, nchunks, nslots, stream);
    case 16: return launch_up_fwd_fp<16>(a, grid, threads, nchunks, nslots, stream);
    case 32: return launch_up_fwd_fp<32>(a, grid, threads, nchunks, nslots, stream);
    case 64: return launch_up_fwd_fp<64>(a, grid, threads, nchunks, nslots, stream);
  }
  return GCONV_E_UNSUPPORTED;
}

// ===========================================================================
// B4: dX = T W_down + s G ; partials of dW_down^T = T^T X, dW_up^T = P^T G,
//     colsum(G), sum(G . X)
// One workgroup = nchunks x nslots waves (8 waves when d <= 512). Every wave keeps its
// dW^T tiles in registers over all the node blocks it walks; at the end the slots of a
// chunk are added through LDS in slot order (deterministic) and ONE partial per
// workgroup goes to HBM.
// ===========================================================================
BwdMainGeom bwd_main_geom(int n, int d) {
  BwdMainGeom g;
  g.nchunks = ceil_div(d, 64);
  g.nslots = chunk_slots(g.nchunks, 8);
  const int nblk = ceil_div(n, 16);
  // <= 1 workgroup per CU (one 8-wave workgroup per CU measured best at S-arxiv); below that, one
  // node block per wave slot: a Cora / PubMed-sized graph is latency-bound and wants every CU it
  // can fill (PubMed: 39 workgroups x 4 blocks per wave 16.7 us -> 155 workgroups), the partial
  // write-out it multiplies is a few KB per workgroup
  static const int per_slot = std::getenv("GCONV_B4_BLOCKS_PER_SLOT") ? std::atoi(std::getenv("GCONV_B4_BLOCKS_PER_SLOT")) : 1;
  int grid = ceil_div(nblk, per_slot * g.nslots);
  if (grid > launch_cus()) grid = launch_cus();
  if (grid < 1) grid = 1;
  g.grid = grid;
  g.nparts = grid;
  return g;
}

template <int FP, bool NEED_DX, int TB>
__global__ __launch_bounds__(TB) void bwd_main_kernel(BwdMainArgs a, int nchunks, int nslots) {
  extern __shared__ float red[];  // [2 FP + 1][nchunks][64] when nslots > 1
  constexpr int KS = FP / 4;
  constexpr int NCB = FP >= 16 ? FP / 16 : 1;
  const int lane = threadIdx.x & 63, w = wave_id();
  const int J = w % nchunks, slot = w / nchunks;
  const int jj = lane & 15, t = lane >> 4;
  const int colbase = 64 * J + 4 * jj;
  const bool colvalid = colbase < a.d;
  const float sval = a.scalar ? a.scalar[0] : 1.f;

  // B fragments for dX: wd[s][c] = W_down[rank_k(s, t)][colbase + c]
  float wd[NEED_DX ? KS : 1][4];
  if (NEED_DX) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = rank_k<FP>(s, t);
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (colvalid && k < a.r) v = ldg4(a.W_down + static_cast<size_t>(k) * a.d + colbase);
#pragma unroll
      for (int c = 0; c < 4; ++c) wd[s][c] = v[c];
    }
  }

  f32x4 awd[NCB][4], awu[NCB][4];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int c = 0; c < 4; ++c) awd[cb][c] = awu[cb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
  float dsx = 0.f;

  const int nblk = ceil_div(a.n, 16);
  const int stride = gridDim.x * nslots;
  int blk = blockIdx.x * nslots + slot;
  f32x4 gv[4], xv[4], gn[4], xn[4];
  // Rank-width operands of a block (T / P rows as MFMA A operands, T fragments for dX). EVERYTHING a
  // block reads is requested one block ahead, tiles and fragments alike: vmcnt retires in order and
  // stores count in it, so a fragment load issued after the previous block's dX stores (round 2) could
  // only be waited for by waiting for those stores' acknowledgements — a write round trip per block and
  // wave: 7 us per block where the MFMAs take 1.3.
  float at[4][NCB], ap[4][NCB], ta[NEED_DX ? KS : 1];
  auto load_frags = [&](int block, float (&atv)[4][NCB], float (&apv)[4][NCB], float (&tav)[NEED_DX ? KS : 1]) {
    const int i0 = block * 16;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const size_t rb = static_cast<size_t>(min(i0 + 4 * t + m, a.n - 1)) * FP;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int rc = min(16 * cb + jj, FP - 1);  // rank column this lane feeds as the A row
        atv[m][cb] = a.T[rb + rc];
        apv[m][cb] = a.P[rb + rc];
      }
    }
    if constexpr (NEED_DX) load_rank_frag<FP>(a.T, min(i0 + jj, a.n - 1), t, tav);
  };
  if (blk < nblk) {
    load_frags(blk, at, ap, ta);
    load_tile_L(a.G, blk * 16, t, a.n, a.d, colbase, colvalid, gv);
    load_tile_L(a.X, blk * 16, t, a.n, a.d, colbase, colvalid, xv);
  }
  while (blk < nblk) {
    const int i0 = blk * 16, nxt = blk + stride;
    float atn[4][NCB], apn[4][NCB], tan[NEED_DX ? KS : 1];
    {  // the next block's fragments and G / X tiles (unconditional, clamped block index)
      const int nc = min(nxt, nblk - 1);
      load_frags(nc, atn, apn, tan);
      // (tiles: clamped addresses, NO select here — a select would have to wait for its load on the spot)
      const int colc = min(colbase, a.d - 4);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const size_t ro = static_cast<size_t>(min(nc * 16 + 4 * t + m, a.n - 1)) * a.d + colc;
        gn[m] = ldg4(a.G + ro);
        xn[m] = ldg4(a.X + ro);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // requests first, then this block's MFMAs and stores
    // rows / rank columns past the end contribute zeros (the loads above were clamped)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const bool ok = i0 + 4 * t + m < a.n && 16 * cb + jj < FP;
        at[m][cb] = ok ? at[m][cb] : 0.f;
        ap[m][cb] = ok ? ap[m][cb] : 0.f;
      }
    }

    // contraction over the 16 nodes: k-step m covers nodes i0 + 4 t + m
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          awd[cb][c] = mfma16(at[m][cb], xv[m][c], awd[cb][c]);
          awu[cb][c] = mfma16(ap[m][cb], gv[m][c], awu[cb][c]);
        }
      }
      cs += gv[m];
      if (a.skip) dsx = dot4_acc(gv[m], xv[m], dsx);
    }

    if (NEED_DX) {
      f32x4 acc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = mfma16(ta[s], wd[s][c], acc[c]);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int row = i0 + 4 * t + m;
        if (colvalid && row < a.n) {
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = acc[c][m] + (a.skip ? sval * gv[m][c] : 0.f);
          stg4_stream(a.dX + static_cast<size_t>(row) * a.d + colbase, o);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const bool ok = colvalid && nxt * 16 + 4 * t + m < a.n;   // (nxt >= nblk: the loop ends, values unused)
      gv[m] = ok ? gn[m] : f32x4{0.f, 0.f, 0.f, 0.f};
      xv[m] = ok ? xn[m] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) { at[m][cb] = atn[m][cb]; ap[m][cb] = apn[m][cb]; }
    }
    if constexpr (NEED_DX) {
#pragma unroll
      for (int s = 0; s < KS; ++s) ta[s] = tan[s];
    }
    blk = nxt;
  }

  bwd_main_tail<FP>(a, awd, awu, cs, dsx, red, nchunks, nslots, J, slot, jj, t, lane, w, colbase, colvalid);
}

template <int FP>
static gconv_status_t launch_bwd_main_fp(const BwdMainArgs& a, const BwdMainGeom& g,
                                         hipStream_t stream) {
  const int threads = 64 * g.nchunks * g.nslots;
  const size_t lds = bwd_main_lds_bytes(g.nchunks, g.nslots, FP);
  if (lds > 160 * 1024) return GCONV_E_UNSUPPORTED;
#define GCONV_BWD_LAUNCH(DX, TB_)                                                           \
  do {                                                                                      \
    if (lds > 48 * 1024)                                                                    \
      GCONV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_main_kernel<FP, DX, TB_>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize,             \
                                    static_cast<int>(lds)));                                \
    bwd_main_kernel<FP, DX, TB_><<<g.grid, threads, lds, stream>>>(a, g.nchunks, g.nslots); \
  } while (0)
  if (a.dX) {
    if (threads <= 256) GCONV_BWD_LAUNCH(true, 256);
    else if (threads <= 512) GCONV_BWD_LAUNCH(true, 512);
    else GCONV_BWD_LAUNCH(true, 1024);
  } else {
    if (threads <= 256) GCONV_BWD_LAUNCH(false, 256);
    else if (threads <= 512) GCONV_BWD_LAUNCH(false, 512);
    else GCONV_BWD_LAUNCH(false, 1024);
  }
#undef GCONV_BWD_LAUNCH
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

gconv_status_t launch_bwd_main(const BwdMainArgs& a, hipStream_t stream) {
  const BwdMainGeom g = bwd_main_geom(a.n, a.d);
  if (g.nchunks > 16) return GCONV_E_UNSUPPORTED;
  switch (a.fp) {
    case 4:  return launch_bwd_main_fp<4>(a, g, stream);
    case 8:  return launch_bwd_main_fp<8>(a, g, stream);
    case 16: return launch_bwd_main_fp<16>(a, g, stream);
    case 32: return launch_bwd_main_fp<32>(a, g, stream);
    case 64: return launch_bwd_main_fp<64>(a, g, stream);
  }
  return GCONV_E_UNSUPPORTED;
}

// ===========================================================================
// B5: fixed-order reduction of the per-workgroup partials
// ===========================================================================
// element space: [0, r d) dW_down | [r d, 2 r d) dW_up | [2 r d, 2 r d + d) colsum(G)
//                | [2 r d + d, 2 r d + d + r) db_down | (+ 1: d(scalar) when complete)
// block = kFinE consecutive elements x kFinS interleaved partial streams; every stream keeps
// 4 loads in flight; streams are added in a fixed tree.
constexpr int kFinE = 32, kFinS = 8;   // (32 consecutive floats: whole 128-byte lines per stream; round 1-4a: 16 x 16,
                                        //  two workgroups — often on two XCDs — fetched every line of the partials)

__device__ __forceinline__ float strided_sum(const float* __restrict__ q, size_t stride, int first,
                                             int count) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int p = first;
  // 16 partials requested before the first is added (256 persistent workgroups = 32 per stream: two round trips
  // instead of eight); the additions keep the order of the 4-wide loop below
  for (; p + 15 * kFinS < count; p += 16 * kFinS) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = q[(p + k * kFinS) * stride];
#pragma unroll
    for (int k = 0; k < 16; k += 4) {
      s0 += v[k]; s1 += v[k + 1]; s2 += v[k + 2]; s3 += v[k + 3];
    }
  }
  for (; p + 3 * kFinS < count; p += 4 * kFinS) {
    s0 += q[p * stride]; s1 += q[(p + kFinS) * stride];
    s2 += q[(p + 2 * kFinS) * stride]; s3 += q[(p + 3 * kFinS) * stride];
  }
  for (; p < count; p += kFinS) s0 += q[p * stride];
  return (s0 + s1) + (s2 + s3);
}

__device__ __forceinline__ void finalize_body(const FinalizeArgs& a, int block, float (*red)[kFinE]) {
  const int le = threadIdx.x % kFinE, ps = threadIdx.x / kFinE;
  const int e = block * kFinE + le;
  const int rd = a.r * a.d, total = 2 * rd + a.d + a.r + (a.ds_complete ? 1 : 0);
  float sum = 0.f;
  if (e < total) {
    if (e < 2 * rd) {
      const float* part = e < rd ? a.part_wd : a.part_wu;
      const int idx = e < rd ? e : e - rd;
      const int c = idx / a.d, j = idx % a.d;
      sum = strided_sum(part + static_cast<size_t>(c) * a.d + j, static_cast<size_t>(a.fp) * a.d, ps,
                        a.nparts);
    } else if (e < 2 * rd + a.d) {
      sum = strided_sum(a.part_cs + (e - 2 * rd), a.d, ps, a.nparts);
    } else if (e < 2 * rd + a.d + a.r) {
      sum = strided_sum(a.part_bd + (e - 2 * rd - a.d), a.fp, ps, a.n_bd);
    } else {
      sum = strided_sum(a.part_ds1, 1, ps, a.n_ds1);
    }
  }
  red[ps][le] = sum;
  __syncthreads();
  if (ps == 0 && e < total) {
    float v[kFinS];
#pragma unroll
    for (int k = 0; k < kFinS; ++k) v[k] = red[k][le];
#pragma unroll
    for (int w = 1; w < kFinS; w <<= 1)
#pragma unroll
      for (int k = 0; k < kFinS; k += 2 * w) v[k] += v[k + w];
    const float r0 = v[0];
    const float sval = a.scalar ? a.scalar[0] : 1.f;
    if (e < rd) {
      a.g_wd[e] = r0;
    } else if (e < 2 * rd) {
      const int idx = e - rd, c = idx / a.d, j = idx % a.d;
      a.g_wu[static_cast<size_t>(j) * a.r + c] = sval * r0;
    } else if (e < 2 * rd + a.d) {
      const int j = e - 2 * rd;
      a.cs_tmp[j] = r0;
      if (a.g_bu) a.g_bu[j] = sval * r0;
    } else if (e < 2 * rd + a.d + a.r) {
      if (a.g_bd) a.g_bd[e - 2 * rd - a.d] = r0;
    } else {
      if (a.g_s) a.g_s[0] = r0;
    }
  }
}

// d(scalar) = sum_i (G_i W_up) . P_i  +  b_up . colsum(G)  +  [skip] sum(G . X)
// ONE extra block of the finalize launch (general path with a learnable scalar; the fused tile
// path hands over complete per-workgroup partials instead): part_ds1 comes from B1, part_dsx from
// B4's waves, which fold their share of b_up . colsum(G) in (bwd_main_tail) — round 1 reduced
// colsum(G) first and ran this as a second launch behind the first.
__device__ __forceinline__ void finalize_scalar_body(const FinalizeArgs& a, float* red) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < a.n_ds1; i += 256) acc += a.part_ds1[i];
  for (int i = threadIdx.x; i < a.n_dsx; i += 256) acc += a.part_dsx[i];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) a.g_s[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

// blocks of one reduction: ceil(total / kFinE) element blocks (+ 1 scalar block)
static int finalize_main_blocks(const FinalizeArgs& f) {
  return ceil_div(2 * f.r * f.d + f.d + f.r + (f.ds_complete ? 1 : 0), kFinE);
}
static bool finalize_wants_scalar_block(const FinalizeArgs& f) { return f.g_s && !f.ds_complete; }

__global__ __launch_bounds__(256) void finalize_kernel(FinalizeArgs a, int main_blocks) {
  __shared__ float red[kFinS][kFinE];
  if (static_cast<int>(blockIdx.x) < main_blocks) finalize_body(a, blockIdx.x, red);   // uniform
  else finalize_scalar_body(a, &red[0][0]);
}

// Several pending reductions in one launch: blocks [start[j], start[j + 1]) belong to entry j,
// the last of them being its scalar block when it has one.
struct FinalizeBatch {
  FinalizeArgs a[GCONV_FINALIZE_BATCH];
  int start[GCONV_FINALIZE_BATCH + 1];
  int count;
};

__global__ __launch_bounds__(256) void finalize_batched_kernel(FinalizeBatch b) {
  __shared__ float red[kFinS][kFinE];
  int j = 0;
  while (j + 1 < b.count && static_cast<int>(blockIdx.x) >= b.start[j + 1]) ++j;   // uniform
  const FinalizeArgs& a = b.a[j];
  const int local = blockIdx.x - b.start[j];
  const int main_blocks = ceil_div(2 * a.r * a.d + a.d + a.r + (a.ds_complete ? 1 : 0), kFinE);
  if (local < main_blocks) finalize_body(a, local, red);
  else finalize_scalar_body(a, &red[0][0]);
}

// Stage 1 for tall partial arrays: [rows, width] -> [kReduceBlocks, width], fixed order.
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ in, int rows,
                                                          int width, float* __restrict__ out) {
  __shared__ float red[256];
  const int c = threadIdx.x % width, lr = threadIdx.x / width, nlr = 256 / width;
  const int per = ceil_div(rows, static_cast<int>(gridDim.x));
  const int r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int r = r0 + lr;
  for (; r + 3 * nlr < r1; r += 4 * nlr) {
    a0 += in[static_cast<size_t>(r) * width + c];
    a1 += in[static_cast<size_t>(r + nlr) * width + c];
    a2 += in[static_cast<size_t>(r + 2 * nlr) * width + c];
    a3 += in[static_cast<size_t>(r + 3 * nlr) * width + c];
  }
  for (; r < r1; r += nlr) a0 += in[static_cast<size_t>(r) * width + c];
  red[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (lr == 0) {
    float v = 0.f;
    for (int k = 0; k < nlr; ++k) v += red[k * width + c];
    out[static_cast<size_t>(blockIdx.x) * width + c] = v;
  }
}

gconv_status_t launch_reduce_rows(const float* in, int rows, int width, float* out,
                                  hipStream_t stream) {
  reduce_rows_kernel<<<kReduceBlocks, 256, 0, stream>>>(in, rows, width, out);
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

gconv_status_t launch_finalize_batched(const FinalizeArgs* a, int count, hipStream_t stream) {
  for (int base = 0; base < count; base += GCONV_FINALIZE_BATCH) {
    FinalizeBatch b;
    b.count = count - base < GCONV_FINALIZE_BATCH ? count - base : GCONV_FINALIZE_BATCH;
    b.start[0] = 0;
    for (int j = 0; j < b.count; ++j) {
      b.a[j] = a[base + j];
      const FinalizeArgs& f = b.a[j];
      b.start[j + 1] = b.start[j] + finalize_main_blocks(f) + (finalize_wants_scalar_block(f) ? 1 : 0);
    }
    for (int j = b.count; j < GCONV_FINALIZE_BATCH; ++j) b.start[j + 1] = b.start[b.count];
    finalize_batched_kernel<<<b.start[b.count], 256, 0, stream>>>(b);
    GCONV_LAUNCH_CHECK();
  }
  return GCONV_OK;
}

gconv_status_t launch_finalize(const FinalizeArgs& a, hipStream_t stream) {
  const int main_blocks = finalize_main_blocks(a);
  finalize_kernel<<<main_blocks + (finalize_wants_scalar_block(a) ? 1 : 0), 256, 0, stream>>>(a, main_blocks);
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

}  // namespace gconv

using namespace gconv;

// ---------------------------------------------------------------------------
// C ABI: stand-alone projections
// ---------------------------------------------------------------------------
static gconv_status_t check_dims(int32_t n, int32_t d, int32_t r) {
  if (n < 0 || d <= 0 || r <= 0) return GCONV_E_INVALID_ARG;
  if (d % 4 != 0 || d > 1024 || padded_rank(r) == 0) return GCONV_E_UNSUPPORTED;
  return GCONV_OK;
}

extern "C" gconv_status_t gconv_gemm_down_f32(const float* x, const float* w_down,
                                              const float* row_scale, float* h0,
                                              int32_t num_nodes, int32_t hidden, int32_t rank,
                                              gconv_stream_t stream) {
  gconv_status_t st = check_dims(num_nodes, hidden, rank);
  if (st != GCONV_OK) return st;
  if (num_nodes == 0) return GCONV_OK;
  if (!x || !w_down || !h0) return GCONV_E_INVALID_ARG;
  if (!aligned16(x) || !aligned16(h0)) return GCONV_E_ALIGNMENT;
  RowsAArgs a{};
  a.X = x; a.W = w_down; a.row_scale = row_scale; a.out = h0;
  a.n = num_nodes; a.d = hidden; a.r = rank; a.fp = padded_rank(rank);
  a.mode = ROWSA_DOWN;
  return launch_rowsA(a, static_cast<hipStream_t>(stream));
}

extern "C" gconv_status_t gconv_gemm_up_fused_f32(const float* p, const float* w_up,
                                                  const float* b_up, const float* scalar,
                                                  const float* x, int32_t skip, float* out,
                                                  int32_t num_nodes, int32_t hidden,
                                                  int32_t rank, gconv_stream_t stream) {
  gconv_status_t st = check_dims(num_nodes, hidden, rank);
  if (st != GCONV_OK) return st;
  if (num_nodes == 0) return GCONV_OK;
  if (!p || !w_up || !out || (skip && !x)) return GCONV_E_INVALID_ARG;
  if (!aligned16(p) || !aligned16(out) || (x && !aligned16(x)) || (b_up && !aligned16(b_up)))
    return GCONV_E_ALIGNMENT;
  UpArgs a{};
  a.P = p; a.W_up = w_up; a.b_up = b_up; a.scalar = scalar; a.X = x; a.out = out;
  a.n = num_nodes; a.d = hidden; a.r = rank; a.fp = padded_rank(rank); a.skip = skip;
  return launch_up_fwd(a, static_cast<hipStream_t>(stream));
}
