// Fused tile-local adapter forward / backward for block-diagonal graphs (molecule batches).
//
// A collated molecule batch is a block-diagonal graph: graph prep cuts the node range
// into component tiles that no edge crosses (gconv_graph_t::tile_ptr). ONE workgroup then
// runs the whole adapter for its tile — the three rank-width intermediates (H0', H2', P,
// resp. dP', dH1', T) never leave LDS, the grid-wide dependencies of the four-launch path
// become __syncthreads(), and X is read from HBM once (its second read, for the
// residual, comes out of L2 a few microseconds later):
//
//   forward   phase 1  H0' = dis . (X_t Wd^T)                   MFMA, A layout, W_d frags in LDS
//             phase 2  H2' = dis . act(dis . (A0 H0' + H0') + bd)      LDS gathers
//             phase 3  P   = dis . (A0 H2' + H2')                      LDS gathers
//             phase 4  out = s (P Wu^T + bu + X_t)              MFMA, L layout
//   backward  phase 1  dP' = dis . s (G_t Wu) ; sum (G Wu) . P  MFMA, A layout
//             phase 2  dH1' = dis . (S^T dP) . act' ; dbd
//             phase 3  T = S^T dH1
//             phase 4  dX_t = T Wd + s G_t ; dWd^T, dWu^T, colsum(G), sum G.X   MFMA, L layout
//
// Same math, same epilogues and the same deterministic B5 finalize as the four-launch
// path (adapter.hip); replaces /root/reference/src/finetune/gconv_adapter.py:80-108 and
// its autograd backward for the batches the inductive script feeds it
// (scripts/finetune_inductive_learning.py:43-67).
#include <cstdlib>

#include "fused_tile.h"

#ifndef GCONV_BWD_WAVES
#define GCONV_BWD_WAVES 2
#endif

namespace gconv {

namespace {

__device__ __forceinline__ f32x4 zero4f() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

template <int FP>
__device__ __forceinline__ int rank_k_(int s, int t) {
  return FP >= 16 ? 16 * (s >> 2) + 4 * t + (s & 3) : 4 * s + t;
}

// A fragments of an LDS rank-width tile Z[cap, FP] for local row `row` (clamped by caller)
template <int FP>
__device__ __forceinline__ void lds_rank_frag(const float* Z, int row, int t, bool ok,
                                              float (&frag)[FP / 4]) {
  const float* p = Z + row * FP;
  if constexpr (FP >= 16) {
#pragma unroll
    for (int q = 0; q < FP / 16; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + 16 * q + 4 * t);
      frag[4 * q + 0] = ok ? v[0] : 0.f; frag[4 * q + 1] = ok ? v[1] : 0.f;
      frag[4 * q + 2] = ok ? v[2] : 0.f; frag[4 * q + 3] = ok ? v[3] : 0.f;
    }
  } else {
#pragma unroll
    for (int s = 0; s < FP / 4; ++s) frag[s] = ok ? p[4 * s + t] : 0.f;
  }
}

// Stage Wt(k, c) (k < d, c < r) into MFMA fragment order [NQ][NCB][64] float4.
template <int NCB, int NQ>
__device__ __forceinline__ void stage_weights(f32x4* wl, const float* __restrict__ W, int d, int r,
                                              bool down) {
  float* wlf = reinterpret_cast<float*>(wl);
  constexpr int kpad = NQ * 16, cpad = NCB * 16;
  for (int idx = threadIdx.x; idx < kpad * cpad; idx += blockDim.x) {
    int k, c;
    if (down) { c = idx / kpad; k = idx % kpad; } else { k = idx / cpad; c = idx % cpad; }
    float v = 0.f;
    if (k < d && c < r) v = down ? W[static_cast<size_t>(c) * d + k] : W[static_cast<size_t>(k) * r + c];
    const int q = k >> 4, kq = (k >> 2) & 3, t = k & 3, cb = c >> 4, ln = (c & 15) + 16 * kq;
    wlf[((q * NCB + cb) * 64 + ln) * 4 + t] = v;
  }
}

// Phase 1 of both directions: dst[local row][c] = row_scale * sval * (M_t Wt), tile rows
// [ts, ts + nt). Returns this lane's partial of sum(acc * Pmat) (backward scalar gradient).
template <int FP, int NQ>
__device__ __forceinline__ float tile_rowsA(const float* __restrict__ M, const f32x4* wl,
                                            float* dst, const float* __restrict__ dis,
                                            const float* __restrict__ Pmat, float sval, int ts,
                                            int nt, int d, int nwaves, int wave, int lane) {
  constexpr int NCB = FP >= 16 ? FP / 16 : 1;
  const int node = lane & 15, kq = lane >> 4;
  const int nb = ceil_div(nt, 16);
  float ds_acc = 0.f;
  int koff[NQ];
  bool kok[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    koff[q] = min(16 * q + 4 * kq, d - 4);
    kok[q] = 16 * q + 4 * kq < d;
  }
  for (int b = wave; b < nb; b += nwaves) {
    const int lrow = min(16 * b + node, nt - 1);
    const float* __restrict__ xb = M + static_cast<size_t>(ts + lrow) * d;
    f32x4 x[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) x[q] = ldg4(xb + koff[q]);
    f32x4 acc[NCB][2];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb][0] = acc[cb][1] = zero4f();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const f32x4 xv = kok[q] ? x[q] : zero4f();
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const f32x4 w = wl[(q * NCB + cb) * 64 + lane];
        acc[cb][0] = mfma16(xv[0], w[0], acc[cb][0]);
        acc[cb][1] = mfma16(xv[1], w[1], acc[cb][1]);
        acc[cb][0] = mfma16(xv[2], w[2], acc[cb][0]);
        acc[cb][1] = mfma16(xv[3], w[3], acc[cb][1]);
      }
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const f32x4 res = acc[cb][0] + acc[cb][1];
      const int c = 16 * cb + node;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int lr = 16 * b + 4 * kq + m;
        if (lr < nt && c < FP) {
          float v = res[m];
          if (Pmat) ds_acc += v * Pmat[static_cast<size_t>(ts + lr) * FP + c];
          v *= sval;
          if (dis) v *= dis[ts + lr];
          dst[lr * FP + c] = v;
        }
      }
    }
  }
  return ds_acc;
}

// Phases 2 / 3: one lane group (FP/4 lanes) per tile row, neighbour rows gathered from LDS.
//   sum = sum over stored edges of src[col - ts]; v = NORM ? dis * (sum + src[row]) : sum
// `rp` / `cl`: the tile's CSR slice staged in LDS (local row offsets, local column ids)
template <int FP, bool NORM>
__device__ __forceinline__ f32x4 tile_row_sum(const float* src, const int32_t* rp,
                                              const int32_t* cl, const float* __restrict__ dis,
                                              int ts, int lrow, int sub) {
  const int e0 = rp[lrow], e1 = rp[lrow + 1];
  f32x4 acc = zero4f();
  for (int e = e0; e < e1; ++e)
    acc += *reinterpret_cast<const f32x4*>(src + cl[e] * FP + 4 * sub);
  if (NORM) {
    acc += *reinterpret_cast<const f32x4*>(src + lrow * FP + 4 * sub);
    acc *= dis[ts + lrow];
  }
  return acc;
}

// Stages the CSR slice of tile rows [ts, ts + nt) into LDS (issued at tile start, so its
// latency hides under phase 1's X loads; the barrier after phase 1 publishes it).
__device__ __forceinline__ void stage_tile_csr(const int32_t* __restrict__ rowptr,
                                               const int32_t* __restrict__ col, int ts, int nt,
                                               int32_t* rp, int32_t* cl) {
  const int e0 = rowptr[ts], ne = rowptr[ts + nt] - e0;
  for (int i = threadIdx.x; i <= nt; i += blockDim.x) rp[i] = rowptr[ts + i] - e0;
  for (int e = threadIdx.x; e < ne; e += blockDim.x) cl[e] = col[e0 + e] - ts;
}

}  // namespace

// ===========================================================================
// forward — persistent workgroups walk the tiles (weights staged / loaded once)
// ===========================================================================
template <int FP, int NQ, bool NORM>
__global__ __launch_bounds__(512, 4) void tile_fwd_kernel(TileFwdArgs a) {
  extern __shared__ f32x4 lds4[];
  constexpr int NCB = FP >= 16 ? FP / 16 : 1;
  constexpr int KS = FP / 4;
  constexpr int LPR = FP / 4;
  f32x4* wl = lds4;                                              // [NQ][NCB][64]
  float* bufA = reinterpret_cast<float*>(lds4 + NQ * NCB * 64);  // [cap][FP]  H0', later P
  float* bufB = bufA + a.cap * FP;                               // [cap][FP]  H2'
  float* bufC = bufB + a.cap * FP;                               // (backward only: P tile)
  int32_t* rp = reinterpret_cast<int32_t*>(bufC + a.cap * FP + 8 * (FP + 2));  // [cap + 1]
  int32_t* cl = rp + a.cap + 16;                                 // [ecap]

  const int lane = threadIdx.x & 63, wave = wave_id();
  const int nwaves = blockDim.x >> 6;
  const int jj = lane & 15, t = lane >> 4;
  const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR, ngrp = blockDim.x / LPR;
  const float sval = a.scalar ? a.scalar[0] : 1.f;

  stage_weights<NCB, NQ>(wl, a.w_down, a.d, a.r, true);

  // phase-4 ownership: a wave owns ONE 64-column chunk (W_up fragments in registers) and
  // every nslots-th node block of a tile (nchunks <= nwaves is guaranteed by the launcher)
  const int nchunks = ceil_div(a.d, 64);
  const int nslots = nwaves / nchunks;
  const int J = wave % nchunks, slot = wave / nchunks;
  const bool active = slot < nslots;
  const int colbase = 64 * J + 4 * jj;
  const bool colvalid = colbase < a.d;
  const int colc = min(colbase, a.d - 4);
  float wb[KS][4];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = rank_k_<FP>(s, t);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float w = a.w_up[static_cast<size_t>(colc + c) * a.r + min(k, a.r - 1)];
      wb[s][c] = (colvalid && k < a.r) ? w : 0.f;
    }
  }
  f32x4 bu = zero4f();
  if (a.b_up) { const f32x4 bv = ldg4(a.b_up + colc); bu = colvalid ? bv : zero4f(); }
  __syncthreads();

  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int ts = a.tile_ptr[tile], nt = a.tile_ptr[tile + 1] - ts;
    if (nt <= 0) continue;  // uniform

    // ---- phase 1: H0' -> bufA (the tile's CSR slice lands in LDS meanwhile)
    stage_tile_csr(a.rowptr, a.col, ts, nt, rp, cl);
    tile_rowsA<FP, NQ>(a.x, wl, bufA, a.dis, nullptr, 1.f, ts, nt, a.d, nwaves, wave, lane);
    __syncthreads();

    // ---- phase 2: H2' -> bufB (+ saved_h, saved_h1)
    for (int lr = grp; lr < nt; lr += ngrp) {
      f32x4 v = tile_row_sum<FP, NORM>(bufA, rp, cl, a.dis, ts, lr, sub);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = 4 * sub + k;
        if (a.b_down && c < a.r) v[k] += a.b_down[c];
      }
      const size_t off = static_cast<size_t>(ts + lr) * FP + 4 * sub;
      if (a.act == GCONV_ACT_SILU) stg4(a.saved_h1 + off, v);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = act_fwd(a.act, v[k]);
      if (NORM) v *= a.dis[ts + lr];
      *reinterpret_cast<f32x4*>(bufB + lr * FP + 4 * sub) = v;
      stg4(a.saved_h + off, v);
    }
    __syncthreads();

    // ---- phase 3: P -> bufA (+ saved_p)
    for (int lr = grp; lr < nt; lr += ngrp) {
      const f32x4 v = tile_row_sum<FP, NORM>(bufB, rp, cl, a.dis, ts, lr, sub);
      *reinterpret_cast<f32x4*>(bufA + lr * FP + 4 * sub) = v;
      stg4(a.saved_p + static_cast<size_t>(ts + lr) * FP + 4 * sub, v);
    }
    __syncthreads();

    // ---- phase 4: out = s (P Wu^T + bu + X); the X tile of the wave's next node block is
    // in flight while the current one computes (unconditional, clamped prefetch)
    if (active) {
      const int nb = ceil_div(nt, 16);
      auto load_x = [&](int b, f32x4 (&xq)[4]) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int lr = min(16 * b + 4 * t + m, nt - 1);
          xq[m] = ldg4(a.x + static_cast<size_t>(ts + lr) * a.d + colc);
        }
      };
      f32x4 xv[4], xn[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) xv[m] = xn[m] = zero4f();
      int b = slot;
      if (b < nb && a.skip) load_x(b, xv);
      while (b < nb) {
        const int nxt = b + nslots;
        float pa[KS];
        lds_rank_frag<FP>(bufA, min(16 * b + jj, nt - 1), t, 16 * b + jj < nt, pa);
        if (a.skip) load_x(min(nxt, nb - 1), xn);
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = zero4f();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = mfma16(pa[s], wb[s][c], acc[c]);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int lr = 16 * b + 4 * t + m;
          if (colvalid && lr < nt) {
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = sval * (acc[c][m] + bu[c] + xv[m][c]);
            stg4(a.out + static_cast<size_t>(ts + lr) * a.d + colbase, o);
          }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) xv[m] = xn[m];
        b = nxt;
      }
    }
    __syncthreads();  // bufA / bufB are reused by the next tile
  }
}

// ===========================================================================
// backward — persistent workgroups; the dW^T tiles, colsum(G), bias gradient and the
// scalar-gradient pieces stay in registers across all tiles of a workgroup, so ONE
// partial row per workgroup reaches HBM and B5 is a single launch.
// ===========================================================================
template <int FP, int NQ, bool NORM, bool NEED_DX>
__global__ __launch_bounds__(512, GCONV_BWD_WAVES) void tile_bwd_kernel(TileBwdArgs a) {
  extern __shared__ f32x4 lds4[];
  constexpr int NCB = FP >= 16 ? FP / 16 : 1;
  constexpr int KS = FP / 4;
  constexpr int LPR = FP / 4;
  f32x4* wl = lds4;                                              // [NQ][NCB][64]  W_up^T frags
  float* bufA = reinterpret_cast<float*>(lds4 + NQ * NCB * 64);  // [cap][FP]  dP', later T
  float* bufB = bufA + a.cap * FP;                               // [cap][FP]  dH1'
  float* bufC = bufB + a.cap * FP;                               // [cap][FP]  P tile
  float* red = bufC + a.cap * FP;                                // [nwaves][FP + 2] scratch
  int32_t* rp = reinterpret_cast<int32_t*>(red + 8 * (FP + 2));  // [cap + 1]
  int32_t* cl = rp + a.cap + 16;                                 // [ecap]

  const int lane = threadIdx.x & 63, wave = wave_id();
  const int nwaves = blockDim.x >> 6;
  const int pid = blockIdx.x;
  const float sval = a.scalar ? a.scalar[0] : 1.f;
  const int nchunks = ceil_div(a.d, 64);
  const int jj = lane & 15, t = lane >> 4;
  const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR, ngrp = blockDim.x / LPR;

  stage_weights<NCB, NQ>(wl, a.w_up, a.d, a.r, false);

  // phase-4 ownership: wave J owns column chunk J (nchunks <= nwaves)
  const bool owner = wave < nchunks;
  const int colbase = 64 * (owner ? wave : 0) + 4 * jj;
  const bool colvalid = owner && colbase < a.d;
  const int colc = min(colbase, a.d - 4);
  float wd[NEED_DX ? KS : 1][4];
  if constexpr (NEED_DX) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = rank_k_<FP>(s, t);
      const f32x4 v = ldg4(a.w_down + static_cast<size_t>(min(k, a.r - 1)) * a.d + colc);
#pragma unroll
      for (int c = 0; c < 4; ++c) wd[s][c] = (colvalid && k < a.r) ? v[c] : 0.f;
    }
  }
  f32x4 awd[NCB][4], awu[NCB][4];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int c = 0; c < 4; ++c) awd[cb][c] = awu[cb][c] = zero4f();
  f32x4 cs = zero4f(), bgrad = zero4f();
  float ds1 = 0.f, dsx = 0.f;
  __syncthreads();

  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int ts = a.tile_ptr[tile], nt = a.tile_ptr[tile + 1] - ts;
    if (nt <= 0) continue;  // uniform

    // ---- phase 1: dP' = dis . s (G Wu) -> bufA ; ds1 += sum (G Wu) . P
    // (the tile's CSR slice and its P rows land in LDS meanwhile)
    stage_tile_csr(a.rowptr, a.col, ts, nt, rp, cl);
    for (int i = threadIdx.x; i < nt * LPR; i += blockDim.x)
      *reinterpret_cast<f32x4*>(bufC + 4 * i) = ldg4(a.saved_p + static_cast<size_t>(ts) * FP + 4 * i);
    ds1 += tile_rowsA<FP, NQ>(a.g, wl, bufA, a.dis, a.want_ds ? a.saved_p : nullptr, sval, ts, nt,
                              a.d, nwaves, wave, lane);
    __syncthreads();

    // ---- phase 2: dH1' = dis . (S^T dP . act') -> bufB ; bias gradient
    for (int lr = grp; lr < nt; lr += ngrp) {
      f32x4 v = tile_row_sum<FP, NORM>(bufA, rp, cl, a.dis, ts, lr, sub);
      const size_t off = static_cast<size_t>(ts + lr) * FP + 4 * sub;
      if (a.act == GCONV_ACT_RELU) {
        const f32x4 h2 = ldg4(a.saved_h + off);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = h2[k] > 0.f ? v[k] : 0.f;
      } else if (a.act == GCONV_ACT_SILU) {
        const f32x4 h1 = ldg4(a.saved_h1 + off);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= silu_grad(h1[k]);
      }
      bgrad += v;
      if (NORM) v *= a.dis[ts + lr];
      *reinterpret_cast<f32x4*>(bufB + lr * FP + 4 * sub) = v;
    }
    __syncthreads();

    // ---- phase 3: T = S^T dH1 -> bufA
    for (int lr = grp; lr < nt; lr += ngrp) {
      const f32x4 v = tile_row_sum<FP, NORM>(bufB, rp, cl, a.dis, ts, lr, sub);
      *reinterpret_cast<f32x4*>(bufA + lr * FP + 4 * sub) = v;
    }
    __syncthreads();

    // ---- phase 4 (owner waves): dX = T Wd + s G ; dWd^T += T^T X ; dWu^T += P^T G ;
    //      colsum(G) ; sum G.X — next block's G / X tiles prefetched
    if (owner) {
      const int nb = ceil_div(nt, 16);
      auto load_gx = [&](int b, f32x4 (&gq)[4], f32x4 (&xq)[4]) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int lrc = min(16 * b + 4 * t + m, nt - 1);
          gq[m] = ldg4(a.g + static_cast<size_t>(ts + lrc) * a.d + colc);
          xq[m] = ldg4(a.x + static_cast<size_t>(ts + lrc) * a.d + colc);
        }
      };
      f32x4 gr[4], xr[4], gn[4], xn[4];
      load_gx(0, gr, xr);
      for (int b = 0; b < nb; ++b) {
        // rank-width operands of this block first (vmcnt retires in order), then the
        // unconditional prefetch of the next block's tiles
        float at[4][NCB], ap[4][NCB];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int lr = 16 * b + 4 * t + m, lrc = min(lr, nt - 1);
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) {
            const int rc = 16 * cb + jj, rcc = min(rc, FP - 1);
            const float vt = bufA[lrc * FP + rcc];
            const float vp = bufC[lrc * FP + rcc];
            const bool ok = lr < nt && rc < FP;
            at[m][cb] = ok ? vt : 0.f;
            ap[m][cb] = ok ? vp : 0.f;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        load_gx(min(b + 1, nb - 1), gn, xn);
        f32x4 gv[4], xv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const bool ok = colvalid && 16 * b + 4 * t + m < nt;
          gv[m] = ok ? gr[m] : zero4f();
          xv[m] = ok ? xr[m] : zero4f();
        }
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
          if (a.skip) dsx += gv[m][0] * xv[m][0] + gv[m][1] * xv[m][1] + gv[m][2] * xv[m][2] + gv[m][3] * xv[m][3];
        }
        if constexpr (NEED_DX) {
          float ta[KS];
          lds_rank_frag<FP>(bufA, min(16 * b + jj, nt - 1), t, 16 * b + jj < nt, ta);
          f32x4 acc[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = zero4f();
#pragma unroll
          for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = mfma16(ta[s], wd[s][c], acc[c]);
          }
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int lr = 16 * b + 4 * t + m;
            if (colvalid && lr < nt) {
              f32x4 o;
#pragma unroll
              for (int c = 0; c < 4; ++c) o[c] = acc[c][m] + (a.skip ? sval * gv[m][c] : 0.f);
              stg4(a.dx + static_cast<size_t>(ts + lr) * a.d + colbase, o);
            }
          }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) { gr[m] = gn[m]; xr[m] = xn[m]; }
      }
    }
    __syncthreads();  // bufA / bufB are reused by the next tile
  }

  // ---- one partial row per workgroup
  // colsum(G): add the four lane quarters (fixed order); quarter 0 stores
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    cs[c] += __shfl_xor(cs[c], 16);
    cs[c] += __shfl_xor(cs[c], 32);
  }
  // b_up . colsum(G) of this wave's chunk joins the scalar-gradient partial
  float bdot = 0.f;
  if (colvalid && t == 0 && a.b_up) {
    const f32x4 bv = ldg4(a.b_up + colbase);
    bdot = bv[0] * cs[0] + bv[1] * cs[1] + bv[2] * cs[2] + bv[3] * cs[3];
  }
  if (colvalid) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int rk = 16 * cb + 4 * t + m;
        if (rk < FP) {
          const size_t o = (static_cast<size_t>(pid) * FP + rk) * a.d + colbase;
          stg4(a.part_wd + o, f32x4{awd[cb][0][m], awd[cb][1][m], awd[cb][2][m], awd[cb][3][m]});
          stg4(a.part_wu + o, f32x4{awu[cb][0][m], awu[cb][1][m], awu[cb][2][m], awu[cb][3][m]});
        }
      }
    }
    if (t == 0) stg4(a.part_cs + static_cast<size_t>(pid) * a.d + colbase, cs);
  }
  // bias gradient and d(scalar): lanes -> wave -> workgroup, fixed order
#pragma unroll
  for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) bgrad[k] += __shfl_xor(bgrad[k], off);
  }
  float dsum = ds1 + (a.skip ? dsx : 0.f) + bdot;
  for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off);
  if (lane < LPR) {
#pragma unroll
    for (int k = 0; k < 4; ++k) red[wave * (FP + 2) + 4 * lane + k] = bgrad[k];
  }
  if (lane == 0) red[wave * (FP + 2) + FP] = dsum;
  __syncthreads();
  if (threadIdx.x < FP) {
    float v = 0.f;
    for (int w = 0; w < nwaves; ++w) v += red[w * (FP + 2) + threadIdx.x];
    a.part_bd[static_cast<size_t>(pid) * FP + threadIdx.x] = v;
  }
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int w = 0; w < nwaves; ++w) v += red[w * (FP + 2) + FP];
    a.part_ds[pid] = v;
  }
}

// ===========================================================================
// launchers
// ===========================================================================
static int tile_nq(int d) {
  static const int kNq[] = {2, 4, 8, 12, 16, 19, 24, 32};
  const int need = ceil_div(d, 16);
  for (int v : kNq)
    if (need <= v) return v;
  return 0;
}

static int tile_threads(int d) {
  int w = ceil_div(d, 64);   // one wave per 64-column chunk in backward phase 4
  if (w < 4) w = 4;
  if (w > 8) w = 8;
  return 64 * w;
}

size_t tile_lds_bytes(int d, int fp, int tile_nodes_max, int tile_edges_max) {
  const int nq = tile_nq(d), ncb = fp >= 16 ? fp / 16 : 1;
  const int cap = ceil_div(tile_nodes_max > 0 ? tile_nodes_max : 1, 16) * 16;
  return static_cast<size_t>(nq) * ncb * 64 * sizeof(f32x4) +              // weight fragments
         static_cast<size_t>(3) * cap * fp * sizeof(float) +                // rank-width tiles
         8 * (fp + 2) * sizeof(float) +                                     // reduction scratch
         (static_cast<size_t>(cap) + 16 + (tile_edges_max > 0 ? tile_edges_max : 1)) * 4 + 64;
}

bool tile_path_ok(const gconv_graph_t* g, int d, int fp) {
  if (g->n_tiles <= 0 || tile_nq(d) == 0 || fp > 32 || ceil_div(d, 64) > 8) return false;
  static const bool off = getenv("GCONV_NO_TILE_PATH") != nullptr;  // debugging / A-B knob
  if (off) return false;
  return tile_lds_bytes(d, fp, g->tile_nodes_max, g->tile_edges_max) <= 64 * 1024;
}

template <int FP, int NQ>
static gconv_status_t launch_tile_fwd_t(const TileFwdArgs& a, int grid, int threads, size_t lds,
                                        hipStream_t stream) {
  if (a.dis) tile_fwd_kernel<FP, NQ, true><<<grid, threads, lds, stream>>>(a);
  else tile_fwd_kernel<FP, NQ, false><<<grid, threads, lds, stream>>>(a);
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

template <int FP, int NQ>
static gconv_status_t launch_tile_bwd_t(const TileBwdArgs& a, int grid, int threads, size_t lds,
                                        hipStream_t stream) {
  if (a.dis) {
    if (a.dx) tile_bwd_kernel<FP, NQ, true, true><<<grid, threads, lds, stream>>>(a);
    else tile_bwd_kernel<FP, NQ, true, false><<<grid, threads, lds, stream>>>(a);
  } else {
    if (a.dx) tile_bwd_kernel<FP, NQ, false, true><<<grid, threads, lds, stream>>>(a);
    else tile_bwd_kernel<FP, NQ, false, false><<<grid, threads, lds, stream>>>(a);
  }
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

#define GCONV_TILE_DISPATCH(FN, ARGS)                                                   \
  switch (a.fp * 100 + nq) {                                                            \
    /* molecule ranks first; every (Fp, NQ) pair here is a compiled kernel */           \
    GCONV_TILE_CASES(FN, ARGS, 4) GCONV_TILE_CASES(FN, ARGS, 8)                          \
    GCONV_TILE_CASES(FN, ARGS, 16) GCONV_TILE_CASES(FN, ARGS, 32)                        \
    default: return GCONV_E_UNSUPPORTED;                                                \
  }
#define GCONV_TILE_CASES(FN, ARGS, FP_)                                                 \
  case FP_ * 100 + 2:  return FN<FP_, 2> ARGS;  case FP_ * 100 + 4:  return FN<FP_, 4> ARGS;   \
  case FP_ * 100 + 8:  return FN<FP_, 8> ARGS;  case FP_ * 100 + 12: return FN<FP_, 12> ARGS;  \
  case FP_ * 100 + 16: return FN<FP_, 16> ARGS; case FP_ * 100 + 19: return FN<FP_, 19> ARGS;  \
  case FP_ * 100 + 24: return FN<FP_, 24> ARGS; case FP_ * 100 + 32: return FN<FP_, 32> ARGS;

int tile_grid(int n_tiles) {  // persistent: <= 2 workgroups per CU
  return n_tiles < 2 * kNumCU ? (n_tiles > 0 ? n_tiles : 1) : 2 * kNumCU;
}

gconv_status_t launch_tile_fwd(TileFwdArgs a, int n_tiles, int tile_nodes_max,
                               int tile_edges_max, hipStream_t stream) {
  const int nq = tile_nq(a.d);
  a.n_tiles = n_tiles;
  n_tiles = tile_grid(n_tiles);
  a.cap = ceil_div(tile_nodes_max > 0 ? tile_nodes_max : 1, 16) * 16;
  const size_t lds = tile_lds_bytes(a.d, a.fp, tile_nodes_max, tile_edges_max);
  const int threads = tile_threads(a.d);
  GCONV_TILE_DISPATCH(launch_tile_fwd_t, (a, n_tiles, threads, lds, stream))
}

gconv_status_t launch_tile_bwd(TileBwdArgs a, int n_tiles, int tile_nodes_max,
                               int tile_edges_max, hipStream_t stream) {
  const int nq = tile_nq(a.d);
  a.n_tiles = n_tiles;
  n_tiles = tile_grid(n_tiles);
  a.cap = ceil_div(tile_nodes_max > 0 ? tile_nodes_max : 1, 16) * 16;
  const size_t lds = tile_lds_bytes(a.d, a.fp, tile_nodes_max, tile_edges_max);
  const int threads = tile_threads(a.d);
  GCONV_TILE_DISPATCH(launch_tile_bwd_t, (a, n_tiles, threads, lds, stream))
}

}  // namespace gconv
