// Internal interface of the SpMM launcher (shared by spmm.hip and adapter.hip).
#pragma once

#include "common.h"

namespace gconv {

enum SpmmEpilogue {
  EPI_GENERIC = 0,  // public API: Y = act(S H + bias)
  EPI_STORE = 1,    // Y = S H                    (P in forward, T in backward)
  EPI_FWD1 = 2,     // h1 = S H + b_down; Y = dis * act(h1); Y2 = h1 (SiLU only)
  EPI_BWD1 = 3      // Y = dis * (S^T dP) * act'(h1); per-block bias-grad partials
};

struct SpmmArgs {
  // graph (filled by launch_spmm from gconv_csr_t)
  int n;
  const int32_t* rowptr;
  const int32_t* col;
  int n_long, n_seg;
  const int32_t* long_rows;
  const int32_t* long_seg_base;
  const int32_t* seg_long;
  uint32_t* long_sync;
  int row_blocks;
  int row_lo, row_hi;   // rows this launch produces; row_hi == 0 -> all (row slabs: SURVEY 8e)
  // operands
  const float* dis;     // nullptr -> unnormalised
  const float* H;       // [n, F] gathered rows
  float* Y;             // [n, F]
  float* Y2;            // [n, F] optional second output (pre-activation)
  const float* saved;   // [n, F] EPI_BWD1: saved act output (relu) / pre-activation (silu)
  const float* bias;    // [rank]
  float* seg_partial;   // [n_seg, F]
  float* bias_partial;  // [spmm_bias_partial_rows, F]
  int act;
  int rank;             // valid columns (bias length)
  uint32_t flags;
  int seg_only;         // launch the long-row segment blocks only (the fused K2'+K3 / B3+B4 kernels
                        // gather the short rows themselves and read the long rows' results)
  int xcd_rows;         // 1: row block -> XCD map of a graph with local neighbourhoods (set by launch_spmm
                        // from gconv_csr_t::near_edges): a contiguous eighth of the row blocks per XCD
};

template <int F>
struct RowShape {
  static constexpr int LPR = F / 4;        // lanes per row
  static constexpr int RPW = kWave / LPR;  // rows per wave
};

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// Cache policy of the SpMM's streamed-once arrays (round 4 A/B on S-arxiv, rank 32, tools/lab/spmm_policy.sh,
// profiles/r04_lab_spmm_cache_policy.txt): non-temporal so that they do not compete with the gathered H
// rows for the 4 MB L2s.   _Y_NT   the output rows: 26.4 -> 25.9 us, traffic 173.0 -> 170.3 MB, L2 hit
//                                  0.340 -> 0.350: ON in the product build
//                          _COL_NT the column ids: 28.0 us (a row's ids are re-read by its own lanes:
//                                  they want to stay in L1 / L2); _SEG_NT the segment path's gathers (hub rows:
//                                  128 read-once rows per segment wave): 26.9 us; all three: 28.7 us — off
#ifndef GCONV_SPMM_Y_NT
#define GCONV_SPMM_Y_NT 1
#endif
#ifndef GCONV_SPMM_COL_NT
#define GCONV_SPMM_COL_NT 0
#endif
#ifndef GCONV_SPMM_SEG_NT
#define GCONV_SPMM_SEG_NT 0
#endif
__device__ __forceinline__ int ld_col(const int32_t* p) {
#if GCONV_SPMM_COL_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ void st_y(float* p, f32x4 v) {
#if GCONV_SPMM_Y_NT
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
  *reinterpret_cast<f32x4*>(p) = v;
#endif
}

// Plain sum over a row's stored edges of the gathered rows of H — the one summation order of
// every short row (<= GCONV_LONG_ROW edges) in the library: spmm_rows_kernel and the fused
// gather kernels (fused_gather.hip) call this, so a row's result has the same bits whichever
// kernel produced it.
//   lane group `grp` (LPR = F/4 lanes, `sub` = lane inside the group) owns the row; `deg` = edges
//   this group sums (0: none), e0 / e1 = the row's CSR range, rowc = a valid row index (padding
//   gathers read that row's own line and are masked out of the sum).
// The loop over 8-edge batches is WAVE-UNIFORM (trip count = longest row of the wave), so it
// can be software-pipelined without hipcc fencing every load: the 8 column indices of a batch
// are fetched by the row's own lanes (8 / LPR coalesced loads, the next batch's are in flight
// during the gathers) and broadcast by lane shuffle.
template <int F>
__device__ __forceinline__ f32x4 row_gather_sum(const float* __restrict__ H, const int32_t* __restrict__ col,
                                                int e0, int e1, int deg, int rowc, int sub, int grp,
                                                bool gather_scale, const float* __restrict__ dis) {
  constexpr int LPR = RowShape<F>::LPR;
  constexpr int CPL = (8 + LPR - 1) / LPR;  // column loads per lane per batch
  int nbat = (deg + 7) >> 3;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) nbat = max(nbat, __shfl_xor(nbat, off));
  nbat = __builtin_amdgcn_readfirstlane(nbat);
  const int elast = max(e1 - 1, 0);
  int cn[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) cn[j] = ld_col(col + min(e0 + sub + j * LPR, elast));
  f32x4 acc0 = zero4(), acc1 = zero4();
  for (int it = 0; it < nbat; ++it) {
    int cc[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) cc[j] = cn[j];
    const int eb = e0 + 8 * it;
#pragma unroll
    for (int j = 0; j < CPL; ++j) cn[j] = ld_col(col + min(eb + 8 + sub + j * LPR, elast));
    f32x4 h[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int c = __shfl(cc[u / LPR], grp * LPR + (u % LPR));
      c = (eb + u < e0 + deg) ? c : rowc;
      h[u] = ldg4(H + static_cast<size_t>(c) * F + 4 * sub);
      if (gather_scale) h[u] *= dis[c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) h[u] = (eb + u < e0 + deg) ? h[u] : zero4();
    // fixed association order: ((h0+h1)+(h2+h3)) + ((h4+h5)+(h6+h7))
    acc0 += (h[0] + h[1]) + (h[2] + h[3]);
    acc1 += (h[4] + h[5]) + (h[6] + h[7]);
  }
  return acc0 + acc1;
}

// rows of `bias_partial` an EPI_BWD1 launch writes for this graph / width (`rows` of them
// produced by the launch: the whole graph, or one row slab of it)
int spmm_bias_partial_rows(const gconv_csr_t* g, int width);
int spmm_bias_partial_rows(const gconv_csr_t* g, int width, int rows);

gconv_status_t launch_spmm(const gconv_csr_t* g, SpmmArgs a, int width, int epi,
                           hipStream_t stream);

}  // namespace gconv
