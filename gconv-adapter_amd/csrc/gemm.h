// Internal interface of the skinny projection kernels (gemm.hip <-> adapter.hip).
#pragma once

#include "common.h"

namespace gconv {

// ---- contraction over the hidden dimension d: out[N, Fp] = f(X[N, d] * Wt[d, Fp])
enum RowsAMode {
  ROWSA_DOWN = 0,  // K1: Wt(k, c) = W_down[c, k];  out = row_scale * acc
  ROWSA_DP = 1     // B1: Wt(k, c) = W_up[k, c];    out = row_scale * s * acc, ds1 += acc . P
};

struct RowsAArgs {
  const float* X;          // [n, d]
  const float* W;          // W_down [r, d] or W_up [d, r]
  const float* row_scale;  // dis [n] or nullptr
  const float* scalar;     // device scalar s or nullptr (ROWSA_DP)
  const float* P;          // [n, Fp] saved P (ROWSA_DP, for the scalar gradient) or nullptr
  float* out;              // [n, Fp]
  float* ds_partial;       // [grid] (ROWSA_DP with P) or nullptr
  int n, d, r, fp;
  int mode;
};

size_t rowsA_lds_bytes(int d, int fp);
constexpr size_t kRowsAMaxLds = 160 * 1024;   // the whole LDS of a CU (hidden x padded rank <= 40960 floats)
int rowsA_grid(int n);
gconv_status_t launch_rowsA(const RowsAArgs& a, hipStream_t stream);

// ---- K3: out[N, d] = s * (P[N, Fp] * W_up^T + b_up + (skip ? X : 0))
struct UpArgs {
  const float* P;
  const float* W_up;    // [d, r]
  const float* b_up;    // [d] or nullptr
  const float* scalar;  // device scalar or nullptr
  const float* X;       // [n, d] (read when skip)
  float* out;           // [n, d]
  int n, d, r, fp;
  int skip;
};
gconv_status_t launch_up_fwd(const UpArgs& a, hipStream_t stream);

// ---- B4: dX, per-workgroup partials of dW_down^T, dW_up^T, colsum(G), sum(G.X)
struct BwdMainArgs {
  const float* G;       // [n, d] upstream gradient
  const float* X;       // [n, d]
  const float* T;       // [n, Fp]  S^T (S^T dP . act')
  const float* P;       // [n, Fp]  saved P
  const float* W_down;  // [r, d]
  const float* scalar;  // device scalar or nullptr
  float* dX;            // [n, d] or nullptr
  float* part_wd;       // [nparts, Fp, d]
  float* part_wu;       // [nparts, Fp, d]
  float* part_cs;       // [nparts, d]
  float* part_dsx;      // [nparts * nchunks]
  int n, d, r, fp;
  int skip;
};
struct BwdMainGeom {
  int nchunks, nslots, grid, nparts;
};
BwdMainGeom bwd_main_geom(int n, int d);
gconv_status_t launch_bwd_main(const BwdMainArgs& a, hipStream_t stream);

// ---- B5: deterministic reduction of all partials into the parameter gradients
struct FinalizeArgs {
  const float* part_wd; const float* part_wu; const float* part_cs; int nparts;
  const float* part_dsx; int n_dsx;
  const float* part_ds1; int n_ds1;
  const float* part_bd;  int n_bd;     // [n_bd, fp] from the EPI_BWD1 SpMM
  const float* scalar; const float* b_up;
  float* cs_tmp;                       // [d] scratch: colsum(G)
  float* g_wd; float* g_bd; float* g_wu; float* g_bu; float* g_s;
  int d, r, fp, skip;
  int ds_complete;   // part_ds1 already holds the whole d(scalar) partial (fused tile path)
};
gconv_status_t launch_finalize(const FinalizeArgs& a, hipStream_t stream);
// up to GCONV_FINALIZE_BATCH reductions per launch (same arithmetic, same order per entry)
gconv_status_t launch_finalize_batched(const FinalizeArgs* a, int count, hipStream_t stream);

// [rows, width] -> [kReduceBlocks, width] column sums in a fixed order (width in 4..64)
constexpr int kReduceBlocks = 32;
gconv_status_t launch_reduce_rows(const float* in, int rows, int width, float* out,
                                  hipStream_t stream);

// ===========================================================================
// Shared helpers for the L-layout kernels
// ===========================================================================
// MFMA k index that lane-quarter t feeds in k-step s of a rank-width operand:
//   FP >= 16: float4 fragments, k = 16 (s / 4) + 4 t + (s % 4);  FP < 16: k = 4 s + t
template <int FP>
__device__ __forceinline__ int rank_k(int s, int t) {
  return FP >= 16 ? 16 * (s >> 2) + 4 * t + (s & 3) : 4 * s + t;
}

// A fragments of a rank-width matrix Z[N, FP] for node row `row`:
// frag[s] = Z[row][rank_k(s, t)], s = 0 .. FP/4 - 1.
template <int FP>
__device__ __forceinline__ void load_rank_frag(const float* __restrict__ Z, int row, int t,
                                               float (&frag)[FP / 4]) {
  const float* p = Z + static_cast<size_t>(row) * FP;
  if constexpr (FP >= 16) {
#pragma unroll
    for (int q = 0; q < FP / 16; ++q) {
      const f32x4 v = ldg4(p + 16 * q + 4 * t);
      frag[4 * q + 0] = v[0]; frag[4 * q + 1] = v[1];
      frag[4 * q + 2] = v[2]; frag[4 * q + 3] = v[3];
    }
  } else {
#pragma unroll
    for (int s = 0; s < FP / 4; ++s) frag[s] = p[4 * s + t];
  }
}

// waves of one workgroup = nchunks (64-column chunks of d) x nslots (node-block slots)
inline int chunk_slots(int nchunks, int target_waves) {
  const int s = target_waves / nchunks;
  return s < 1 ? 1 : s;
}

// Loads the four L-layout rows (i0 + 4 t + m, m = 0..3) of a [n, d] matrix.
__device__ __forceinline__ void load_tile_L(const float* __restrict__ M, int i0, int t, int n,
                                            int d, int colbase, bool colvalid, f32x4 (&v)[4]) {
  // clamped address + select instead of a per-lane branch around the load (a branch makes
  // hipcc wait vmcnt(0) after every load); n == 0 means "operand not used": all zeros.
  const int col = min(colbase, d - 4);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int row = i0 + 4 * t + m;
    const f32x4 x = ldg4(M + static_cast<size_t>(min(row, max(n, 1) - 1)) * d + col);
    v[m] = (colvalid && row < n) ? x : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}


}  // namespace gconv
