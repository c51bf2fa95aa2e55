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
int rowsA_grid(int n, int d, int fp);   // also the number of ds_partial entries a ROWSA_DP launch writes
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
  const float* b_up;    // [d] or nullptr: its share b_up . colsum(G) of d(scalar) rides in part_dsx
  float* dX;            // [n, d] or nullptr
  float* part_wd;       // [nparts, Fp, d]
  float* part_wu;       // [nparts, Fp, d]
  float* part_cs;       // [nparts, d]
  float* part_dsx;      // [nparts * nchunks * nslots]: per-wave sum(G . X) [skip] + b_up . colsum(G)
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


// sum(G . X) contribution of one L-layout row pair, as an explicit fma chain: B4 exists twice
// (bwd_main_kernel, bwd_gather_kernel) and hipcc contracts `a*b + c*d + ...` differently from
// kernel to kernel — spelled out, both produce the same bits.
__device__ __forceinline__ float dot4_acc(f32x4 g, f32x4 x, float acc) {
  return __builtin_fmaf(g[0], x[0], __builtin_fmaf(g[1], x[1], __builtin_fmaf(g[2], x[2], __builtin_fmaf(g[3], x[3], acc))));
}

// LDS of B4's final slot reduction ([2 FP + 1][nchunks][64] floats when a chunk has several slots)
inline size_t bwd_main_lds_bytes(int nchunks, int nslots, int fp) {
  if (nslots == 1) return 0;
  return (static_cast<size_t>(2) * fp + 1) * nchunks * 64 * sizeof(float);
}

// End of B4 (shared by bwd_main_kernel and the gather-fused bwd_gather_kernel): lane-quarter sums
// of colsum(G) and sum(G.X), then the slots of each chunk added through LDS in slot order and ONE
// partial per workgroup written out.
template <int FP>
__device__ __forceinline__ void bwd_main_tail(const BwdMainArgs& a, f32x4 (&awd)[FP >= 16 ? FP / 16 : 1][4],
                                              f32x4 (&awu)[FP >= 16 ? FP / 16 : 1][4], f32x4 cs, float dsx,
                                              float* red, int nchunks, int nslots, int J, int slot, int jj,
                                              int t, int lane, int w, int colbase, bool colvalid) {
  constexpr int NCB = FP >= 16 ? FP / 16 : 1;
  // colsum(G): add the four lane quarters (fixed order); every quarter then holds the sum
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    cs[c] += __shfl_xor(cs[c], 16);
    cs[c] += __shfl_xor(cs[c], 32);
  }
  // d(scalar) also holds b_up . colsum(G): this wave's share of the column sums is final here, so
  // its dot product with b_up joins the wave's sum(G . X) partial (one lane quarter only: all four
  // hold the same sums) and the finalize launch never has to re-reduce colsum(G) for it
  if (a.b_up && colvalid && t == 0) dsx = dot4_acc(ldg4(a.b_up + colbase), cs, dsx);
  for (int off = 32; off > 0; off >>= 1) dsx += __shfl_xor(dsx, off);
  if (lane == 0) a.part_dsx[static_cast<size_t>(blockIdx.x) * nchunks * nslots + w] = dsx;

  // ---- add the slots of each chunk in slot order, then one partial per workgroup
  const int pid = blockIdx.x;
  if (nslots > 1) {
    float* rwd = red;                                   // [FP][nchunks * 64]
    float* rwu = red + static_cast<size_t>(FP) * nchunks * 64;
    float* rcs = red + static_cast<size_t>(2) * FP * nchunks * 64;
    const int lc = J * 64 + 4 * jj;                     // column inside the LDS row
    for (int s = 0; s < nslots; ++s) {
      if (slot == s) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int rk = 16 * cb + 4 * t + m;
            if (rk < FP) {
              float* pd = rwd + static_cast<size_t>(rk) * nchunks * 64 + lc;
              float* pu = rwu + static_cast<size_t>(rk) * nchunks * 64 + lc;
              f32x4 vd = f32x4{awd[cb][0][m], awd[cb][1][m], awd[cb][2][m], awd[cb][3][m]};
              f32x4 vu = f32x4{awu[cb][0][m], awu[cb][1][m], awu[cb][2][m], awu[cb][3][m]};
              if (s > 0) {
                vd += *reinterpret_cast<f32x4*>(pd);
                vu += *reinterpret_cast<f32x4*>(pu);
              }
              *reinterpret_cast<f32x4*>(pd) = vd;
              *reinterpret_cast<f32x4*>(pu) = vu;
            }
          }
        if (t == 0) {
          f32x4 v = cs;
          if (s > 0) v += *reinterpret_cast<f32x4*>(rcs + lc);
          *reinterpret_cast<f32x4*>(rcs + lc) = v;
        }
      }
      __syncthreads();
    }
    // cooperative, coalesced write-out of the reduced tiles
    const int width = nchunks * 64;
    for (int idx = threadIdx.x; idx < FP * width; idx += blockDim.x) {
      const int rk = idx / width, col = idx % width;
      if (col < a.d) {
        const size_t o = (static_cast<size_t>(pid) * FP + rk) * a.d + col;
        a.part_wd[o] = rwd[idx];
        a.part_wu[o] = rwu[idx];
      }
    }
    for (int col = threadIdx.x; col < width; col += blockDim.x)
      if (col < a.d) a.part_cs[static_cast<size_t>(pid) * a.d + col] = rcs[col];
  } else {
    if (colvalid) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int rk = 16 * cb + 4 * t + m;  // D row = rank index
          if (rk < FP) {
            const size_t o = (static_cast<size_t>(pid) * FP + rk) * a.d + colbase;
            stg4(a.part_wd + o, f32x4{awd[cb][0][m], awd[cb][1][m], awd[cb][2][m], awd[cb][3][m]});
            stg4(a.part_wu + o, f32x4{awu[cb][0][m], awu[cb][1][m], awu[cb][2][m], awu[cb][3][m]});
          }
        }
      }
      if (t == 0) stg4(a.part_cs + static_cast<size_t>(pid) * a.d + colbase, cs);
    }
  }
}

}  // namespace gconv
