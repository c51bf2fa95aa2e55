// Internal interface of the gather-fused projection kernels (fused_gather.hip <-> adapter.hip):
// the general path's last propagation of each pass runs INSIDE the projection kernel that
// consumes it (one launch less each way; the rank-width rows go from the gather to the MFMA
// through LDS).
#pragma once

#include "gemm.h"

namespace gconv {

// K2' + K3:  P = S H2'  (rows of P gathered per 16-node block, kept in LDS, also stored for the
// backward)  ->  out = s (P W_up^T + b_up + X).
// Rows with more than GCONV_LONG_ROW edges are NOT gathered here: their P rows must already be
// in `P` (launch_spmm(..., seg_only = 1) on the same stream before this launch).
struct UpGatherArgs {
  const int32_t* rowptr; const int32_t* col;
  const float* dis;      // nullptr: unnormalised
  const float* H;        // [n, Fp] gathered operand (H2', pre-scaled by dis when normalised)
  float* P;              // [n, Fp] out: short rows; in: long rows
  UpArgs up;             // W_up, b_up, scalar, X, out, n, d, r, fp, skip (up.P unused)
};
gconv_status_t launch_up_gather(const UpGatherArgs& a, hipStream_t stream);

// B3 + B4:  T = S^T dH1' gathered per node super-block into LDS (never stored) -> B4's products
// dX = T W_down + s G, partials of dW_down^T = T^T X, dW_up^T = P^T G, colsum(G), sum(G . X).
// Long rows: their T rows must already be in `T` (launch_spmm(..., seg_only = 1) in front).
// Same workgroup geometry and partial layout as launch_bwd_main (bwd_main_geom).
struct BwdGatherArgs {
  const int32_t* rowptr; const int32_t* col;   // CSR of S^T
  const float* dis;
  const float* H;        // [n, Fp] dH1' (pre-scaled by dis when normalised)
  const float* T;        // [n, Fp] long rows' T (read only for those rows)
  BwdMainArgs main;      // main.T unused
};
gconv_status_t launch_bwd_gather(const BwdGatherArgs& a, hipStream_t stream);

// the fused launches are used for graphs WITHOUT long rows whose [N, hidden] operand has at most
// 2.5 M elements (10 MB), unless GCONV_NO_FUSED_GATHER is set (A/B knob: the separate K2' / K3 and
// B3 / B4 launches); GCONV_FUSED_GATHER_MAX_NODES, when set, replaces both conditions by a node limit
bool fused_gather_enabled(const gconv_csr_t* graph, int hidden);

}  // namespace gconv
