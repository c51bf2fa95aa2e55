// Internal interface of the gather-fused projection kernels (fused_gather.hip <-> adapter.hip):
// the general path's last propagation of each pass runs INSIDE the projection kernel that
// consumes it, so the rank-width product never makes a round trip through HBM between two
// launches and the random gathers overlap the kernel's [N, d] streams.
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

// B3 + B4:  T = S^T dH1' gathered per node block into LDS  ->  bwd_main's products.
struct BwdGatherArgs {
  const int32_t* rowptr; const int32_t* col;   // CSR of S^T
  const float* dis;
  const float* H;        // [n, Fp] dH1' (pre-scaled)
  float* T;              // [n, Fp] scratch: long rows' T (read); short rows are not stored
  BwdMainArgs main;      // main.T unused
};
gconv_status_t launch_bwd_gather(const BwdGatherArgs& a, hipStream_t stream);

// the fused launches are used for graphs of up to GCONV_FUSED_GATHER_MAX_NODES nodes (default
// 65536) unless GCONV_NO_FUSED_GATHER is set (A/B knob: the separate K2' / K3 and B3 / B4 launches)
bool fused_gather_enabled(int num_nodes);

}  // namespace gconv
