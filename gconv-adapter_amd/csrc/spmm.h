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
};

// rows of `bias_partial` an EPI_BWD1 launch writes for this graph / width (`rows` of them
// produced by the launch: the whole graph, or one row slab of it)
int spmm_bias_partial_rows(const gconv_csr_t* g, int width);
int spmm_bias_partial_rows(const gconv_csr_t* g, int width, int rows);

gconv_status_t launch_spmm(const gconv_csr_t* g, SpmmArgs a, int width, int epi,
                           hipStream_t stream);

}  // namespace gconv
