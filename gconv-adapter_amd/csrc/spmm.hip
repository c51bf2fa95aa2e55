// K2 — CSR SpMM at rank width F in {4, 8, 16, 32, 64} for gfx950.
//
// Replaces MessagePassing.propagate (index_select -> weight mul -> scatter_add,
// float atomics on GPU) + `out + bias` of PyG's GCNConv.forward, reached from
// /root/reference/src/finetune/gconv_adapter.py:92. No atomics: every output
// row is produced by exactly one lane group in a fixed edge order, so results
// are bitwise reproducible.
//
// Work decomposition
//   * a row of F floats is F/4 lanes x float4; a wave carries 64/(F/4) rows.
//     Each lane group walks its row's edge list (8 gathers in flight per lane).
//   * symmetric normalisation never touches an edge weight: the producer
//     stores rows pre-multiplied by dis[j], the kernel adds the implicit
//     self-loop and multiplies the finished row by dis[i]:
//         Y[i] = dis[i] * (sum_j H'[j] + H'[i]),   H'[j] = dis[j] * H[j].
//   * rows longer than GCONV_LONG_ROW edges are skipped by the row kernel and
//     cut into GCONV_SEG_EDGES-edge segments; one wave reduces one segment
//     (extra blocks at the tail of the same launch), and a second small launch
//     adds a row's segment sums in segment order and runs the epilogue.
#include <cstdlib>

#include "spmm.h"

namespace gconv {

// ---------------------------------------------------------------------------
// Epilogue shared by the row path and the long-row completion of the row kernel.
//   `sum` = plain sum over the row's stored edges of the gathered rows.
// Its operands (dis, the row's own H line, bias, saved activation) do not depend on the gathers:
// the row path requests them BEFORE walking the edge list (load_row_operands), so that their
// latency passes under the gathers instead of forming a second dependent round trip at the end
// of every wave's short life (round 2; the epilogue used to issue them itself).
// ---------------------------------------------------------------------------
struct RowOperands {
  float di;
  f32x4 self, bias, saved;
};

template <int F, bool NORM, int EPI>
__device__ __forceinline__ RowOperands load_row_operands(const SpmmArgs& a, int rowc, int sub) {
  RowOperands o;
  const size_t off = static_cast<size_t>(rowc) * F + 4 * sub;
  o.di = 1.f;
  o.self = o.bias = o.saved = zero4();
  if (NORM) {
    o.di = a.dis[rowc];
    o.self = ldg4(a.H + off);
  }
  if (EPI == EPI_GENERIC) {
    if (a.bias) o.bias = ldg4(a.bias + 4 * sub);          // (uniform branch)
  } else if (EPI == EPI_FWD1) {
    if (a.bias) {                                         // clamped addresses, masked in the epilogue
#pragma unroll
      for (int k = 0; k < 4; ++k) o.bias[k] = a.bias[min(4 * sub + k, a.rank - 1)];
    }
  } else if (EPI == EPI_BWD1) {
    if (a.act != GCONV_ACT_NONE) o.saved = ldg4(a.saved + off);
  }
  return o;
}

template <int F, bool NORM, int EPI>
__device__ __forceinline__ void spmm_epilogue(const SpmmArgs& a, int row, int sub, f32x4 sum,
                                              f32x4& bias_grad, const RowOperands& o) {
  const size_t off = static_cast<size_t>(row) * F + 4 * sub;
  const float di = o.di;
  f32x4 v = sum;
  if (NORM) {
    f32x4 self = o.self;
    if (EPI == EPI_GENERIC && !(a.flags & GCONV_SPMM_IN_PRESCALED)) self *= di;
    v = (sum + self) * di;
  }
  if (EPI == EPI_STORE) {
    st_y(a.Y + off, v);
  } else if (EPI == EPI_GENERIC) {
    if (a.bias) v += o.bias;
    for (int k = 0; k < 4; ++k) v[k] = act_fwd(a.act, v[k]);
    if (NORM && (a.flags & GCONV_SPMM_OUT_PRESCALE)) v *= di;
    st_y(a.Y + off, v);
  } else if (EPI == EPI_FWD1) {
    // h1 = v + b_down ; h2 = act(h1) ; store h2 (row-scaled for the next hop)
    for (int k = 0; k < 4; ++k) {
      const int c = 4 * sub + k;
      v[k] = (a.bias && c < a.rank) ? v[k] + o.bias[k] : v[k];
    }
    if (a.act == GCONV_ACT_SILU && a.Y2) stg4(a.Y2 + off, v);
    for (int k = 0; k < 4; ++k) v[k] = act_fwd(a.act, v[k]);
    if (NORM) v *= di;
    st_y(a.Y + off, v);
  } else {  // EPI_BWD1: dh1 = dh2 * act'(h1); accumulate bias grad; store row-scaled dh1
    if (a.act == GCONV_ACT_RELU) {
      const f32x4 h2 = o.saved;  // dis * relu(h1): same sign pattern as relu(h1)
      for (int k = 0; k < 4; ++k) v[k] = h2[k] > 0.f ? v[k] : 0.f;
    } else if (a.act == GCONV_ACT_SILU) {
      const f32x4 h1 = o.saved;
      for (int k = 0; k < 4; ++k) v[k] *= silu_grad(h1[k]);
    }
    bias_grad += v;
    if (NORM) v *= di;
    st_y(a.Y + off, v);
  }
}

// Sums `bias_grad` over the lanes of a block that own the same column slot and
// writes one [F] row of partials per block (EPI_BWD1 only).
template <int F>
__device__ __forceinline__ void block_bias_partial(f32x4 bias_grad, float* part_row) {
  constexpr int LPR = RowShape<F>::LPR;
  __shared__ float red[4][F];
  const int lane = threadIdx.x & 63, wave = wave_id();
#pragma unroll
  for (int off = LPR; off < kWave; off <<= 1) {
    for (int k = 0; k < 4; ++k) bias_grad[k] += __shfl_xor(bias_grad[k], off);
  }
  if (lane < LPR) {
    for (int k = 0; k < 4; ++k) red[wave][4 * lane + k] = bias_grad[k];
  }
  __syncthreads();
  if (threadIdx.x < F) {
    part_row[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] +
                            red[3][threadIdx.x];
  }
}

// ---------------------------------------------------------------------------
// Row kernel: blocks [0, seg_blocks) reduce long-row segments (one wave per segment; they
// carry the longest dependent-load chain, so they are dispatched FIRST and run under the row
// blocks instead of forming the tail of the launch), blocks [seg_blocks, grid) walk rows.
// ---------------------------------------------------------------------------
template <int F, bool NORM, int EPI>
__global__ __launch_bounds__(256) void spmm_rows_kernel(SpmmArgs a) {
  constexpr int LPR = RowShape<F>::LPR;
  constexpr int RPW = RowShape<F>::RPW;
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int sub = lane % LPR, grp = lane / LPR;
  const bool gather_scale = NORM && EPI == EPI_GENERIC && !(a.flags & GCONV_SPMM_IN_PRESCALED);
  const float* __restrict__ H = a.H;
  const int32_t* __restrict__ col = a.col;

  const int seg_blocks = (a.n_seg + 3) >> 2;
  if (static_cast<int>(blockIdx.x) < seg_blocks) {
    // ---- long-row segment: the 64/LPR lane groups stride over the segment's edges. The wave
    // that completes a row (last arrival at the row's counter) adds the row's segment sums in
    // segment order and runs the epilogue: no second launch, still no float atomics, and the
    // result does not depend on which wave arrives last.
    f32x4 bias_grad = zero4();
    const int s = blockIdx.x * 4 + wave;
    if (s < a.n_seg) {   // wave-uniform
      const int lr = a.seg_long[s];
      const int row = a.long_rows[lr];
      const int base = a.long_seg_base[lr];
      if (row < a.row_lo || row >= a.row_hi) {   // another device's row (wave-uniform)
        if (EPI == EPI_BWD1 && s == base && grp == 0)   // its bias-gradient slot still gets summed
          stg4(a.bias_partial + static_cast<size_t>(a.row_blocks + lr) * F + 4 * sub, zero4());
        return;
      }
      const int r0 = a.rowptr[row], r1 = a.rowptr[row + 1];
      const int e0 = r0 + (s - base) * GCONV_SEG_EDGES;
      const int e1 = min(e0 + GCONV_SEG_EDGES, r1);
      const int nseg = (r1 - r0 + GCONV_SEG_EDGES - 1) / GCONV_SEG_EDGES;
      f32x4 acc = zero4();
#pragma unroll 4
      for (int e = e0 + grp; e < e1; e += RPW) {
        int c = ld_col(col + e);
#if GCONV_SPMM_SEG_NT
        f32x4 h = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(H + static_cast<size_t>(c) * F + 4 * sub));
#else
        f32x4 h = ldg4(H + static_cast<size_t>(c) * F + 4 * sub);
#endif
        if (gather_scale) h *= a.dis[c];
        acc += h;
      }
#pragma unroll
      for (int off = LPR; off < kWave; off <<= 1) {
        for (int q = 0; q < 4; ++q) acc[q] += __shfl_xor(acc[q], off);
      }
      // publish the partial: write-through (sc1) store — the other segments of the row may run
      // on other XCDs, whose L2s are private — drained before the ticket is taken; the wave
      // that completes the row reads all partials with sc1 loads (no fences on either side)
      const rsrc_t rpart = make_rsrc(a.seg_partial, static_cast<uint32_t>(a.n_seg) * F * 4u);
      if (grp == 0)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, acc), rpart, (s * F + 4 * sub) * 4, 0, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned ticket = 0;
      if (lane == 0)
        ticket = __hip_atomic_fetch_add(a.long_sync + lr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ticket = __builtin_amdgcn_readfirstlane(ticket);
      if (ticket == static_cast<unsigned>(nseg - 1)) {   // wave-uniform: every segment has landed
        f32x4 sum = zero4();
        for (int k = grp; k < nseg; k += RPW)
          sum += __builtin_bit_cast(
              f32x4, __builtin_amdgcn_raw_buffer_load_b128(rpart, ((base + k) * F + 4 * sub) * 4, 0, 16));
#pragma unroll
        for (int off = LPR; off < kWave; off <<= 1) {
          for (int q = 0; q < 4; ++q) sum[q] += __shfl_xor(sum[q], off);
        }
        if (grp == 0) {
          const RowOperands ops = load_row_operands<F, NORM, EPI>(a, row, sub);
          spmm_epilogue<F, NORM, EPI>(a, row, sub, sum, bias_grad, ops);
          // the row's bias-gradient term goes to the row's OWN partial slot, whichever wave
          // completes it: the later reduction adds the same numbers in the same order every run
          if (EPI == EPI_BWD1) stg4(a.bias_partial + static_cast<size_t>(a.row_blocks + lr) * F + 4 * sub, bias_grad);
        }
        if (lane == 0)   // ready for the next launch on this CSR
          __hip_atomic_store(a.long_sync + lr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }

  f32x4 bias_grad = zero4();
  int rblock = blockIdx.x - seg_blocks;
  if (a.xcd_rows) {
    // Local neighbourhoods (gconv_csr_t::near_edges): workgroups are dealt round-robin over the eight XCDs, so
    // with consecutive row blocks on consecutive XCDs every one of the eight private L2s ends up fetching every
    // H row (banded control: L2 hit 0.47, 7 x the table through the fabric). Blocks b and b + 8 share an XCD:
    // they take CONSECUTIVE row blocks of one contiguous eighth instead. A permutation of the row blocks — a
    // row's sum, and the bias-partial slot of its block, do not depend on it (same bits).
    const int per = a.row_blocks >> 3;
    if (rblock < 8 * per) rblock = (rblock & 7) * per + (rblock >> 3);
  }
  const int row = a.row_lo + (rblock * 4 + wave) * RPW + grp;
  const int rowc = min(row, a.row_hi - 1);
  const int e0 = a.rowptr[rowc], e1 = a.rowptr[rowc + 1];
  const bool mine = row < a.row_hi && (e1 - e0) <= GCONV_LONG_ROW;  // long rows: segment path
  const int deg = mine ? e1 - e0 : 0;

  const RowOperands ops = load_row_operands<F, NORM, EPI>(a, rowc, sub);   // in flight under the gathers
  const f32x4 sum = row_gather_sum<F>(H, col, e0, e1, deg, rowc, sub, grp, gather_scale, a.dis);
  if (mine) spmm_epilogue<F, NORM, EPI>(a, row, sub, sum, bias_grad, ops);
  if (EPI == EPI_BWD1) block_bias_partial<F>(bias_grad, a.bias_partial + static_cast<size_t>(rblock) * F);
}

template <int F, bool NORM, int EPI>
static gconv_status_t launch_spmm_t(SpmmArgs& a, hipStream_t stream) {
  constexpr int RPW = RowShape<F>::RPW;
  a.row_blocks = ceil_div(a.row_hi - a.row_lo, 4 * RPW);
  const int seg_blocks = ceil_div(a.n_seg, 4);
  const int grid = (a.seg_only ? 0 : a.row_blocks) + seg_blocks;
  if (grid > 0) {
    spmm_rows_kernel<F, NORM, EPI><<<grid, 256, 0, stream>>>(a);
    GCONV_LAUNCH_CHECK();
  }
  return GCONV_OK;
}

template <int F, bool NORM>
static gconv_status_t launch_spmm_e(SpmmArgs& a, int epi, hipStream_t stream) {
  switch (epi) {
    case EPI_GENERIC: return launch_spmm_t<F, NORM, EPI_GENERIC>(a, stream);
    case EPI_STORE:   return launch_spmm_t<F, NORM, EPI_STORE>(a, stream);
    case EPI_FWD1:    return launch_spmm_t<F, NORM, EPI_FWD1>(a, stream);
    case EPI_BWD1:    return launch_spmm_t<F, NORM, EPI_BWD1>(a, stream);
  }
  return GCONV_E_INVALID_ARG;
}

template <int F>
static gconv_status_t launch_spmm_n(SpmmArgs& a, int epi, hipStream_t stream) {
  return a.dis ? launch_spmm_e<F, true>(a, epi, stream) : launch_spmm_e<F, false>(a, epi, stream);
}

int spmm_bias_partial_rows(const gconv_csr_t* g, int width, int rows) {
  const int rpw = kWave / (width / 4);
  return ceil_div(rows, 4 * rpw) + g->n_long;
}
int spmm_bias_partial_rows(const gconv_csr_t* g, int width) {
  return spmm_bias_partial_rows(g, width, g->num_nodes);
}

gconv_status_t launch_spmm(const gconv_csr_t* g, SpmmArgs a, int width, int epi,
                           hipStream_t stream) {
  a.n = g->num_nodes;
  a.rowptr = g->rowptr;
  a.col = g->col;
  a.n_long = g->n_long;
  a.n_seg = g->n_seg;
  a.long_rows = g->long_rows;
  a.long_seg_base = g->long_seg_base;
  a.seg_long = g->seg_long;
  a.long_sync = g->long_sync;
  if (a.n == 0) return GCONV_OK;
  // at least half of the stored edges stay inside their row's eighth of the node range, and there are enough
  // row blocks for the eighths to matter: XCD-contiguous rows (GCONV_SPMM_XCD_ROWS=0 / 1 forces it off / on)
  static const char* xr = getenv("GCONV_SPMM_XCD_ROWS");
  a.xcd_rows = xr ? (xr[0] == '1') : (2 * static_cast<int64_t>(g->near_edges) >= g->nnz && g->nnz > 0 && a.n >= kXcdRowsMinNodes);
  if (a.row_hi == 0) { a.row_lo = 0; a.row_hi = a.n; }
  if (a.row_lo < 0 || a.row_hi > a.n || a.row_lo >= a.row_hi) return GCONV_E_INVALID_ARG;
  if (a.n_seg > 0 && !a.seg_partial) return GCONV_E_WORKSPACE;
  switch (width) {
    case 4:  return launch_spmm_n<4>(a, epi, stream);
    case 8:  return launch_spmm_n<8>(a, epi, stream);
    case 16: return launch_spmm_n<16>(a, epi, stream);
    case 32: return launch_spmm_n<32>(a, epi, stream);
    case 64: return launch_spmm_n<64>(a, epi, stream);
  }
  return GCONV_E_UNSUPPORTED;
}

}  // namespace gconv

using namespace gconv;

extern "C" size_t gconv_spmm_workspace_bytes(const gconv_csr_t* graph, int32_t width) {
  if (!graph || width <= 0) return 0;
  return align_up(static_cast<size_t>(graph->n_seg) * width * sizeof(float) + 16, 256);
}

extern "C" gconv_status_t gconv_spmm_csr_f32(const gconv_csr_t* graph, const float* dis,
                                             const float* H, float* Y, int32_t width,
                                             const float* bias, int32_t act, uint32_t flags,
                                             void* workspace, size_t workspace_bytes,
                                             gconv_stream_t stream) {
  if (!graph || !graph->rowptr || (graph->nnz > 0 && !graph->col) || graph->num_nodes < 0)
    return GCONV_E_INVALID_ARG;
  if (graph->num_nodes == 0) return GCONV_OK;
  if (!H || !Y) return GCONV_E_INVALID_ARG;
  if (act < GCONV_ACT_NONE || act > GCONV_ACT_SILU) return GCONV_E_INVALID_ARG;
  if (width != 4 && width != 8 && width != 16 && width != 32 && width != 64)
    return GCONV_E_UNSUPPORTED;
  if (!aligned16(H) || !aligned16(Y) || (bias && !aligned16(bias))) return GCONV_E_ALIGNMENT;
  if (graph->n_seg > 0 &&
      (!workspace || workspace_bytes < gconv_spmm_workspace_bytes(graph, width)))
    return GCONV_E_WORKSPACE;
  SpmmArgs a{};
  a.dis = dis;
  a.H = H;
  a.Y = Y;
  a.bias = bias;
  a.act = act;
  a.flags = flags;
  a.rank = width;
  a.seg_partial = static_cast<float*>(workspace);
  return launch_spmm(graph, a, width, EPI_GENERIC, static_cast<hipStream_t>(stream));
}
