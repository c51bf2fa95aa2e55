// Next to the path: the aggregation of the frozen GIN layer that sits between two
// adapters — everything MolecularGINConv.forward does before its MLP
// (/root/reference/src/layers/inductive/gin_conv.py:38-73): add_self_loops, the two bond-embedding
// look-ups (self-loop edges get bond type 4, direction 0), message x_j + e_ij, sum aggregation:
//
//   agg[i] = x[i] + sum_{j -> i} x[j]  +  emb_type[4] + emb_dir[0]
//                                      +  sum_{e: j -> i} (emb_type[type_e] + emb_dir[dir_e])
//
// The reference materialises an [E + N, d] message tensor and scatter-adds it. Here the bond term
// does not need the edges at all: there are only 6 x 3 = 18 distinct bond embeddings, so a node's
// bond term is sum_c count[i][c] * T[c]; the per-node code counts are built once per BATCH by an
// integer histogram (exact, order-free) and shared by every GIN layer, and the neighbour sum walks
// the unnormalised CSR of the batch's graph plan — the same plan the adapters use. One wave per
// row, a row of d floats as float4 lanes (d = 300: 75 of them, two passes), fixed summation
// order, no float atomics. The gradient w.r.t. x (what an earlier adapter needs from a frozen
// layer) is the same kernel on the transposed CSR without the bond term.
#include "mlp_split.h"

namespace gconv {

constexpr int kBondTypes = 6;   // NUM_BOND_TYPE, gin_conv.py:6
constexpr int kBondDirs = 3;    // NUM_BOND_DIRECTION, gin_conv.py:7
constexpr int kBondCodes = kBondTypes * kBondDirs;
constexpr int kSelfLoopType = 4;   // gin_conv.py:54
constexpr int kGinPasses = 4;      // float4 lanes per row / 64 at most: hidden <= 1024

__global__ __launch_bounds__(256) void gin_code_count_kernel(const int64_t* __restrict__ edge_index,
                                                             const int64_t* __restrict__ edge_attr,
                                                             int64_t num_edges, int n,
                                                             int32_t* __restrict__ count) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= num_edges) return;
  const int64_t tgt = edge_index[num_edges + e];
  const int64_t ty = edge_attr[2 * e], di = edge_attr[2 * e + 1];
  if (tgt < 0 || tgt >= n || ty < 0 || ty >= kBondTypes || di < 0 || di >= kBondDirs) {
    atomicAdd(count + static_cast<size_t>(n) * kBondCodes, 1);   // skipped, never a fault
    return;
  }
  atomicAdd(count + static_cast<size_t>(tgt) * kBondCodes + static_cast<int>(ty) * kBondDirs + static_cast<int>(di), 1);
}

// NP = float4 lanes per row / 64 (d = 300: 2). Waves are persistent over rows so that the nine
// bond-embedding rows live in registers (9 x NP float4 per lane): re-reading them per row cost
// 3.6x the row's own L1 traffic (18 table loads against ~5 row loads) and made the bond term as
// expensive as the neighbour sum.
// SPLIT (round 4): the row leaves as the scaled fp16 PAIR the frozen layer's MLP multiplies (mlp_split.h;
// out2 = [2][n][Kp] fp16, rexp[n]) — the fp32 aggregate and the separate split pass over it (a 15.7 MB
// round trip and a launch per layer on a 512-molecule batch) never exist.
template <int NP, bool BOND, bool SPLIT = false>
__global__ __launch_bounds__(256) void gin_aggregate_kernel(const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col,
                                                            const float* __restrict__ x,
                                                            const int32_t* __restrict__ count,
                                                            const float* __restrict__ emb_type,
                                                            const float* __restrict__ emb_dir,
                                                            float* __restrict__ out, int n, int d,
                                                            _Float16* __restrict__ out2 = nullptr,
                                                            int32_t* __restrict__ rexp = nullptr) {
  const int lane = threadIdx.x & 63;
  const int nq = d >> 2;
  f32x4 table[BOND ? (kBondTypes + kBondDirs) : 1][NP];
  if (BOND) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int q = lane + 64 * p;
#pragma unroll
      for (int t = 0; t < kBondTypes + kBondDirs; ++t) {
        const float* src = t < kBondTypes ? emb_type + static_cast<size_t>(t) * d
                                          : emb_dir + static_cast<size_t>(t - kBondTypes) * d;
        table[t][p] = q < nq ? ldg4(src + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  for (int row = blockIdx.x * 4 + wave_id(); row < n; row += gridDim.x * 4) {   // wave-uniform
    const int e0 = uniform(rowptr[row]), e1 = uniform(rowptr[row + 1]);
    int mine = 0;   // the node's 18 code counts: ONE coalesced load, issued before the gathers
    if (BOND) mine = lane < kBondCodes ? count[static_cast<size_t>(row) * kBondCodes + lane] : 0;
    f32x4 acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int q = lane + 64 * p;
      acc[p] = q < nq ? ldg4(x + static_cast<size_t>(row) * d + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int e = e0; e < e1; e += 4) {   // 4 neighbour rows in flight per pass
      int c[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) c[k] = uniform(col[min(e + k, e1 - 1)]);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int q = lane + 64 * p;
        if (q < nq) {
          f32x4 h[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) h[k] = ldg4(x + static_cast<size_t>(c[k]) * d + 4 * q);
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (e + k < e1) acc[p] += h[k];   // uniform condition; CSR order
        }
      }
    }
    if (BOND) {
      // bond term = sum_t n_type[t] * emb_type[t] + sum_k n_dir[k] * emb_dir[k] (self-loop: type
      // 4, direction 0), the 18 code counts folded into 6 + 3 wave-uniform multiplicities
      float nt[kBondTypes], nd[kBondDirs];
#pragma unroll
      for (int k = 0; k < kBondDirs; ++k) nd[k] = 0.f;
#pragma unroll
      for (int t = 0; t < kBondTypes; ++t) {
        nt[t] = 0.f;
#pragma unroll
        for (int k = 0; k < kBondDirs; ++k) {
          const float c = static_cast<float>(__builtin_amdgcn_readlane(mine, t * kBondDirs + k));
          nt[t] += c;
          nd[k] += c;
        }
      }
      nt[kSelfLoopType] += 1.f;
      nd[0] += 1.f;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int t = 0; t < kBondTypes; ++t) acc[p] += table[t][p] * nt[t];
#pragma unroll
        for (int k = 0; k < kBondDirs; ++k) acc[p] += table[kBondTypes + k][p] * nd[k];
      }
    }
    if constexpr (SPLIT) {
      float m = 0.f;
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int k = 0; k < 4; ++k) m = fmaxf(m, fabsf(acc[p][k]));   // (lanes past the row hold zeros)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
      const int e = scale_exp(m);
      const int kp = mlp_pad_k(d);
      const size_t img = static_cast<size_t>(n) * kp;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int q = lane + 64 * p;
        if (4 * q < kp) {   // columns d .. Kp: zeros
          f16x4 h, l;
          split2x4(acc[p], e, h, l);
          const size_t o = static_cast<size_t>(row) * kp + 4 * q;
          *reinterpret_cast<f16x4*>(out2 + o) = h;
          *reinterpret_cast<f16x4*>(out2 + img + o) = l;
        }
      }
      if (lane == 0) rexp[row] = e;
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int q = lane + 64 * p;
        if (q < nq) stg4(out + static_cast<size_t>(row) * d + 4 * q, acc[p]);
      }
    }
  }
}

template <int NP>
static void launch_gin_aggregate_pair(const gconv_csr_t* csr, const float* x, const int32_t* count,
                                      const float* emb_type, const float* emb_dir, void* out_pair, int d,
                                      hipStream_t stream) {
  const int grid = min(ceil_div(csr->num_nodes, 4), launch_cus() * 4);
  _Float16* o = static_cast<_Float16*>(out_pair);
  int32_t* rexp = reinterpret_cast<int32_t*>(static_cast<char*>(out_pair) + mlp_pair_exp_offset(csr->num_nodes, d));
  gin_aggregate_kernel<NP, true, true><<<grid, 256, 0, stream>>>(csr->rowptr, csr->col, x, count, emb_type, emb_dir,
                                                                 nullptr, csr->num_nodes, d, o, rexp);
}

template <int NP>
static void launch_gin_aggregate(const gconv_csr_t* csr, const float* x, const int32_t* count,
                                 const float* emb_type, const float* emb_dir, float* out, int d,
                                 hipStream_t stream) {
  // persistent waves, as many as fit (the rows of a molecule batch are short: what hides their
  // latency is the number of waves, not the work per wave): the bond variant holds 9 x NP table
  // registers per lane (110 VGPRs at NP = 2: 4 workgroups of 4 waves per CU), the plain one 8
  const int grid = min(ceil_div(csr->num_nodes, 4), launch_cus() * (count ? 4 : 8));
  if (count)
    gin_aggregate_kernel<NP, true><<<grid, 256, 0, stream>>>(csr->rowptr, csr->col, x, count, emb_type, emb_dir,
                                                             out, csr->num_nodes, d);
  else
    gin_aggregate_kernel<NP, false><<<grid, 256, 0, stream>>>(csr->rowptr, csr->col, x, nullptr, nullptr, nullptr,
                                                              out, csr->num_nodes, d);
}

}  // namespace gconv

using namespace gconv;

extern "C" size_t gconv_gin_code_count_bytes(int32_t num_nodes) {
  if (num_nodes < 0) return 0;
  return (static_cast<size_t>(num_nodes) * kBondCodes + 1) * sizeof(int32_t);
}

extern "C" gconv_status_t gconv_gin_code_count(const int64_t* edge_index, const int64_t* edge_attr,
                                               int64_t num_edges, int32_t num_nodes,
                                               int32_t* code_count, gconv_stream_t stream_) {
  if (num_edges < 0 || num_nodes < 0 || !code_count || (num_edges > 0 && (!edge_index || !edge_attr)))
    return GCONV_E_INVALID_ARG;
  if (num_edges > (static_cast<int64_t>(1) << 31) * 255) return GCONV_E_UNSUPPORTED;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GCONV_HIP(hipMemsetAsync(code_count, 0, gconv_gin_code_count_bytes(num_nodes), stream));
  if (num_edges == 0) return GCONV_OK;
  const unsigned grid = static_cast<unsigned>((num_edges + 255) / 256);
  gin_code_count_kernel<<<grid, 256, 0, stream>>>(edge_index, edge_attr, num_edges, num_nodes, code_count);
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

extern "C" gconv_status_t gconv_gin_aggregate_f32(const gconv_csr_t* csr, const float* x,
                                                  const int32_t* code_count, const float* emb_type,
                                                  const float* emb_dir, float* out, int32_t hidden,
                                                  gconv_stream_t stream_) {
  if (!csr || !csr->rowptr || (csr->nnz > 0 && !csr->col) || csr->num_nodes < 0) return GCONV_E_INVALID_ARG;
  if (hidden <= 0 || hidden % 4 != 0 || hidden > 256 * kGinPasses) return GCONV_E_UNSUPPORTED;
  if (csr->num_nodes == 0) return GCONV_OK;
  if (!x || !out || (code_count && (!emb_type || !emb_dir))) return GCONV_E_INVALID_ARG;
  if (!aligned16(x) || !aligned16(out) || (code_count && (!aligned16(emb_type) || !aligned16(emb_dir))))
    return GCONV_E_ALIGNMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  switch (ceil_div(hidden, 256)) {
    case 1: launch_gin_aggregate<1>(csr, x, code_count, emb_type, emb_dir, out, hidden, stream); break;
    case 2: launch_gin_aggregate<2>(csr, x, code_count, emb_type, emb_dir, out, hidden, stream); break;
    case 3: launch_gin_aggregate<3>(csr, x, code_count, emb_type, emb_dir, out, hidden, stream); break;
    default: launch_gin_aggregate<4>(csr, x, code_count, emb_type, emb_dir, out, hidden, stream); break;
  }
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

// The same aggregation (bond term included), written as the scaled fp16 pair the frozen MLP's first GEMM
// multiplies (gconv_mlp_gemm_f16x2's `a_pair`; gconv_mlp_rows_pair_bytes(num_nodes, hidden) bytes).
extern "C" gconv_status_t gconv_gin_aggregate_f16x2(const gconv_csr_t* csr, const float* x,
                                                    const int32_t* code_count, const float* emb_type,
                                                    const float* emb_dir, void* out_pair, int32_t hidden,
                                                    gconv_stream_t stream_) {
  if (!csr || !csr->rowptr || (csr->nnz > 0 && !csr->col) || csr->num_nodes < 0) return GCONV_E_INVALID_ARG;
  if (hidden <= 0 || hidden % 4 != 0 || hidden > 256 * kGinPasses) return GCONV_E_UNSUPPORTED;
  if (csr->num_nodes == 0) return GCONV_OK;
  if (!x || !out_pair || !code_count || !emb_type || !emb_dir) return GCONV_E_INVALID_ARG;
  if (!aligned16(x) || !aligned16(out_pair) || !aligned16(emb_type) || !aligned16(emb_dir)) return GCONV_E_ALIGNMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  switch (ceil_div(mlp_pad_k(hidden), 256)) {
    case 1: launch_gin_aggregate_pair<1>(csr, x, code_count, emb_type, emb_dir, out_pair, hidden, stream); break;
    case 2: launch_gin_aggregate_pair<2>(csr, x, code_count, emb_type, emb_dir, out_pair, hidden, stream); break;
    case 3: launch_gin_aggregate_pair<3>(csr, x, code_count, emb_type, emb_dir, out_pair, hidden, stream); break;
    default: launch_gin_aggregate_pair<4>(csr, x, code_count, emb_type, emb_dir, out_pair, hidden, stream); break;
  }
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}
