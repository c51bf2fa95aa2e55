// K0 — graph preparation: COO edge list -> CSR (+ deg^-1/2, long-row segments).
//
// Replaces what PyG's gcn_norm recomputes on every GCNConv call
// (add_remaining_self_loops, degree scatter, deg^-1/2; reached from
// /root/reference/src/finetune/gconv_adapter.py:92) with a one-off build per
// batch that all adapters of a model forward share.
//
// Pipeline (all on `stream`):
//   1. make_keys   : key = (row << 32) | col, dropped / invalid edges -> row = N
//   2. radix sort  : hipCUB DeviceRadixSort over 32 + bits(N) key bits
//   3. build_csr   : rowptr from key run boundaries, col = low key word
//   4. row_stats   : dis = (deg + 1)^-1/2, long-row / segment lists, max degree
//   5. one D2H copy of the counters + stream sync
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace gconv {

enum { INFO_NNZ = 0, INFO_NLONG = 1, INFO_NSEG = 2, INFO_MAXDEG = 3, INFO_BAD = 4, INFO_WORDS = 8 };

__global__ void make_keys_kernel(const int64_t* __restrict__ ei, int64_t E, int32_t n,
                                 int normalize, int transpose, uint64_t* __restrict__ keys,
                                 int32_t* __restrict__ info) {
  int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t s = ei[e], d = ei[E + e];
  uint64_t key;
  if (s < 0 || s >= n || d < 0 || d >= n) {
    atomicOr(&info[INFO_BAD], 1);
    key = static_cast<uint64_t>(n) << 32;
  } else if (normalize && s == d) {
    key = static_cast<uint64_t>(n) << 32;  // add_remaining_self_loops drops it
  } else {
    uint64_t row = transpose ? s : d, col = transpose ? d : s;
    key = (row << 32) | col;
  }
  keys[e] = key;
}

// One thread per sorted position p in [0, E]: closes the rowptr entries of all
// rows that start between key[p-1] and key[p]; copies the column.
__global__ void build_csr_kernel(const uint64_t* __restrict__ keys, int64_t E, int32_t n,
                                 int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                 int32_t* __restrict__ info) {
  int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p > E) return;
  int64_t r_prev = (p == 0) ? -1 : static_cast<int64_t>(keys[p - 1] >> 32);
  int64_t r_cur = (p == E) ? n : static_cast<int64_t>(keys[p] >> 32);
  for (int64_t r = r_prev + 1; r <= r_cur; ++r) rowptr[r] = static_cast<int32_t>(p);
  if (p < E && r_cur < n) col[p] = static_cast<int32_t>(keys[p] & 0xffffffffu);
  if (r_cur == n && r_prev < n) info[INFO_NNZ] = static_cast<int32_t>(p);
}

__global__ void row_stats_kernel(const int32_t* __restrict__ rowptr, int32_t n,
                                 float* __restrict__ dis, int32_t* __restrict__ long_rows,
                                 int32_t* __restrict__ long_seg_base,
                                 int32_t* __restrict__ seg_long, int32_t* __restrict__ info) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int deg = 0;
  if (i < n) {
    deg = rowptr[i + 1] - rowptr[i];
    if (dis) dis[i] = 1.0f / sqrtf(static_cast<float>(deg + 1));
    if (deg > GCONV_LONG_ROW) {
      int nseg = (deg + GCONV_SEG_EDGES - 1) / GCONV_SEG_EDGES;
      int lr = atomicAdd(&info[INFO_NLONG], 1);
      int base = atomicAdd(&info[INFO_NSEG], nseg);
      long_rows[lr] = i;
      long_seg_base[lr] = base;
      for (int k = 0; k < nseg; ++k) seg_long[base + k] = lr;
    }
  }
  // wave-level max, one atomic per wave
  int m = deg;
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(&info[INFO_MAXDEG], m);
}

__global__ void csr_equal_kernel(const int32_t* __restrict__ ra, const int32_t* __restrict__ ca,
                                 const int32_t* __restrict__ rb, const int32_t* __restrict__ cb,
                                 int32_t n, int32_t nnz, int32_t* __restrict__ differ) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  bool bad = false;
  if (i <= n) bad |= ra[i] != rb[i];
  if (i < nnz) bad |= ca[i] != cb[i];
  if (bad) atomicOr(differ, 1);
}

static int key_bits(int32_t n) {
  int b = 1;
  while ((static_cast<int64_t>(1) << b) <= n) ++b;
  return 32 + b;
}

struct PrepLayout {
  size_t keys_a, keys_b, info, cub, total, cub_bytes;
};

static PrepLayout prep_layout(int64_t E, int32_t n) {
  PrepLayout L{};
  size_t off = 0;
  size_t kb = align_up(static_cast<size_t>(E > 0 ? E : 1) * sizeof(uint64_t), 256);
  L.keys_a = off; off += kb;
  L.keys_b = off; off += kb;
  L.info = off; off += 256;
  size_t cub_bytes = 0;
  hipcub::DeviceRadixSort::SortKeys(nullptr, cub_bytes, static_cast<const uint64_t*>(nullptr),
                                    static_cast<uint64_t*>(nullptr), static_cast<int>(E > 0 ? E : 1),
                                    0, key_bits(n), nullptr);
  L.cub = off; L.cub_bytes = cub_bytes; off += align_up(cub_bytes + 256, 256);
  L.total = off;
  return L;
}

}  // namespace gconv

using namespace gconv;

extern "C" int64_t gconv_graph_max_long_rows(int64_t num_edges) {
  return num_edges / (GCONV_LONG_ROW + 1) + 1;
}

extern "C" int64_t gconv_graph_max_segments(int64_t num_edges) {
  // every long row wastes at most one partially filled segment
  return num_edges / GCONV_SEG_EDGES + gconv_graph_max_long_rows(num_edges) + 1;
}

extern "C" size_t gconv_graph_prep_workspace_bytes(int64_t num_edges, int32_t num_nodes) {
  if (num_edges < 0 || num_nodes < 0) return 0;
  return prep_layout(num_edges, num_nodes).total;
}

extern "C" gconv_status_t gconv_graph_prep(const int64_t* edge_index, int64_t num_edges,
                                           int32_t num_nodes, int32_t normalize,
                                           int32_t transpose, int32_t* rowptr_out,
                                           int32_t* col_out, float* dis_out,
                                           int32_t* long_rows_out, int32_t* long_seg_base_out,
                                           int32_t* seg_long_out, void* workspace,
                                           size_t workspace_bytes, gconv_graph_info_t* info_host,
                                           gconv_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (num_edges < 0 || num_nodes < 0 || !rowptr_out || !info_host || !workspace ||
      !long_rows_out || !long_seg_base_out || !seg_long_out)
    return GCONV_E_INVALID_ARG;
  if (num_edges > 0 && (!edge_index || !col_out)) return GCONV_E_INVALID_ARG;
  if (num_edges >= (static_cast<int64_t>(1) << 31) - 1) return GCONV_E_UNSUPPORTED;
  PrepLayout L = prep_layout(num_edges, num_nodes);
  if (workspace_bytes < L.total) return GCONV_E_WORKSPACE;

  char* ws = static_cast<char*>(workspace);
  uint64_t* keys_a = reinterpret_cast<uint64_t*>(ws + L.keys_a);
  uint64_t* keys_b = reinterpret_cast<uint64_t*>(ws + L.keys_b);
  int32_t* info = reinterpret_cast<int32_t*>(ws + L.info);
  void* cub_tmp = ws + L.cub;

  GCONV_HIP(hipMemsetAsync(info, 0, INFO_WORDS * sizeof(int32_t), stream));
  const int T = 256;
  const uint64_t* sorted = keys_a;
  if (num_edges > 0) {
    make_keys_kernel<<<static_cast<unsigned>((num_edges + T - 1) / T), T, 0, stream>>>(
        edge_index, num_edges, num_nodes, normalize, transpose, keys_a, info);
    GCONV_LAUNCH_CHECK();
    size_t cub_bytes = L.cub_bytes;
    GCONV_HIP(hipcub::DeviceRadixSort::SortKeys(cub_tmp, cub_bytes, keys_a, keys_b,
                                                static_cast<int>(num_edges), 0,
                                                key_bits(num_nodes), stream));
    sorted = keys_b;
  }
  build_csr_kernel<<<static_cast<unsigned>((num_edges + 1 + T - 1) / T), T, 0, stream>>>(
      sorted, num_edges, num_nodes, rowptr_out, col_out, info);
  GCONV_LAUNCH_CHECK();
  if (num_nodes > 0) {
    row_stats_kernel<<<(num_nodes + T - 1) / T, T, 0, stream>>>(
        rowptr_out, num_nodes, dis_out, long_rows_out, long_seg_base_out, seg_long_out, info);
    GCONV_LAUNCH_CHECK();
  }
  int32_t h[INFO_WORDS];
  GCONV_HIP(hipMemcpyAsync(h, info, sizeof(h), hipMemcpyDeviceToHost, stream));
  GCONV_HIP(hipStreamSynchronize(stream));
  info_host->nnz = h[INFO_NNZ];
  info_host->n_long = h[INFO_NLONG];
  info_host->n_seg = h[INFO_NSEG];
  info_host->max_degree = h[INFO_MAXDEG];
  info_host->bad_index = h[INFO_BAD];
  info_host->reserved[0] = info_host->reserved[1] = info_host->reserved[2] = 0;
  return h[INFO_BAD] ? GCONV_E_BAD_INDEX : GCONV_OK;
}

extern "C" gconv_status_t gconv_csr_equal(const int32_t* rowptr_a, const int32_t* col_a,
                                          const int32_t* rowptr_b, const int32_t* col_b,
                                          int32_t num_nodes, int32_t nnz, int32_t* flag_dev,
                                          int32_t* equal_host, gconv_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!rowptr_a || !rowptr_b || !flag_dev || !equal_host || num_nodes < 0 || nnz < 0)
    return GCONV_E_INVALID_ARG;
  if (nnz > 0 && (!col_a || !col_b)) return GCONV_E_INVALID_ARG;
  GCONV_HIP(hipMemsetAsync(flag_dev, 0, sizeof(int32_t), stream));
  int64_t work = (static_cast<int64_t>(num_nodes) + 1 > nnz) ? num_nodes + 1 : nnz;
  csr_equal_kernel<<<static_cast<unsigned>((work + 255) / 256), 256, 0, stream>>>(
      rowptr_a, col_a, rowptr_b, col_b, num_nodes, nnz, flag_dev);
  GCONV_LAUNCH_CHECK();
  int32_t differ = 1;
  GCONV_HIP(hipMemcpyAsync(&differ, flag_dev, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  GCONV_HIP(hipStreamSynchronize(stream));
  *equal_host = differ ? 0 : 1;
  return GCONV_OK;
}
