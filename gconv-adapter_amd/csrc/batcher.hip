// Device-side collation of molecule batches from a dataset resident in HBM.
//
// Replaces, for datasets that fit in device memory (HIV: 41 k molecules, 1.3 GB of features),
//   PyG DataLoader -> Batch.from_data_list -> batch.to(device)
//   (/root/reference/scripts/finetune_inductive_learning.py:291-293, 45): concatenate the chosen
// graphs' node features, offset their edge indices, build the node -> graph vector — on the
// device, in two launches, with no host <-> device copy of graph data per step. Same result as
// the CPU collation, bit for bit (pure data movement and integer offsets).
//
// Dataset layout (built once by gconv_adapter_amd.data.MoleculeStore): graphs stored back to
// back; node_ptr / edge_ptr [G + 1] are their node / edge offsets; edge_src / edge_dst hold
// LOCAL node ids, each graph's edges contiguous, in the graph's own edge order.
#include "common.h"

namespace gconv {

// exclusive scans of the chosen graphs' node and edge counts: offs[0][B + 1], offs[1][B + 1]
__global__ __launch_bounds__(1024) void collate_offsets_kernel(const int32_t* __restrict__ node_ptr,
                                                               const int32_t* __restrict__ edge_ptr,
                                                               const int32_t* __restrict__ ids, int nb,
                                                               int32_t* __restrict__ noff,
                                                               int32_t* __restrict__ eoff) {
  __shared__ int sn[1024], se[1024];
  __shared__ int carry_n, carry_e;
  if (threadIdx.x == 0) { carry_n = 0; carry_e = 0; }
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    const int p = base + threadIdx.x;
    int cn = 0, ce = 0;
    if (p < nb) {
      const int g = ids[p];
      cn = node_ptr[g + 1] - node_ptr[g];
      ce = edge_ptr[g + 1] - edge_ptr[g];
    }
    sn[threadIdx.x] = cn; se[threadIdx.x] = ce;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
      int vn = 0, ve = 0;
      if (static_cast<int>(threadIdx.x) >= off) { vn = sn[threadIdx.x - off]; ve = se[threadIdx.x - off]; }
      __syncthreads();
      sn[threadIdx.x] += vn; se[threadIdx.x] += ve;
      __syncthreads();
    }
    if (p < nb) {
      noff[p] = carry_n + sn[threadIdx.x] - cn;
      eoff[p] = carry_e + se[threadIdx.x] - ce;
    }
    __syncthreads();
    if (threadIdx.x == 1023) { carry_n += sn[1023]; carry_e += se[1023]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { noff[nb] = carry_n; eoff[nb] = carry_e; }
}

// one workgroup per chosen graph: feature rows (16-byte copies), edges (offset + widen), vectors
__global__ __launch_bounds__(256) void collate_copy_kernel(
    const float* __restrict__ x_all, int d, const int32_t* __restrict__ node_ptr,
    const int32_t* __restrict__ edge_ptr, const int32_t* __restrict__ edge_src,
    const int32_t* __restrict__ edge_dst, const int32_t* __restrict__ ids,
    const int32_t* __restrict__ noff, const int32_t* __restrict__ eoff, long long e_total,
    float* __restrict__ x_out, int64_t* __restrict__ ei_out, int32_t* __restrict__ graph_ptr_out,
    int32_t* __restrict__ batch_out, int nb) {
  const int p = blockIdx.x;
  const int g = ids[p];
  const int n0 = node_ptr[g], nn = node_ptr[g + 1] - n0, no = noff[p];
  const int e0 = edge_ptr[g], ne = edge_ptr[g + 1] - e0, eo = eoff[p];
  const int d4 = d >> 2;
  const f32x4* src = reinterpret_cast<const f32x4*>(x_all + static_cast<size_t>(n0) * d);
  f32x4* dst = reinterpret_cast<f32x4*>(x_out + static_cast<size_t>(no) * d);
  for (int i = threadIdx.x; i < nn * d4; i += 256) dst[i] = src[i];
  for (int k = threadIdx.x; k < ne; k += 256) {
    ei_out[eo + k] = static_cast<int64_t>(edge_src[e0 + k]) + no;
    ei_out[e_total + eo + k] = static_cast<int64_t>(edge_dst[e0 + k]) + no;
  }
  if (batch_out)
    for (int i = threadIdx.x; i < nn; i += 256) batch_out[no + i] = p;
  if (threadIdx.x == 0) {
    graph_ptr_out[p] = no;
    if (p == nb - 1) graph_ptr_out[nb] = no + nn;
  }
}

}  // namespace gconv

using namespace gconv;

extern "C" size_t gconv_collate_workspace_bytes(int32_t num_graphs) {
  return num_graphs < 0 ? 0 : align_up(2 * (static_cast<size_t>(num_graphs) + 1) * sizeof(int32_t), 256);
}

extern "C" gconv_status_t gconv_collate_f32(const float* x_all, int32_t hidden, const int32_t* node_ptr,
                                            const int32_t* edge_ptr, const int32_t* edge_src,
                                            const int32_t* edge_dst, const int32_t* graph_ids,
                                            int32_t num_graphs, int64_t num_edges_out, float* x_out,
                                            int64_t* edge_index_out, int32_t* graph_ptr_out,
                                            int32_t* batch_out, void* workspace, size_t workspace_bytes,
                                            gconv_stream_t stream_) {
  if (num_graphs < 0 || hidden <= 0 || num_edges_out < 0) return GCONV_E_INVALID_ARG;
  if (hidden % 4 != 0) return GCONV_E_UNSUPPORTED;
  if (!graph_ptr_out || !workspace) return GCONV_E_INVALID_ARG;
  if (workspace_bytes < gconv_collate_workspace_bytes(num_graphs)) return GCONV_E_WORKSPACE;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (num_graphs == 0) {
    GCONV_HIP(hipMemsetAsync(graph_ptr_out, 0, sizeof(int32_t), stream));
    return GCONV_OK;
  }
  if (!x_all || !node_ptr || !edge_ptr || !graph_ids || !x_out || (num_edges_out > 0 && (!edge_src || !edge_dst || !edge_index_out)))
    return GCONV_E_INVALID_ARG;
  if (!aligned16(x_all) || !aligned16(x_out)) return GCONV_E_ALIGNMENT;
  int32_t* noff = static_cast<int32_t*>(workspace);
  int32_t* eoff = noff + num_graphs + 1;
  collate_offsets_kernel<<<1, 1024, 0, stream>>>(node_ptr, edge_ptr, graph_ids, num_graphs, noff, eoff);
  GCONV_LAUNCH_CHECK();
  collate_copy_kernel<<<num_graphs, 256, 0, stream>>>(x_all, hidden, node_ptr, edge_ptr, edge_src, edge_dst,
                                                      graph_ids, noff, eoff, num_edges_out, x_out, edge_index_out,
                                                      graph_ptr_out, batch_out, num_graphs);
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}
