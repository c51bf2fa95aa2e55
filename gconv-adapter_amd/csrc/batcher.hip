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
#include <hipcub/hipcub.hpp>

#include "common.h"
#include "tiles.h"

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

// ---- graph plan of a collated batch, assembled from the dataset's plan (no sort, no sync) ----
// one workgroup per chosen graph: CSR rows re-based, columns offset, dis copied
__global__ __launch_bounds__(256) void plan_copy_kernel(
    const int32_t* __restrict__ rowptr_all, const int32_t* __restrict__ col_all,
    const float* __restrict__ dis_all, const int32_t* __restrict__ node_ptr,
    const int32_t* __restrict__ ids, const int32_t* __restrict__ noff, const int32_t* __restrict__ eoff,
    int32_t* __restrict__ rowptr_b, int32_t* __restrict__ col_b, float* __restrict__ dis_b, int nb) {
  const int p = blockIdx.x;
  const int g = ids[p];
  const int n0 = node_ptr[g], nn = node_ptr[g + 1] - n0, no = noff[p], eo = eoff[p];
  const int r0 = rowptr_all[n0], ne = rowptr_all[n0 + nn] - r0;
  for (int i = threadIdx.x; i < nn; i += 256) {
    rowptr_b[no + i] = rowptr_all[n0 + i] - r0 + eo;
    if (dis_b) dis_b[no + i] = dis_all[n0 + i];
  }
  for (int k = threadIdx.x; k < ne; k += 256) col_b[eo + k] = col_all[r0 + k] - n0 + no;
  if (p == nb - 1 && threadIdx.x == 0) rowptr_b[no + nn] = eo + ne;
}

// tile t starts at the last graph boundary at or below t * T (graphs are the components: no edge
// crosses a graph boundary, so these tiles carry no halo and the bounds of the LDS budget are
// known without reading anything back — this call never synchronises)
__global__ void plan_tiles_kernel(const int32_t* __restrict__ noff, int nb, int n, int ntiles, int T,
                                  int32_t* __restrict__ tile_ptr) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > ntiles) return;
  if (t == ntiles) { tile_ptr[t] = n; return; }
  const int q = t * T;
  int lo = 0, hi = nb;                       // largest p with noff[p] <= q  (noff[0] = 0)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (noff[mid] <= q) lo = mid; else hi = mid - 1;
  }
  tile_ptr[t] = noff[lo];
}

struct StoreLayout {
  size_t rowptr, col, dis, tile_ptr, recs, tiles, lcol, halo, key, key_out, id, id_out, info, noff, cub, total, cub_bytes;
  int ntiles, T;
};
static StoreLayout store_layout(int32_t n, int64_t nnz, int32_t nb) {
  StoreLayout L{};
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes + 16, 256); return o; };
  L.T = tile_nominal(n);
  L.ntiles = ceil_div(n > 0 ? n : 1, L.T);
  const size_t nt = static_cast<size_t>(L.ntiles), e1 = static_cast<size_t>(nnz > 0 ? nnz : 1);
  L.rowptr = take((static_cast<size_t>(n) + 1) * 4);
  L.col = take(e1 * 4);
  L.dis = take((static_cast<size_t>(n) + 1) * 4);
  L.tile_ptr = take((nt + 1) * 4);
  L.recs = take(nt * sizeof(gconv_tile_t));
  L.tiles = take(nt * sizeof(gconv_tile_t));
  L.lcol = take(e1 * 4);
  L.halo = take(nt * kHaloSlot * 4);
  L.key = take(nt * 4);
  L.key_out = take(nt * 4);
  L.id = take(nt * 4);
  L.id_out = take(nt * 4);
  L.info = take(64 * 4);
  L.noff = take(2 * (static_cast<size_t>(nb) + 1) * 4);
  size_t cb = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, cb, static_cast<const uint32_t*>(nullptr),
                                     static_cast<uint32_t*>(nullptr), static_cast<const uint32_t*>(nullptr),
                                     static_cast<uint32_t*>(nullptr), L.ntiles, 0, kTileSizeBits, nullptr);
  L.cub_bytes = cb;
  L.cub = take(cb + 256);
  L.total = off;
  return L;
}

}  // namespace gconv

using namespace gconv;

extern "C" size_t gconv_graph_from_store_bytes(int32_t num_nodes, int64_t nnz, int32_t num_graphs) {
  if (num_nodes < 0 || nnz < 0 || num_graphs < 0) return 0;
  return store_layout(num_nodes, nnz, num_graphs).total;
}

extern "C" gconv_status_t gconv_graph_from_store(const gconv_graph_t* dataset, const int32_t* node_ptr,
                                                 const int32_t* nnz_ptr, const int32_t* graph_ids,
                                                 int32_t num_graphs, int32_t num_nodes, int64_t nnz,
                                                 int32_t max_graph_nodes, void* arena_, size_t arena_bytes,
                                                 gconv_graph_t* g, gconv_stream_t stream_) {
  if (!dataset || !g || !arena_ || num_graphs <= 0 || num_nodes <= 0 || nnz < 0) return GCONV_E_INVALID_ARG;
  if (!node_ptr || !nnz_ptr || !graph_ids) return GCONV_E_INVALID_ARG;
  // the shortcut needs what molecule datasets have: a symmetric edge list and no long rows
  if (!dataset->symmetric || dataset->fwd.n_long != 0) return GCONV_E_UNSUPPORTED;
  if (!aligned16(arena_)) return GCONV_E_ALIGNMENT;
  const StoreLayout L = store_layout(num_nodes, nnz, num_graphs);
  if (arena_bytes < L.total) return GCONV_E_WORKSPACE;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  char* arena = static_cast<char*>(arena_);
  int32_t* rowptr = reinterpret_cast<int32_t*>(arena + L.rowptr);
  int32_t* col = reinterpret_cast<int32_t*>(arena + L.col);
  float* dis = dataset->normalize ? reinterpret_cast<float*>(arena + L.dis) : nullptr;
  int32_t* tile_ptr = reinterpret_cast<int32_t*>(arena + L.tile_ptr);
  int32_t* noff = reinterpret_cast<int32_t*>(arena + L.noff);
  int32_t* eoff = noff + num_graphs + 1;
  collate_offsets_kernel<<<1, 1024, 0, stream>>>(node_ptr, nnz_ptr, graph_ids, num_graphs, noff, eoff);
  GCONV_LAUNCH_CHECK();
  plan_copy_kernel<<<num_graphs, 256, 0, stream>>>(dataset->fwd.rowptr, dataset->fwd.col, dataset->dis, node_ptr,
                                                   graph_ids, noff, eoff, rowptr, col, dis, num_graphs);
  GCONV_LAUNCH_CHECK();
  const int T = 256;
  plan_tiles_kernel<<<(L.ntiles + 1 + T - 1) / T, T, 0, stream>>>(noff, num_graphs, num_nodes, L.ntiles, L.T, tile_ptr);
  GCONV_LAUNCH_CHECK();
  gconv_tile_t* recs = reinterpret_cast<gconv_tile_t*>(arena + L.recs);
  gconv_tile_t* tiles = reinterpret_cast<gconv_tile_t*>(arena + L.tiles);
  int32_t* lcol = reinterpret_cast<int32_t*>(arena + L.lcol);
  int32_t* halo = reinterpret_cast<int32_t*>(arena + L.halo);
  int32_t* info = reinterpret_cast<int32_t*>(arena + L.info);
  GCONV_HIP(hipMemsetAsync(info, 0, 64 * sizeof(int32_t), stream));
  // records and tile-local columns on the given cuts (graph boundaries: every halo is empty)
  tile_build_kernel<false><<<(L.ntiles + 3) / 4, 256, 0, stream>>>(rowptr, col, dis, num_nodes, L.ntiles, L.T, 0, tile_ptr,
                                                                   lcol, halo, recs, info, TileInfoSlots{0, 1, 2});
  GCONV_LAUNCH_CHECK();
  if (L.ntiles <= kNumCU) {   // one tile per persistent workgroup: node order, no sort (see gconv_graph_build)
    tiles = recs;
  } else if (L.ntiles <= kSmallSortMax) {
    tile_sort_small(recs, nullptr, L.ntiles, tiles, nullptr, stream);
    GCONV_LAUNCH_CHECK();
  } else {
    uint32_t* key = reinterpret_cast<uint32_t*>(arena + L.key);
    uint32_t* key_out = reinterpret_cast<uint32_t*>(arena + L.key_out);
    uint32_t* id = reinterpret_cast<uint32_t*>(arena + L.id);
    uint32_t* id_out = reinterpret_cast<uint32_t*>(arena + L.id_out);
    tile_sort_keys_kernel<<<(L.ntiles + T - 1) / T, T, 0, stream>>>(recs, L.ntiles, key, id);
    GCONV_LAUNCH_CHECK();
    size_t cub_bytes = L.cub_bytes;
    GCONV_HIP(hipcub::DeviceRadixSort::SortPairs(arena + L.cub, cub_bytes, key, key_out, id, id_out, L.ntiles, 0,
                                                 kTileSizeBits, stream));
    tile_sort_gather_kernel<<<(L.ntiles + T - 1) / T, T, 0, stream>>>(recs, nullptr, id_out, L.ntiles, tiles, nullptr);
    GCONV_LAUNCH_CHECK();
  }
  *g = gconv_graph_t{};
  g->fwd.num_nodes = num_nodes; g->fwd.nnz = static_cast<int32_t>(nnz);
  g->fwd.rowptr = rowptr; g->fwd.col = col;
  g->fwd.near_edges = static_cast<int32_t>(nnz);   // block-diagonal by construction: every neighbour sits in its row's molecule
  g->bwd = g->fwd;
  g->dis = dis;
  g->tile_ptr = tile_ptr;
  g->tiles_fwd = gconv_tiling_t{tiles, lcol, halo};
  g->tiles_bwd = g->tiles_fwd;
  g->n_tiles = L.ntiles;
  // upper bounds known without reading anything back: a tile starts at most T - 1 nodes before
  // its nominal start and ends inside the graph that covers its nominal end (they only size the
  // LDS of the fused kernels)
  const int bound = L.T + max_graph_nodes;
  g->tile_nodes_max = bound < num_nodes ? bound : num_nodes;
  const int64_t eb = static_cast<int64_t>(g->tile_nodes_max) * (dataset->max_degree > 0 ? dataset->max_degree : 1);
  g->tile_edges_max = static_cast<int32_t>(eb < nnz ? eb : nnz);
  g->symmetric = 1; g->normalize = dataset->normalize; g->max_degree = dataset->max_degree; g->bad_index = 0;
  return GCONV_OK;
}

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
