// Tiles of the fused adapter kernels (fused_tile.hip): how a graph plan cuts the node range into
// pieces that ONE workgroup each runs end to end, and what a piece needs besides its own rows.
//
// A tile owns a contiguous node range [first, first + count). Cuts are placed near the multiples
// of the nominal tile size T (tile_nominal: about N / 256 nodes, so that each of the 256
// persistent workgroups gets ONE tile of about the same size), preferably on a boundary that no
// edge crosses — collated molecule batches have one between any two molecules — but a cut may go
// THROUGH a component: the two-hop neighbourhood the adapter needs then comes along as a halo.
//   phase 1 (down projection)  runs on own rows + halo1 + halo2   (n0 rows)
//   phase 2 (first propagate)  runs on own rows + halo1           (n1 rows)
//   phase 3 / 4                run on the own rows                (count rows)
// halo1 = nodes outside the range that an own row gathers from, halo2 = nodes outside both that a
// halo1 row gathers from. Halo rows are recomputed by every tile that needs them (a row's value
// is a fixed-order sum: every copy has the same bits) and never stored, so tiles exchange
// nothing and each output row has exactly one writer. Round 2 cut only at edge-free boundaries:
// a batch's largest molecule (129 of 13104 nodes in the 512-molecule HIV shard) was one tile on
// one CU and set the kernel's length.
//
// Per tile the plan stores a record (gconv_tile_t), the own rows' column ids translated to tile-
// local row numbers (`lcol`, indexed like the CSR's `col`; local id = node - first for own nodes,
// count + k for the k-th halo node) and a fixed-size halo slot: halo node ids (halo1 first, in
// order of first appearance), the local rowptr of the halo1 rows and their local column ids.
// A tile whose halo outgrows the slot (64 nodes, 256 edges) makes the graph "not tileable"
// (n_tiles = 0: the general path runs) — molecules stay far below, expanders exceed it at once.
#pragma once

#include <climits>

#include "common.h"

namespace gconv {

constexpr int kHaloNodes = GCONV_TILE_HALO_NODES;   // 64: one lane per halo node
constexpr int kHaloEdges = GCONV_TILE_HALO_EDGES;   // 256
constexpr int kHaloRpOff = kHaloNodes;                       // [kHaloNodes + 1] local rowptr of the halo1 rows
constexpr int kHaloColOff = 2 * kHaloNodes + 4;              // [kHaloEdges] local column ids of the halo1 rows
constexpr int kHaloDisOff = kHaloColOff + kHaloEdges;        // [kHaloNodes] dis of the halo nodes (float)
constexpr int kHaloSlot = GCONV_TILE_HALO_SLOT;              // 4-byte words per tile
static_assert(kHaloDisOff + kHaloNodes <= kHaloSlot && kHaloColOff % 4 == 0 && kHaloSlot % 4 == 0, "halo slot layout");
static_assert(kHaloNodes == 64, "tile_halo_wave keeps one halo node per lane");

// counters of a build that the tile kernels report into (indices into the build's info words)
struct TileInfoSlots { int notile, nodes_max, edges_max; };

// How the tile routines read the CSR: from global memory (the multi-launch build, the plan assembled
// from a resident dataset) or — the single-workgroup build — from the LDS copy the workgroup has just
// produced (row offsets as they are, columns as the low field of the sorted keys).
struct CsrGlobal {
  const int32_t* __restrict__ rowptr_;
  const int32_t* __restrict__ col_;
  __device__ __forceinline__ int rowptr(int i) const { return rowptr_[i]; }
  __device__ __forceinline__ int col(int e) const { return col_[e]; }
};

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
  return v;
}

// Cut k of `ntiles` (0 < k < ntiles) at nominal position q0 = k * T: the candidate closest to q0
// in (q0 - slack, q0] that no edge of the rows within reach crosses, q0 itself if there is none.
// "Within reach" = 128 rows below and 64 rows above q0, through each row's smallest and largest
// column (rows are sorted): exact for the tree-plus-chords layouts of molecules, a heuristic in
// general — correctness never depends on it (tile_halo_wave finds every crossing edge).
// One wave; all lanes return the cut.
// The loads of one cut, apart from the arithmetic: a wave that places two cuts (its tile's lower and
// upper one) issues both sets before it waits for either — two memory round trips, not four.
struct CutReach { int mxA, mxB, mnB, mnC, rpB, rpC; };   // rp: rowptr of the lane's row of B / C

template <class Csr>
__device__ __forceinline__ CutReach tile_cut_reach(const Csr& csr, int n, int k, int ntiles, int T, int lane) {
  // branch-free (clamped row, clamped edge; E = 0 builds still own one readable column word), so that
  // the compiler keeps the six row reads of two cuts in one batch
  const int q0 = (k <= 0 || k >= ntiles) ? 0 : k * T;
  auto reach = [&](int row, int& mx, int& mn, int& a) {
    const bool in = row >= 0 && row < n;
    const int rc = in ? row : 0;
    a = csr.rowptr(rc);
    const int b = csr.rowptr(rc + 1);
    const bool any = in && b > a;
    const int lo = csr.col(any ? a : 0), hi = csr.col(any ? b - 1 : 0);
    mn = any ? lo : INT_MAX;
    mx = any ? hi : -1;
  };
  CutReach r;
  int mnA, mxC, rpA;
  reach(q0 - 128 + lane, r.mxA, mnA, rpA);
  reach(q0 - 64 + lane, r.mxB, r.mnB, r.rpB);    // lane b <-> row q0 - 64 + b
  reach(q0 + lane, mxC, r.mnC, r.rpC);
  (void)mnA; (void)mxC; (void)rpA;
  return r;
}

__device__ __forceinline__ int tile_cut_pick(const CutReach& r, int n, int k, int ntiles, int T, int slack, int lane) {
  if (k <= 0) return 0;
  if (k >= ntiles) return n;
  const int q0 = k * T;
  const int maxA = wave_max_i(r.mxA), minC = wave_min_i(r.mnC);
  int pm = r.mxB;                      // inclusive prefix maximum over the lanes of B
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(pm, off);
    if (lane >= off) pm = max(pm, v);
  }
  int sm = r.mnB;                      // inclusive suffix minimum over the lanes of B
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_down(sm, off);
    if (lane + off < 64) sm = min(sm, v);
  }
  int sme = __shfl_down(sm, 1);        // exclusive: rows above this lane's
  if (lane == 63) sme = INT_MAX;
  // lane j <-> candidate q = q0 - 63 + j: rows below q are A and the lanes <= j of B
  const int q = q0 - 63 + lane;
  const bool crossed = max(maxA, pm) >= q || min(sme, minC) < q;
  const bool cand = q >= 1 && q0 - q < slack;
  const unsigned long long m = __ballot(cand && !crossed);
  if (m == 0) return q0;
  return q0 - 63 + (63 - __clzll(static_cast<long long>(m)));
}

// rowptr of an inner cut q of cut k (0 < k < ntiles), from the reach's loads: q is in [q0 - 63, q0]
__device__ __forceinline__ int tile_cut_rowptr(const CutReach& r, int k, int T, int q) {
  const int q0 = k * T;
  return q == q0 ? __builtin_amdgcn_readlane(r.rpC, 0) : __builtin_amdgcn_readlane(r.rpB, q - q0 + 64);
}

// both cuts of tile t: rows [ts, te), edges [e0, e1) — the cuts' rowptr values come out of the same loads
template <class Csr>
__device__ __forceinline__ void tile_cuts_wave(const Csr& csr, int n, int t, int ntiles, int T, int slack, int lane,
                                               int& ts, int& te, int& e0, int& e1) {
  const CutReach lo = tile_cut_reach(csr, n, t, ntiles, T, lane);
  const CutReach hi = tile_cut_reach(csr, n, t + 1, ntiles, T, lane);
  const int nnz = csr.rowptr(n);
  ts = tile_cut_pick(lo, n, t, ntiles, T, slack, lane);
  te = tile_cut_pick(hi, n, t + 1, ntiles, T, slack, lane);
  e0 = t <= 0 ? csr.rowptr(0) : tile_cut_rowptr(lo, t, T, ts);
  e1 = t + 1 >= ntiles ? nnz : tile_cut_rowptr(hi, t + 1, T, te);
}

// Halo, local column ids and record of tile t = rows [ts, te) of one CSR (forward or transposed).
// One wave. `rec` receives the unsorted record (slot = t).
//
// The halo is collected in two sweeps: the own rows' edges, 64 per step (halo1: outside columns in
// edge order), then the halo1 rows' edges with ONE LANE PER HALO ROW, all lanes stepping through
// their rows together (halo2: outside columns in (edge position, halo row) order) — a handful of
// memory round trips per tile whatever the halo's size (a row-by-row walk of the halo cost two
// dependent round trips per halo node: 25 us per tile, most of the graph build). A tile that cannot
// be a tile — more own edges than kTileMaxOwnEdges, a halo beyond its slot — is given up at once
// (hub rows of a power-law graph would otherwise be walked edge by edge), and once one tile has
// been given up the graph is not tileable and the other waves stop as well.
constexpr int kTileMaxOwnEdges = 1 << 14;
constexpr int kOwnRegs = 4, kHaloRegs = 6;   // edges of a tile kept in registers between the two sweeps

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

template <class Csr, bool COHERENT_DIS = false>
__device__ __forceinline__ void tile_halo_wave(const Csr& csr, const float* __restrict__ dis, int ts, int te, int e0,
                                               int e1, int t, int32_t* __restrict__ lcol, int32_t* __restrict__ slots,
                                               gconv_tile_t* __restrict__ rec, int32_t* info,
                                               TileInfoSlots is, int lane) {
  const int nt = te - ts;
  int32_t* hnodes = slots + static_cast<size_t>(t) * kHaloSlot;
  int32_t* hrp = hnodes + kHaloRpOff;
  int32_t* hcol = hnodes + kHaloColOff;
  auto give_up = [&]() {
    if (lane == 0) {
      atomicOr(&info[is.notile], 1);
      *rec = gconv_tile_t{ts, nt, nt, nt, t, e0, e1 - e0, 0};
    }
  };
  if (__hip_atomic_load(&info[is.notile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
      e1 - e0 > kTileMaxOwnEdges) {
    give_up();
    return;
  }
  int hval = -1;       // lane j: the j-th halo node
  int hcount = 0;      // wave-uniform
  bool overflow = false;
  // appends the nodes c of the lanes in m that are not in the list yet, in lane order
  auto append = [&](unsigned long long m, int c) {
    while (m && !overflow) {
      const int b = __ffsll(static_cast<long long>(m)) - 1;   // wave-uniform: v_readlane, not a permute through LDS
      m &= m - 1;
      const int v = __builtin_amdgcn_readlane(c, b);
      if (__ballot(hval == v) == 0) {
        if (hcount < kHaloNodes) {
          if (lane == hcount) hval = v;
          ++hcount;
        } else {
          overflow = true;
        }
      }
    }
  };
  // the first kOwnRegs x 64 own edges (all of a molecule tile's) are read ONCE, in one batch of loads
  int cown[kOwnRegs];
#pragma unroll
  for (int j = 0; j < kOwnRegs; ++j) {
    const int e = e0 + 64 * j + lane;
    cown[j] = e < e1 ? csr.col(e) : ts;
  }
#pragma unroll
  for (int j = 0; j < kOwnRegs; ++j)                          // halo1: outside columns of the own rows
    if (e0 + 64 * j < e1 && !overflow) append(__ballot(cown[j] < ts || cown[j] >= te), cown[j]);
  for (int base = e0 + 64 * kOwnRegs; base < e1 && !overflow; base += 64) {
    const int e = base + lane;
    const int c = e < e1 ? csr.col(e) : ts;
    append(__ballot(e < e1 && (c < ts || c >= te)), c);
  }
  if (overflow) { give_up(); return; }
  const int nh1 = hcount;
  // the halo1 rows, one per lane: offsets, degrees, exclusive prefix of the degrees
  const int ha = lane < nh1 ? csr.rowptr(hval) : 0;
  const int hdeg = lane < nh1 ? csr.rowptr(hval + 1) - ha : 0;
  const int he = wave_sum_i(hdeg);
  const int dmax = wave_max_i(hdeg);
  if (he > kHaloEdges) { give_up(); return; }
  int hoff = hdeg;                                            // inclusive scan over the lanes
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(hoff, off);
    if (lane >= off) hoff += v;
  }
  hoff -= hdeg;                                               // exclusive: first edge slot of this lane's row
  int chalo[kHaloRegs];                                       // the halo1 rows' first columns, one batch
#pragma unroll
  for (int k = 0; k < kHaloRegs; ++k) chalo[k] = k < hdeg ? csr.col(ha + k) : ts;
#pragma unroll
  for (int k = 0; k < kHaloRegs; ++k)                         // halo2: outside columns of the halo1 rows
    if (k < dmax && !overflow) append(__ballot(chalo[k] < ts || chalo[k] >= te), chalo[k]);
  for (int k = kHaloRegs; k < dmax && !overflow; ++k) {
    const bool ok = k < hdeg;
    const int c = ok ? csr.col(ha + k) : ts;
    append(__ballot(ok && (c < ts || c >= te)), c);
  }
  if (overflow) { give_up(); return; }
  // tile-local id of column c: own nodes by subtraction, halo nodes by their position in the list —
  // one step per OUTSIDE lane (a few per 64 edges), not per list entry (uniform control flow)
  auto local_id = [&](int c, bool outside) {
    int id = c - ts;
    unsigned long long m = __ballot(outside);
    while (m) {
      const int b = __ffsll(static_cast<long long>(m)) - 1;
      m &= m - 1;
      const int v = __builtin_amdgcn_readlane(c, b);
      const int j = __ffsll(static_cast<long long>(__ballot(hval == v))) - 1;   // v is in the list
      if (lane == b) id = nt + j;
    }
    return id;
  };
#pragma unroll
  for (int j = 0; j < kOwnRegs; ++j) {
    const int e = e0 + 64 * j + lane;
    if (e0 + 64 * j < e1) {                                   // wave-uniform
      const int id = local_id(cown[j], cown[j] < ts || cown[j] >= te);
      if (e < e1) lcol[e] = id;
    }
  }
  for (int base = e0 + 64 * kOwnRegs; base < e1; base += 64) {
    const int e = base + lane;
    const bool ok = e < e1;
    const int c = ok ? csr.col(e) : ts;
    const int id = local_id(c, ok && (c < ts || c >= te));
    if (ok) lcol[e] = id;
  }
#pragma unroll
  for (int k = 0; k < kHaloRegs; ++k) {
    if (k < dmax) {                                           // wave-uniform
      const int id = local_id(chalo[k], chalo[k] < ts || chalo[k] >= te);
      if (k < hdeg) hcol[hoff + k] = id;
    }
  }
  for (int k = kHaloRegs; k < dmax; ++k) {
    const bool ok = k < hdeg;
    const int c = ok ? csr.col(ha + k) : ts;
    const int id = local_id(c, ok && (c < ts || c >= te));
    if (ok) hcol[hoff + k] = id;
  }
  if (lane == 0) hrp[0] = 0;
  if (lane < nh1) hrp[lane + 1] = hoff + hdeg;
  if (lane < hcount) {
    hnodes[lane] = hval;
    // (COHERENT_DIS: dis was written earlier in this launch by another workgroup — see graph_prep.hip)
    if (dis)
      hnodes[kHaloDisOff + lane] =
          COHERENT_DIS ? __hip_atomic_load(reinterpret_cast<const int32_t*>(dis) + hval, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT)
                       : __float_as_int(dis[hval]);
  }
  if (lane == 0) {
    gconv_tile_t r;
    r.first = ts; r.count = nt; r.n1 = nt + nh1; r.n0 = nt + hcount; r.slot = t; r.e0 = e0;
    r.ne_own = e1 - e0; r.ne_halo = he;
    *rec = r;
    atomicMax(&info[is.nodes_max], nt + hcount);
    atomicMax(&info[is.edges_max], e1 - e0 + he);
  }
}

// One wave per tile. CUTS: the wave places its own two cuts (and publishes the lower one in
// tile_ptr); otherwise the cuts are read from tile_ptr (the transposed CSR of a directed edge list
// is tiled on the cuts chosen for the forward one).
template <bool CUTS>
__global__ __launch_bounds__(256) void tile_build_kernel(const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ col,
                                                         const float* __restrict__ dis, int n, int ntiles,
                                                         int T, int slack, int32_t* __restrict__ tile_ptr,
                                                         int32_t* __restrict__ lcol, int32_t* __restrict__ slots,
                                                         gconv_tile_t* __restrict__ recs, int32_t* info,
                                                         TileInfoSlots is) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= ntiles) return;   // wave-uniform
  int ts, te, e0, e1;
  const CsrGlobal csr{rowptr, col};
  if (CUTS) {
    tile_cuts_wave(csr, n, t, ntiles, T, slack, lane, ts, te, e0, e1);
    if (lane == 0) {
      tile_ptr[t] = ts;
      if (t == ntiles - 1) tile_ptr[ntiles] = n;
    }
  } else {
    ts = tile_ptr[t];
    te = tile_ptr[t + 1];
    e0 = rowptr[ts];
    e1 = rowptr[te];
  }
  tile_halo_wave(csr, dis, ts, te, e0, e1, t, lcol, slots, recs + t, info, is, lane);
}

// Size-sorted tile list for small tile counts (a molecule batch has a few hundred tiles): ONE
// workgroup ranks every tile by (rows incl. halo descending, tile index ascending) — the order a
// stable sort gives — and writes the records of both directions in that order. The persistent
// workgroups deal themselves tiles from this list boustrophedon, so no workgroup gets two big ones.
// Four waves, PER tiles per thread, one unique key per tile ((size descending) : index) in LDS: a
// tile's rank is the number of smaller keys. (Round 5: this was one 1024-thread workgroup of 40
// VGPRs — 160 per SIMD, which does not fit beside a backward tile workgroup (2 x 208 of 512): on
// the side stream it waited for a CU to fall free at a kernel boundary and the next tile kernel's
// workgroup for that CU then waited for IT. One wave per SIMD under 96 VGPRs runs beside.)
constexpr int kSmallSortMax = 2048, kSmallSortThreads = 256;
template <int PER>
static __global__ __launch_bounds__(kSmallSortThreads) void tile_sort_small_kernel(const gconv_tile_t* __restrict__ recs_f,
                                                                                   const gconv_tile_t* __restrict__ recs_b,
                                                                                   int ntiles, gconv_tile_t* __restrict__ out_f,
                                                                                   gconv_tile_t* __restrict__ out_b) {
  static_assert(PER * kSmallSortThreads <= kSmallSortMax, "a key per tile in LDS");
  __shared__ uint32_t key[kSmallSortMax];
  const int tid = threadIdx.x;
  uint32_t mine[PER];
  int rank[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int t = tid + j * kSmallSortThreads;
    // (11 bits of index below 2048 tiles; a tile has fewer than 2^20 rows)
    mine[j] = t < ntiles ? (static_cast<uint32_t>(0xFFFFF - min(recs_f[t].n0, 0xFFFFF)) << 11) | static_cast<uint32_t>(t) : 0xFFFFFFFFu;
    if (t < ntiles) key[t] = mine[j];
    rank[j] = 0;
  }
  __syncthreads();
  for (int u = 0; u < ntiles; ++u) {
    const uint32_t ku = key[u];
#pragma unroll
    for (int j = 0; j < PER; ++j) rank[j] += ku < mine[j] ? 1 : 0;
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int t = tid + j * kSmallSortThreads;
    if (t < ntiles) {
      out_f[rank[j]] = recs_f[t];
      if (recs_b) out_b[rank[j]] = recs_b[t];
    }
  }
}
// (queued on `stream`; ntiles <= kSmallSortMax)
static inline void tile_sort_small(const gconv_tile_t* recs_f, const gconv_tile_t* recs_b, int ntiles, gconv_tile_t* out_f,
                                   gconv_tile_t* out_b, hipStream_t stream) {
  if (ntiles <= 2 * kSmallSortThreads) tile_sort_small_kernel<2><<<1, kSmallSortThreads, 0, stream>>>(recs_f, recs_b, ntiles, out_f, out_b);
  else if (ntiles <= 4 * kSmallSortThreads) tile_sort_small_kernel<4><<<1, kSmallSortThreads, 0, stream>>>(recs_f, recs_b, ntiles, out_f, out_b);
  else tile_sort_small_kernel<8><<<1, kSmallSortThreads, 0, stream>>>(recs_f, recs_b, ntiles, out_f, out_b);
}

// large tile counts: stable radix sort of (key, tile id) pairs, then this gather
constexpr int kTileSizeMax = 1023, kTileSizeBits = 10;
static __global__ void tile_sort_keys_kernel(const gconv_tile_t* __restrict__ recs_f, int ntiles,
                                             uint32_t* __restrict__ key, uint32_t* __restrict__ id) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntiles) return;
  key[t] = static_cast<uint32_t>(kTileSizeMax - min(recs_f[t].n0, kTileSizeMax));
  id[t] = static_cast<uint32_t>(t);
}
static __global__ void tile_sort_gather_kernel(const gconv_tile_t* __restrict__ recs_f,
                                               const gconv_tile_t* __restrict__ recs_b,
                                               const uint32_t* __restrict__ id_sorted, int ntiles,
                                               gconv_tile_t* __restrict__ out_f, gconv_tile_t* __restrict__ out_b) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ntiles) return;
  const uint32_t t = id_sorted[p];
  out_f[p] = recs_f[t];
  if (recs_b) out_b[p] = recs_b[t];
}

}  // namespace gconv
