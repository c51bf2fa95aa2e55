// K0 — graph build: COO edge list -> everything the kernels need, in one call.
//
// Replaces what PyG's gcn_norm recomputes on every GCNConv call
// (add_remaining_self_loops, degree scatter, deg^-1/2; reached from
// /root/reference/src/finetune/gconv_adapter.py:92) with a one-off build per
// batch that all adapters of a model forward share.
//
// Pipeline (all on `stream`, one host sync for symmetric edge lists):
//   1. make_keys   : key = (row << b) | col, dropped / invalid edges -> row = N
//   2. radix sort  : hipCUB, 32-bit keys when 2 bits(N) <= 32 (molecule batches), else 64-bit
//   3. build_csr   : rowptr from key run boundaries, col = low key field
//   4. row_stats   : dis = (deg + 1)^-1/2, long-row / segment lists, max degree
//   5. sym_check   : exact multiset test S == S^T by ranked binary search in the CSR
//   6. tiles       : one wave per tile places its cuts, finds its halo, translates its columns
//                    (tiles.h; round 2: span-diff atomics + scan + edge-free boundary search)
//   7. D2H of the counters, sync; S^T (second sort + its tiles) only when the test failed;
//      size sort of the tile records
#include <hipcub/hipcub.hpp>

#include <atomic>
#include <cstdlib>
#include <mutex>

#include "common.h"
#include "tiles.h"

namespace gconv {

int tile_nominal(int n) {
  static const int forced = [] {
    const char* v = std::getenv("GCONV_TILE_NODES_OVERRIDE");
    return v ? std::atoi(v) : 0;
  }();
  if (forced > 0) return forced < 16 ? 16 : (forced > 1024 ? 1024 : forced);
  static const int tmax = [] {
    const char* v = std::getenv("GCONV_TILE_NODES_MAX");
    const int k = v ? std::atoi(v) : 0;
    return k >= GCONV_TILE_NODES ? k : GCONV_TILE_NODES_MAX;
  }();
  if (n <= 0) return GCONV_TILE_NODES;
  const int per_wg = ceil_div(n, kNumCU);
  const int k = ceil_div(per_wg, tmax);             // tiles per workgroup
  const int t = ceil_div(n, kNumCU * k);
  return t < GCONV_TILE_NODES ? GCONV_TILE_NODES : t;
}

constexpr int kSmallMaxEdges = 4096, kSmallMaxNodes = 4096;   // the single-workgroup build (below)

enum {
  INFO_NNZ = 0, INFO_NLONG = 1, INFO_NSEG = 2, INFO_MAXDEG = 3, INFO_BAD = 4,
  INFO_ASYM = 5, INFO_NOTILE = 6, INFO_TILEMAX = 7, INFO_TILEEDGES = 8, INFO_WORDS = 9,
  INFO_DONE = 10   // completion mark of the one-launch builders (pinned host copy only)
};

int tile_slack(int T, int n) {
  static const int div = [] {
    const char* v = std::getenv("GCONV_TILE_SLACK_DIV");
    return v ? std::atoi(v) : 0;
  }();
  const bool one_per_workgroup = ceil_div(n > 0 ? n : 1, T) <= kNumCU;
  const int s = div > 0 ? T / div : (one_per_workgroup ? T / 4 : T);
  return s < 1 ? 1 : (s > 64 ? 64 : s);
}

template <typename K>
struct KeyOps {
  int shift;
  __host__ __device__ K make(uint32_t row, uint32_t col) const {
    return (static_cast<K>(row) << shift) | static_cast<K>(col);
  }
  __host__ __device__ uint32_t row(K k) const { return static_cast<uint32_t>(k >> shift); }
  __host__ __device__ uint32_t col(K k) const {
    return static_cast<uint32_t>(k & ((static_cast<K>(1) << shift) - 1));
  }
};

static int index_bits(int32_t n) {  // bits needed to represent the values 0 .. n
  int b = 1;
  while ((static_cast<int64_t>(1) << b) <= n) ++b;
  return b;
}

template <typename K>
__global__ void make_keys_kernel(const int64_t* __restrict__ ei, int64_t E, int32_t n,
                                 int normalize, int transpose, KeyOps<K> ko,
                                 K* __restrict__ keys, int32_t* __restrict__ info) {
  int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t s = ei[e], d = ei[E + e];
  K key;
  if (s < 0 || s >= n || d < 0 || d >= n) {
    atomicOr(&info[INFO_BAD], 1);
    key = ko.make(n, 0);
  } else if (normalize && s == d) {
    key = ko.make(n, 0);  // add_remaining_self_loops drops it
  } else {
    key = transpose ? ko.make(static_cast<uint32_t>(s), static_cast<uint32_t>(d))
                    : ko.make(static_cast<uint32_t>(d), static_cast<uint32_t>(s));
  }
  keys[e] = key;
}

// One thread per sorted position p in [0, E]: closes the rowptr entries of all
// rows that start between key[p-1] and key[p]; copies the column.
template <typename K>
__global__ void build_csr_kernel(const K* __restrict__ keys, int64_t E, int32_t n, KeyOps<K> ko,
                                 int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                 int32_t* __restrict__ nnz_out, int32_t* __restrict__ near_out) {
  int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  bool near = false;
  if (p <= E) {
    const int64_t r_prev = (p == 0) ? -1 : static_cast<int64_t>(ko.row(keys[p - 1]));
    const int64_t r_cur = (p == E) ? n : static_cast<int64_t>(ko.row(keys[p]));
    for (int64_t r = r_prev + 1; r <= r_cur; ++r) rowptr[r] = static_cast<int32_t>(p);
    if (p < E && r_cur < n) {
      const int64_t c = ko.col(keys[p]);
      col[p] = static_cast<int32_t>(c);
      near = (c * 8) / n == (r_cur * 8) / n;   // column in the row's eighth of the node range (gconv_csr_t::near_edges)
    }
    if (r_cur == n && r_prev < n) *nnz_out = static_cast<int32_t>(p);
  }
  if (near_out) {   // (graphs below kXcdRowsMinNodes never use the count: no atomics on their side-stream builds)
    const int cnt = __popcll(__ballot(near));    // one add per wave
    if ((threadIdx.x & 63) == 0 && cnt > 0) atomicAdd(near_out, cnt);
  }
}

__global__ void row_stats_kernel(const int32_t* __restrict__ rowptr, int32_t n,
                                 float* __restrict__ dis, int32_t* __restrict__ long_rows,
                                 int32_t* __restrict__ long_seg_base,
                                 int32_t* __restrict__ seg_long, uint32_t* __restrict__ long_sync,
                                 int32_t* __restrict__ counters) {
  // counters: [0] n_long, [1] n_seg, [2] max degree
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int deg = 0;
  if (i < n) {
    deg = rowptr[i + 1] - rowptr[i];
    if (dis) dis[i] = 1.0f / sqrtf(static_cast<float>(deg + 1));
    if (deg > GCONV_LONG_ROW) {
      int nseg = (deg + GCONV_SEG_EDGES - 1) / GCONV_SEG_EDGES;
      int lr = atomicAdd(&counters[0], 1);
      int base = atomicAdd(&counters[1], nseg);
      long_rows[lr] = i;
      long_seg_base[lr] = base;
      long_sync[lr] = 0u;
      for (int k = 0; k < nseg; ++k) seg_long[base + k] = lr;
    }
  }
  int m = deg;
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(&counters[2], m);
}

__device__ __forceinline__ int lower_bound_row(const int32_t* __restrict__ col, int lo, int hi,
                                               int v) {
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (col[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Exact test that the stored edge multiset equals its transpose: the k-th copy of
// (i <- j) must be matched by the k-th copy of (j <- i). With equal cardinality on both
// sides (same array) an injective match for every entry is a bijection.
__global__ void sym_check_kernel(const int32_t* __restrict__ rowptr,
                                 const int32_t* __restrict__ col, int32_t n,
                                 const int32_t* __restrict__ nnz_p, int32_t* __restrict__ asym) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= *nnz_p) return;
  // row of entry p: upper_bound in rowptr
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (rowptr[mid + 1] <= p) lo = mid + 1; else hi = mid;
  }
  const int i = lo, j = col[p];
  const int k = p - lower_bound_row(col, rowptr[i], rowptr[i + 1], j);  // rank among duplicates
  const int q = lower_bound_row(col, rowptr[j], rowptr[j + 1], i) + k;
  if (q >= rowptr[j + 1] || col[q] != i) atomicOr(asym, 1);
}

// ---------------------------------------------------------------------------
// arena layout
// ---------------------------------------------------------------------------
struct CsrSlots {
  size_t rowptr, col, long_rows, long_seg_base, seg_long, long_sync;
  size_t recs, tiles, lcol, halo;   // tile data of this direction (unsorted records, sorted records, ...)
};
struct BuildLayout {
  CsrSlots f, b;
  size_t dis, tile_ptr, tile_key, tile_key_out, tile_id, tile_id_out, keys_a, keys_b, ell, info, cub, total, cub_bytes;
  int ntiles, T;
};

// sizes the counting build takes (counting_build_kernel; the constants are repeated there)
static bool counting_build_size_ok(int64_t E, int32_t n) {
  return n > kSmallMaxNodes && n <= (1 << 18) && E > 0 && E <= (static_cast<int64_t>(1) << 20);
}

static int64_t max_long_rows(int64_t E) { return E / (GCONV_LONG_ROW + 1) + 1; }
static int64_t max_segments(int64_t E) { return E / GCONV_SEG_EDGES + max_long_rows(E) + 1; }

static size_t cub_bytes_for(int64_t E, int ntiles) {
  thread_local int64_t last_e = -1;      // (training loops build the same shapes over and over)
  thread_local int last_nt = -1;
  thread_local size_t last = 0;
  if (E == last_e && ntiles == last_nt) return last;
  const int cnt = static_cast<int>(E > 0 ? E : 1);
  size_t s64 = 0, s32 = 0;
  hipcub::DeviceRadixSort::SortKeys(nullptr, s64, static_cast<const uint64_t*>(nullptr),
                                    static_cast<uint64_t*>(nullptr), cnt, 0, 64, nullptr);
  hipcub::DeviceRadixSort::SortKeys(nullptr, s32, static_cast<const uint32_t*>(nullptr),
                                    static_cast<uint32_t*>(nullptr), cnt, 0, 32, nullptr);
  size_t st = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, st, static_cast<const uint32_t*>(nullptr),
                                     static_cast<uint32_t*>(nullptr), static_cast<const uint32_t*>(nullptr),
                                     static_cast<uint32_t*>(nullptr), ntiles, 0, kTileSizeBits, nullptr);
  size_t m = s64 > s32 ? s64 : s32;
  last_e = E; last_nt = ntiles;
  last = m > st ? m : st;
  return last;
}

static BuildLayout build_layout(int64_t E, int32_t n) {
  BuildLayout L{};
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes + 16, 256); return o; };
  const size_t e1 = static_cast<size_t>(E > 0 ? E : 1);
  L.T = tile_nominal(n);
  L.ntiles = ceil_div(n > 0 ? n : 1, L.T);
  const size_t nt = static_cast<size_t>(L.ntiles);
  auto csr = [&](CsrSlots& c) {
    c.rowptr = take((static_cast<size_t>(n) + 1) * 4);
    c.col = take(e1 * 4);
    c.long_rows = take(static_cast<size_t>(max_long_rows(E)) * 4);
    c.long_seg_base = take(static_cast<size_t>(max_long_rows(E)) * 4);
    c.seg_long = take(static_cast<size_t>(max_segments(E)) * 4);
    c.long_sync = take(static_cast<size_t>(max_long_rows(E)) * 4);
    c.recs = take(nt * sizeof(gconv_tile_t));
    c.tiles = take(nt * sizeof(gconv_tile_t));
    c.lcol = take(e1 * 4);
    c.halo = take(nt * kHaloSlot * 4);
  };
  csr(L.f);
  csr(L.b);
  L.dis = take((static_cast<size_t>(n) + 1) * 4);
  L.tile_ptr = take((nt + 1) * 4);
  L.tile_key = take(nt * 4);
  L.tile_key_out = take(nt * 4);
  L.tile_id = take(nt * 4);
  L.tile_id_out = take(nt * 4);
  L.keys_a = take(e1 * 8);
  L.keys_b = take(e1 * 8);
  L.ell = take(counting_build_size_ok(E, n) ? static_cast<size_t>(n) * 64 : 0);   // kEllWidth x 4 bytes per row
  L.info = take(64 * 4);
  L.cub_bytes = cub_bytes_for(E, L.ntiles);
  L.cub = take(L.cub_bytes + 256);
  L.total = off;
  return L;
}

struct CsrOut {
  int32_t *rowptr, *col, *long_rows, *long_seg_base, *seg_long;
  uint32_t* long_sync;
};

// sort + CSR + row statistics for one direction; counters land in info[base .. base + 3]
template <typename K>
static gconv_status_t build_direction(const int64_t* ei, int64_t E, int32_t n, int normalize,
                                      int transpose, int shift, char* arena, const BuildLayout& L,
                                      const CsrOut& o, float* dis, int32_t* info, int base,
                                      hipStream_t stream) {
  KeyOps<K> ko{shift};
  K* keys_a = reinterpret_cast<K*>(arena + L.keys_a);
  K* keys_b = reinterpret_cast<K*>(arena + L.keys_b);
  const int T = 256;
  const K* sorted = keys_a;
  if (E > 0) {
    make_keys_kernel<K><<<static_cast<unsigned>((E + T - 1) / T), T, 0, stream>>>(
        ei, E, n, normalize, transpose, ko, keys_a, info);
    GCONV_LAUNCH_CHECK();
    size_t cub_bytes = L.cub_bytes;
    GCONV_HIP(hipcub::DeviceRadixSort::SortKeys(arena + L.cub, cub_bytes, keys_a, keys_b,
                                                static_cast<int>(E), 0, 2 * shift, stream));
    sorted = keys_b;
  }
  build_csr_kernel<K><<<static_cast<unsigned>((E + 1 + T - 1) / T), T, 0, stream>>>(
      sorted, E, n, ko, o.rowptr, o.col, info + base, n >= kXcdRowsMinNodes ? info + base + 4 : nullptr);
  GCONV_LAUNCH_CHECK();
  if (n > 0) {
    row_stats_kernel<<<(n + T - 1) / T, T, 0, stream>>>(o.rowptr, n, dis, o.long_rows,
                                                        o.long_seg_base, o.seg_long, o.long_sync,
                                                        info + base + 1);
    GCONV_LAUNCH_CHECK();
  }
  return GCONV_OK;
}

// ---------------------------------------------------------------------------
// Small graphs (a 32-molecule batch: ~430 nodes, ~870 edges): the WHOLE build in ONE workgroup.
// The pipeline above is ~18 launches of which each is a few microseconds of latency — 75 us of
// wall time per ESOL batch, half of the five-adapter step it precedes when the caller does not
// prefetch. Here keys, CSR, span counters and tile starts live in LDS; same outputs, same info
// words (long rows come out in ascending row order: the multi-launch path lists them in atomic
// arrival order, both are valid), and the transposed CSR of a directed edge list is built in the
// same launch, so there is ONE host synchronisation whatever the graph.
// ---------------------------------------------------------------------------
// up to 4096 edges; beyond that the multi-launch pipeline is as fast (6960 edges: 103 us against 94).
// One workgroup of 1024 threads: with 256 (cheaper barriers, 4-8 keys per thread) the dependent
// binary searches per thread dominate and the ESOL batch builds in 53 us instead of 48.

__device__ __forceinline__ int small_lower_bound(const uint32_t* keys, int count, uint32_t v) {
  int lo = 0, hi = count;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// in-place exclusive prefix sum of data[0 .. count) (any count); returns the total.
// Chunk sums are scanned inside each wave by shuffles, the wave totals through LDS: two
// barriers per scan (a Hillis-Steele over 1024 LDS words costs twenty, ~0.4 us each at 16 waves).
template <int T>
__device__ __forceinline__ int small_exclusive_scan(int32_t* data, int count, int32_t* tmp) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (count + T - 1) / T;
  const int b0 = min(tid * per, count), b1 = min(b0 + per, count);
  int sum = 0;
  for (int i = b0; i < b1; ++i) sum += data[i];
  int incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == 63) tmp[wave] = incl;
  __syncthreads();
  int wave_base = 0, total = 0;
  for (int wv = 0; wv < T / 64; ++wv) {
    const int v = tmp[wv];
    if (wv < wave) wave_base += v;
    total += v;
  }
  int run = wave_base + incl - sum;
  for (int i = b0; i < b1; ++i) { const int v = data[i]; data[i] = run; run += v; }
  __syncthreads();
  return total;
}

// Bitonic sort of nsort = T * NR keys (ascending), NR keys per thread in registers:
// key i lives in register i / 1024 of thread i % 1024, so a compare-exchange distance j is a lane
// shuffle (j < 64: no memory, no barrier), an LDS exchange between waves (64 <= j < 1024) or a
// swap between two registers of the same thread (j >= 1024). Ten LDS stages at any size instead of
// one barrier per stage of the network (55 at 1024 keys, 91 at 8192).
template <int NR, int T>
__device__ __forceinline__ void small_bitonic_sort(uint32_t (&v)[NR], uint32_t* xch) {
  const int tid = threadIdx.x;
  constexpr int nsort = T * NR;
  for (int k = 2; k <= nsort; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= T) {          // partner = another register of this thread
        const int jr = j / T;
#pragma unroll
        for (int jb = 1; jb < NR; jb <<= 1) {   // (register indices stay compile-time constants)
          if (jr == jb) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
              if ((r & jb) == 0) {
                const int i = r * T + tid;
                const bool up = (i & k) == 0;
                const uint32_t x = v[r], y = v[r | jb];
                if ((x > y) == up) { v[r] = y; v[r | jb] = x; }
              }
            }
          }
        }
      } else if (j >= 64) {              // partner in another wave: through LDS
#pragma unroll
        for (int r = 0; r < NR; ++r) xch[r * T + tid] = v[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int i = r * T + tid;
          const uint32_t y = xch[r * T + (tid ^ j)];
          const bool up = (i & k) == 0, lower = (tid & j) == 0;
          const uint32_t x = v[r];
          v[r] = (lower == up) ? min(x, y) : max(x, y);
        }
        __syncthreads();
      } else {                           // partner in this wave: shuffle
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int i = r * T + tid;
          const uint32_t x = v[r];
          const uint32_t y = static_cast<uint32_t>(__shfl_xor(static_cast<int>(x), j));
          const bool up = (i & k) == 0, lower = (tid & j) == 0;
          v[r] = (lower == up) ? min(x, y) : max(x, y);
        }
      }
    }
  }
}

struct CsrLds {   // the single-workgroup build's CSR as it sits in LDS (tiles.h: tile_cut_wave / tile_halo_wave)
  const int32_t* rowptr_;
  const uint32_t* keys_;
  KeyOps<uint32_t> ko_;
  __device__ __forceinline__ int rowptr(int i) const { return rowptr_[i]; }
  __device__ __forceinline__ int col(int e) const { return static_cast<int>(ko_.col(keys_[e])); }
};

struct SmallBuildArgs {
  const int64_t* ei; int32_t E, n, normalize, shift, nsort, ntiles, T, slack;
  // outputs (device): forward / transposed CSR slots, dis, tiles, info
  int32_t *rowptr_f, *col_f, *long_rows_f, *long_seg_base_f, *seg_long_f; uint32_t* long_sync_f;
  int32_t *rowptr_b, *col_b, *long_rows_b, *long_seg_base_b, *seg_long_b; uint32_t* long_sync_b;
  float* dis; int32_t* tile_ptr; int32_t* info;
  gconv_tile_t *recs_f, *recs_b, *tiles_f, *tiles_b;   // unsorted / size-sorted tile records
  int32_t *lcol_f, *lcol_b, *halo_f, *halo_b;
};

// sort + CSR + row statistics of one direction out of LDS; counters -> sinfo[base .. base + 3]
template <int NR, int T>
__device__ __forceinline__ void small_direction(const SmallBuildArgs& a, int transpose, uint32_t* keys,
                                                int32_t* rowptr, int32_t* tmp, int32_t* flags,
                                                int32_t* flags2, int32_t* out_rowptr, int32_t* out_col,
                                                float* out_dis, int32_t* out_long_rows,
                                                int32_t* out_long_seg_base, int32_t* out_seg_long,
                                                uint32_t* out_long_sync, int base, int32_t* sinfo) {
  const int tid = threadIdx.x;
  const KeyOps<uint32_t> ko{a.shift};
  const int n = a.n, E = a.E;
  uint32_t kv[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int e = r * T + tid;
    uint32_t key = 0xFFFFFFFFu;   // padding of the bitonic network
    if (e < E) {
      const int64_t s = a.ei[e], d = a.ei[static_cast<int64_t>(E) + e];
      if (s < 0 || s >= n || d < 0 || d >= n) {
        sinfo[INFO_BAD] = 1;
        key = ko.make(n, 0);
      } else if (a.normalize && s == d) {
        key = ko.make(n, 0);  // add_remaining_self_loops drops it
      } else {
        key = transpose ? ko.make(static_cast<uint32_t>(s), static_cast<uint32_t>(d))
                        : ko.make(static_cast<uint32_t>(d), static_cast<uint32_t>(s));
      }
    }
    kv[r] = key;
  }
  small_bitonic_sort<NR, T>(kv, keys);
#pragma unroll
  for (int r = 0; r < NR; ++r) keys[r * T + tid] = kv[r];
  __syncthreads();
  // CSR: rowptr[r] = first key of row >= r; nnz = first key of row n
  for (int r = tid; r <= n; r += T) {
    const int p = small_lower_bound(keys, a.nsort, ko.make(static_cast<uint32_t>(r), 0));
    rowptr[r] = p;
    out_rowptr[r] = p;
  }
  __syncthreads();
  const int nnz = rowptr[n];
  for (int p = tid; p < nnz; p += T) out_col[p] = static_cast<int32_t>(ko.col(keys[p]));
  // row statistics
  int maxdeg = 0;
  for (int i = tid; i < n; i += T) {
    const int deg = rowptr[i + 1] - rowptr[i];
    if (out_dis) out_dis[i] = 1.0f / sqrtf(static_cast<float>(deg + 1));
    maxdeg = max(maxdeg, deg);
  }
  for (int off = 32; off > 0; off >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, off));
  if ((tid & 63) == 0 && maxdeg > 0) atomicMax(&sinfo[base + 3], maxdeg);
  __syncthreads();
  int n_long = 0, n_seg = 0;
  if (sinfo[base + 3] > GCONV_LONG_ROW) {   // uniform; long rows in ascending row order (two scans)
    for (int i = tid; i < n; i += T) {
      const int deg = rowptr[i + 1] - rowptr[i];
      const bool is_long = deg > GCONV_LONG_ROW;
      flags[i] = is_long ? 1 : 0;
      flags2[i] = is_long ? (deg + GCONV_SEG_EDGES - 1) / GCONV_SEG_EDGES : 0;
    }
    __syncthreads();
    n_long = small_exclusive_scan<T>(flags, n, tmp);     // flags[i]  = index of row i among the long rows
    n_seg = small_exclusive_scan<T>(flags2, n, tmp);     // flags2[i] = first segment of row i
    for (int i = tid; i < n; i += T) {
      const int deg = rowptr[i + 1] - rowptr[i];
      if (deg > GCONV_LONG_ROW) {
        const int lr = flags[i], sb = flags2[i], nseg = (deg + GCONV_SEG_EDGES - 1) / GCONV_SEG_EDGES;
        out_long_rows[lr] = i;
        out_long_seg_base[lr] = sb;
        out_long_sync[lr] = 0u;
        for (int k = 0; k < nseg; ++k) out_seg_long[sb + k] = lr;
      }
    }
  }
  if (tid == 0) { sinfo[base + 0] = nnz; sinfo[base + 1] = n_long; sinfo[base + 2] = n_seg; }
  __syncthreads();
}

template <int NR, int T>
__global__ __launch_bounds__(T) void small_graph_build_kernel(SmallBuildArgs a) {
  extern __shared__ int32_t sm[];
  // layout: keys[nsort] | rowptr[n + 1] | flags[n] | flags2[n] | tmp[1024] | tsize[ntiles] | tptr[ntiles + 1] | sinfo[64]
  uint32_t* keys = reinterpret_cast<uint32_t*>(sm);
  int32_t* rowptr = sm + a.nsort;
  int32_t* flags = rowptr + a.n + 1;
  int32_t* flags2 = flags + a.n;
  int32_t* tmp = flags2 + a.n;
  int32_t* tsize = tmp + T;
  int32_t* tptr = tsize + a.ntiles;
  int32_t* sinfo = tptr + a.ntiles + 1;   // the info words (enum above; forward counters at 16, transposed at 24)
  const int tid = threadIdx.x, n = a.n;
  const int lane = tid & 63, wave = tid >> 6;
  if (tid < 64) sinfo[tid] = 0;
  __syncthreads();

  small_direction<NR, T>(a, 0, keys, rowptr, tmp, flags, flags2, a.rowptr_f, a.col_f, a.dis, a.long_rows_f,
                  a.long_seg_base_f, a.seg_long_f, a.long_sync_f, 16, sinfo);
  const KeyOps<uint32_t> ko{a.shift};
  const int nnz = rowptr[n];

  // exact symmetry test (sym_check_kernel) on the LDS copy: col(p) = low field of keys[p]
  for (int p = tid; p < nnz; p += T) {
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (rowptr[mid + 1] <= p) lo = mid + 1; else hi = mid;
    }
    const uint32_t i = static_cast<uint32_t>(lo), j = ko.col(keys[p]);
    const int k = p - (rowptr[i] + small_lower_bound(keys + rowptr[i], rowptr[i + 1] - rowptr[i], ko.make(i, j)));  // rank among duplicates
    const int q = rowptr[j] + small_lower_bound(keys + rowptr[j], rowptr[j + 1] - rowptr[j], ko.make(j, i)) + k;
    if (q >= rowptr[j + 1] || keys[q] != ko.make(j, i)) sinfo[INFO_ASYM] = 1;
  }
  __syncthreads();   // (also: the CSR this workgroup wrote to global memory is complete)

  // tiles (tiles.h): one wave per tile places its cuts, finds its halo and translates its columns —
  // the routine of the multi-launch build, reading the CSR out of LDS (rowptr as it stands, a column
  // = the low field of a sorted key)
  const TileInfoSlots is{INFO_NOTILE, INFO_TILEMAX, INFO_TILEEDGES};
  const CsrLds csr_lds{rowptr, keys, ko};
  for (int t = wave; t < a.ntiles; t += T / 64) {
    int ts, te, e0, e1;
    tile_cuts_wave(csr_lds, n, t, a.ntiles, a.T, a.slack, lane, ts, te, e0, e1);
    if (lane == 0) {
      a.tile_ptr[t] = ts;
      tptr[t] = ts;
      if (t == a.ntiles - 1) { a.tile_ptr[a.ntiles] = n; tptr[a.ntiles] = n; }
    }
    tile_halo_wave(csr_lds, a.dis, ts, te, e0, e1, t, a.lcol_f, a.halo_f, a.recs_f + t, sinfo, is, lane);
  }
  __syncthreads();
  const bool asym = sinfo[INFO_ASYM] != 0;   // uniform: directed edge list -> S^T as well, in the same launch
  if (asym) {
    small_direction<NR, T>(a, 1, keys, rowptr, tmp, flags, flags2, a.rowptr_b, a.col_b, nullptr, a.long_rows_b,
                    a.long_seg_base_b, a.seg_long_b, a.long_sync_b, 24, sinfo);
    __syncthreads();
    for (int t = wave; t < a.ntiles; t += T / 64)   // (keys / rowptr now hold the transposed CSR)
      tile_halo_wave(csr_lds, a.dis, tptr[t], tptr[t + 1], csr_lds.rowptr(tptr[t]), csr_lds.rowptr(tptr[t + 1]), t,
                     a.lcol_b, a.halo_b, a.recs_b + t, sinfo, is, lane);
    __syncthreads();
  }
  // size-sorted tile list: rank by (rows incl. halo descending, index ascending) == tile_sort_small_kernel;
  // node order when every persistent workgroup gets ONE tile anyway (the kernels then know a tile's
  // halo slot from the workgroup index, before its record has arrived)
  for (int t = tid; t < a.ntiles; t += T) tsize[t] = a.recs_f[t].n0;
  __syncthreads();
  for (int t = tid; t < a.ntiles; t += T) {
    const int s0 = tsize[t];
    int rank = a.ntiles <= kNumCU ? t : 0;
    for (int u = 0; u < a.ntiles && a.ntiles > kNumCU; ++u) {
      const int su = tsize[u];
      rank += (su > s0) || (su == s0 && u < t);
    }
    a.tiles_f[rank] = a.recs_f[t];
    if (asym) a.tiles_b[rank] = a.recs_b[t];
  }
  // counters -> pinned host memory (a.info is the device address of a mapped host buffer): every global
  // store of the workgroup first, then 63 words, then the completion mark (word INFO_DONE) the host polls
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid < 64) {
    if (tid != INFO_DONE) a.info[tid] = sinfo[tid];
    __threadfence_system();
    if (tid == INFO_DONE) a.info[tid] = 1;
    __threadfence_system();
  }
}

// ---------------------------------------------------------------------------
// Mid-size graphs of small degree (molecule batches of 10^4 - 10^5 nodes): CSR by COUNTING, no
// comparison sort of the edge list, in ONE launch. The radix pipeline above is ~12 launches (the
// hipCUB sort alone is four or five) and two memsets, ~100 us of wall time for a 512-molecule
// shard whose five-adapter step takes 200. Here kCountGrid workgroups — all resident: one per CU —
// walk through three phases together, separated by two grid barriers (grid_sync):
//   A gather : every edge takes the next free slot of its row (atomic fetch-add on the row's degree
//              counter: the returned value is its position) and drops its source into that slot of
//              the row's 64-byte line of a temporary ELL table (kEllWidth = 16 slots); it also
//              counts itself into its row CHUNK's total (256 chunks: one per workgroup of phase B)
//              — and validates the indices on the way
//   B rows   : workgroup w owns chunk w: the chunk totals below it are its first edge offset, a
//              workgroup scan of the degrees gives rowptr; each thread ranks its rows' <= 16 slots
//              in registers and writes them ascending into col; dis; maximum degree. Then, a thread
//              per edge, the exact symmetry test on the ELL rows: the multiset equals its transpose
//              iff every stored (i <- j) has as many copies as (j <- i)
//   C tiles  : one wave per tile (tiles.h); the counters go back to zero; the last workgroup to
//              finish copies the info words into pinned host memory: the host only synchronises the
//              stream, no D2H copy command
// A row with more than 16 edges, a directed edge list or a graph beyond kCountMaxNodes make the
// host repeat the build on the radix pipeline (arrival order in the ELL rows is not reproducible,
// the ascending row is).
//
// State between launches — barrier words, info words, chunk totals, degree counters — is library-
// owned device memory, one block per device, ALL ZEROS between launches (the kernel puts back what
// it used), used under a host mutex: the build is synchronous, so no two launches ever share it.
// No zeroing pass, no third barrier. A workgroup that waits longer than kSpinMax polls (~1 s: some
// workgroup never became resident, which a grid of one workgroup per CU rules out short of a hung
// device) leaves; the others follow at their next barrier, so the grid always drains; the host
// sees no completion mark, clears the state block and rebuilds on the radix pipeline.
//
// Data that one workgroup writes and another reads in a later phase of the SAME launch goes through
// device-scope accesses (sc1: stores write through the XCD's L2, loads do not hit a stale line of
// it): the eight L2s are not coherent with one another inside a kernel, and the generic remedy — a
// device-scope fence per barrier, i.e. an L2 write-back + invalidate from each of the 256 workgroups —
// cost 23 us per barrier (measured: 159 us per build; a chain of six launches took 97).
// ---------------------------------------------------------------------------
constexpr int kCountMaxNodes = 1 << 18, kCountMinNodes = kSmallMaxNodes + 1;
constexpr int64_t kCountMaxEdges = static_cast<int64_t>(1) << 20;
// grid widths: kCountGrid workgroups (a device with that many CUs: one each) for in-stream builds, the
// narrow kCountGridNarrow for builds next to other kernels (GCONV_BUILD_BACKGROUND) and smaller devices
constexpr int kCountGrid = 256, kCountGridNarrow = 64, kCountThreads = 256, kCountGroups = 8;
constexpr int kEllWidth = 16;
constexpr int kSpinMax = 1 << 20;
// state block, in 4-byte words: barrier words (one 256-byte line per group + one for the grid),
// info words, chunk totals, degree counters
constexpr int kBarLine = 64;
constexpr int kStateBar = 0, kStateInfo = (kCountGroups + 2) * kBarLine, kStateChunk = kStateInfo + 64,
              kStateDeg = kStateChunk + kCountGrid, kStateWords = kStateDeg + kCountMaxNodes + 1;
static_assert(kEllWidth <= GCONV_LONG_ROW, "rows of the counting build are never long rows");

struct CountArgs {
  const int64_t* ei; int64_t E; int32_t n, normalize;
  int32_t* state;                              // library-owned (see above)
  int32_t* ell;                                // arena scratch: [n][kEllWidth]
  int32_t *rowptr, *col; float* dis;
  int32_t *tile_ptr, *lcol, *slots; gconv_tile_t* recs;
  int32_t* host_info;
  int32_t ntiles, T, slack, fwd;               // fwd: first forward counter word of info
  TileInfoSlots is;
  int32_t phases;                              // bit set of the phases this launch runs: 1 = A, 2 = B, 4 = C (7: all,
                                               // separated by grid barriers; one bit: a launch per phase, no barrier)
};

__device__ __forceinline__ int ldc(const int32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stc(int32_t* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#ifndef GCONV_COUNT_STAMPS
#define GCONV_COUNT_STAMPS 0
#endif
#ifndef GCONV_COUNT_SPLIT_COH   // (lab: 1 = the split launches with the one-launch kernel's device-scope accesses)
#define GCONV_COUNT_SPLIT_COH 0
#endif
#ifndef GCONV_COUNT_TILES_CACHED
#define GCONV_COUNT_TILES_CACHED 1
#endif
constexpr int kAuxDeviceScope = 16;            // sc1 of a buffer access
struct CsrCoherent {   // tiles.h accessor over the CSR this launch has just produced
  const int32_t* rowptr_;
  const int32_t* col_;
  __device__ __forceinline__ int rowptr(int i) const { return ldc(rowptr_ + i); }
  __device__ __forceinline__ int col(int e) const { return ldc(col_ + e); }
};

// All GRID workgroups arrive — in kCountGroups groups, each on its own counter (256 arrivals
// on one word took 6 - 10 us: device-scope atomics on one address are serialised at the memory
// side), the last of a group for the group at the grid's counter — and the last one of all opens
// the next generation in every group's line. false: gave up waiting.
// No device-scope fence (see above): every wave waits for its stores (vmcnt(0)), which for a
// device-scope store means written through.
template <int GRID>
__device__ __forceinline__ bool grid_sync(int32_t* state, uint32_t& gen, int* ok_s) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t* bar = reinterpret_cast<uint32_t*>(state + kStateBar);
    const int grp = blockIdx.x % kCountGroups;
    uint32_t* mine = bar + grp * kBarLine;           // [0] arrivals of the group, [1] its copy of the generation
    uint32_t* top = bar + kCountGroups * kBarLine;   // [0] groups arrived
    bool ok = true;
    bool opener = false;
    if (__hip_atomic_fetch_add(&mine[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
        static_cast<uint32_t>(GRID / kCountGroups - 1)) {
      __hip_atomic_store(&mine[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(&top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
          static_cast<uint32_t>(kCountGroups - 1)) {
        __hip_atomic_store(&top[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int g = 0; g < kCountGroups; ++g)
          __hip_atomic_store(&bar[g * kBarLine + 1], gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        opener = true;
      }
    }
    if (!opener) {
      int spins = 0;
      while (__hip_atomic_load(&mine[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinMax) { ok = false; break; }
      }
    }
    *ok_s = ok ? 1 : 0;
  }
  ++gen;
  __syncthreads();
  return *ok_s != 0;
}

// COH: data that one workgroup writes and another reads travels INSIDE the launch (phases separated by grid barriers):
// device-scope accesses. !COH: a launch per phase (the split builder of background builds) — what a phase reads was
// written by an earlier LAUNCH, so plain cached accesses do: the per-edge symmetry test of phase B then finds a row's
// 64-byte ELL line and degree in L1 / L2 instead of fetching both across the fabric for every edge.
template <int GRID, bool COH = true>
__global__ __launch_bounds__(kCountThreads) void counting_build_kernel(CountArgs a) {
  static_assert(GRID % 64 == 0 && GRID % kCountGroups == 0 && GRID <= kCountGrid, "chunk totals: GRID / 64 per lane");
  __shared__ int ok_s, last_s;
  __shared__ int32_t wred[kCountThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
  const int64_t gtid = static_cast<int64_t>(wg) * kCountThreads + tid, gsize = static_cast<int64_t>(GRID) * kCountThreads;
  const int32_t n = a.n;
  const int64_t E = a.E;
  int32_t* info = a.state + kStateInfo;
  int32_t* chunk_cnt = a.state + kStateChunk;
  int32_t* deg = a.state + kStateDeg;
  // generation of this workgroup's group line, read before any arrival (all groups hold the same)
  uint32_t gen = static_cast<uint32_t>(ldc(a.state + kStateBar + (wg % kCountGroups) * kBarLine + 1));
  const int chunk = (n + GRID - 1) / GRID;                        // rows per workgroup in phase B
  const bool run_a = a.phases & 1, run_b = a.phases & 2, run_c = a.phases & 4;   // grid-uniform
#define GCONV_GRID_SYNC()                               \
  do {                                                  \
    if (!grid_sync<GRID>(a.state, gen, &ok_s)) return;        \
  } while (0)
#if GCONV_COUNT_STAMPS   // lab builds (tools/lab/k0_stamps.py): workgroup 0's clock at the phase boundaries
#define GCONV_STAMP(k) do { if (wg == 0 && tid == 0) stc(info + 32 + (k), static_cast<int32_t>(wall_clock64())); } while (0)
#else
#define GCONV_STAMP(k) do { } while (0)
#endif

  auto ldv = [](const int32_t* p) -> int { if constexpr (COH) return ldc(p); else return *p; };
  auto stv = [](int32_t* p, int v) { if constexpr (COH) stc(p, v); else *p = v; };

  // ---- A: gather ----
  GCONV_STAMP(0);
  int s0 = 0, d0 = 0;                                  // the thread's first edge, kept for phase B (0, 0: none / invalid)
  if (!run_a && run_b && gtid < E) {                   // (phase B as its own launch: asked for again)
    const int64_t s = a.ei[gtid], d = a.ei[E + gtid];
    if (s >= 0 && s < n && d >= 0 && d < n) { s0 = static_cast<int>(s); d0 = static_cast<int>(d); }
  }
  for (int64_t e0 = static_cast<int64_t>(wg) * kCountThreads + (tid & ~63); run_a && e0 < E; e0 += gsize) {   // wave-uniform trip count
    const int64_t e = e0 + lane;
    int ck = -1;                                       // the edge's row chunk, -1: not stored
    if (e < E) {
      const int64_t s = a.ei[e], d = a.ei[E + e];
      if (s < 0 || s >= n || d < 0 || d >= n) {
        atomicOr(&info[INFO_BAD], 1);
      } else {
        const int di = static_cast<int>(d);
        if (e == gtid) { s0 = static_cast<int>(s); d0 = di; }
        if (!(a.normalize && s == d)) {
          const int p = atomicAdd(&deg[di], 1);
          if (p < kEllWidth) stv(a.ell + static_cast<size_t>(di) * kEllWidth + p, static_cast<int32_t>(s));
          ck = di / chunk;
        }
      }
    }
    // chunk totals, one atomic per distinct chunk of the wave (a collated batch keeps a wave's 64 edges
    // within two or three chunks; 64 same-address atomics per wave made this phase 33 us)
    unsigned long long todo = __ballot(ck >= 0);
    while (todo) {
      const int c0 = __builtin_amdgcn_readlane(ck, __ffsll(static_cast<long long>(todo)) - 1);
      const unsigned long long same = __ballot(ck == c0);
      if (lane == __ffsll(static_cast<long long>(same)) - 1) atomicAdd(&chunk_cnt[c0], __popcll(same));
      todo &= ~same;
    }
  }
  GCONV_STAMP(1);
  if (run_a && run_b) GCONV_GRID_SYNC();
  GCONV_STAMP(2);

  // ---- B: rows. Workgroup w owns rows [w * chunk, (w + 1) * chunk), thread t rows t * per .. ----
  const int per = (chunk + kCountThreads - 1) / kCountThreads;
  const int r0 = min(wg * chunk + tid * per, n);
  const int r1 = min(min(r0 + per, (wg + 1) * chunk), n);
  const rsrc_t ell_r = make_rsrc(a.ell, static_cast<uint32_t>(n) * kEllWidth * 4u);
  auto ell_row = [&](int row, i32x4 (&v)[kEllWidth / 4]) {
#pragma unroll
    for (int q = 0; q < kEllWidth / 4; ++q)
      v[q] = __builtin_amdgcn_raw_buffer_load_b128(ell_r, (row * kEllWidth + 4 * q) * 4, 0, COH ? kAuxDeviceScope : 0);
  };
  if (run_b) {
    // every load of the phase that does not depend on the scan is issued here, in one batch: the chunk
    // totals, the degrees and the slots of the thread's first row, and both ELL rows of its first edge
    const bool bad = ldv(info + INFO_BAD) != 0;                   // grid-uniform: the host refuses the list
    int cc[GRID / 64];
#pragma unroll
    for (int k = 0; k < GRID / 64; ++k) cc[k] = ldv(chunk_cnt + k * 64 + lane);
    i32x4 row0[kEllWidth / 4], rd[kEllWidth / 4], rs[kEllWidth / 4];
    ell_row(min(r0, n - 1), row0);
    const int dd0 = ldv(deg + d0), ds0 = ldv(deg + s0);
    ell_row(d0, rd);
    ell_row(s0, rs);
    int below = 0, all = 0;                            // GRID chunk totals: GRID / 64 per lane
#pragma unroll
    for (int k = 0; k < GRID / 64; ++k) {
      all += cc[k];
      if (k * 64 + lane < wg) below += cc[k];
    }
    below = wave_sum_i(below);
    all = wave_sum_i(all);
    int mysum = 0, mymax = 0;
    for (int i = r0; i < r1; ++i) {
      const int v = ldv(deg + i);
      mysum += v;
      mymax = max(mymax, v);
    }
    int incl = mysum;                                  // inclusive scan over the workgroup's threads
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wred[wave] = incl;
    mymax = wave_max_i(mymax);
    if (lane == 0 && mymax > 0) atomicMax(&info[a.fwd + 3], mymax);
    __syncthreads();
    int wbase = 0;
#pragma unroll
    for (int w = 0; w < kCountThreads / 64; ++w)
      if (w < wave) wbase += wred[w];
    int run = below + wbase + incl - mysum;
    for (int i = r0; i < r1; ++i) {
      const int dg = ldv(deg + i);
      stv(a.rowptr + i, run);
      if (a.dis) stv(reinterpret_cast<int32_t*>(a.dis) + i, __float_as_int(1.0f / sqrtf(static_cast<float>(dg + 1))));
      if (i > r0) ell_row(i, row0);
      if (!bad && dg <= kEllWidth && dg > 0) {         // the row's slots, ranked in registers -> ascending
#pragma unroll
        for (int j = 0; j < kEllWidth; ++j) {
          if (j < dg) {
            const int vj = row0[j / 4][j % 4];
            int rank = 0;
#pragma unroll
            for (int k = 0; k < kEllWidth; ++k) {
              const int vk = row0[k / 4][k % 4];
              rank += (k < dg) && (vk < vj || (vk == vj && k < j));
            }
            stv(a.col + run + rank, vj);
          }
        }
      }
      run += dg;
    }
    if (wg == 0 && tid == 0) {
      stv(a.rowptr + n, all);
      atomicExch(&info[a.fwd + 0], all);               // nnz; no long rows on this path (n_long = n_seg = 0)
    }
    if (!bad) {   // symmetry, a thread per edge, on the ELL rows (complete since the barrier)
      bool asym = false;
      auto test = [&](int s, int d, int dd, int ds) {
        if (s == d || dd > kEllWidth || ds > kEllWidth) return;    // (the host rebuilds: maximum degree)
        int fwd = 0, back = 0;
#pragma unroll
        for (int k = 0; k < kEllWidth; ++k) {
          fwd += (k < dd) && rd[k / 4][k % 4] == s;
          back += (k < ds) && rs[k / 4][k % 4] == d;
        }
        asym |= fwd != back;
      };
      // gconv_csr_t::near_edges on the way: stored edges whose column lies in its row's eighth of the node range
      int near_cnt = 0;
      auto near = [&](int s, int d) {
        return !(a.normalize && s == d) && (static_cast<int64_t>(s) * 8) / n == (static_cast<int64_t>(d) * 8) / n;
      };
      test(s0, d0, dd0, ds0);                          // (0, 0 when the thread has no first edge: skipped as a self-loop)
      const bool count_near = n >= kXcdRowsMinNodes;   // (grid-uniform; smaller graphs never use the count)
      if (count_near && gtid < E) near_cnt += near(s0, d0);
      for (int64_t e = gtid + gsize; e < E; e += gsize) {
        const int s = static_cast<int>(a.ei[e]), d = static_cast<int>(a.ei[E + e]);
        const int dd = ldv(deg + d), ds = ldv(deg + s);
        ell_row(d, rd);
        ell_row(s, rs);
        test(s, d, dd, ds);
        if (count_near) near_cnt += near(s, d);
      }
      if (asym) atomicOr(&info[INFO_ASYM], 1);
      if (count_near) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) near_cnt += __shfl_xor(near_cnt, off);
        if (lane == 0 && near_cnt > 0) atomicAdd(&info[a.fwd + 4], near_cnt);
      }
    }
  }
  GCONV_STAMP(3);
  if (run_b && run_c) GCONV_GRID_SYNC();
  GCONV_STAMP(4);
  if (!run_c) return;

  // ---- C: tiles: tile t on workgroup t mod grid (one wave per tile, a tile per CU first); counters
  //         back to zero ----
  for (int i = r0; i < r1; ++i) stc(deg + i, 0);
  if (tid == 0) stc(chunk_cnt + wg, 0);
  const int c_bad = ldc(info + INFO_BAD), c_deg = ldc(info + a.fwd + 3), c_asym = ldc(info + INFO_ASYM);   // one batch
  if ((c_bad == 0) & (c_deg <= kEllWidth) & (c_asym == 0)) {
#if GCONV_COUNT_TILES_CACHED
    // rowptr / col / dis are complete and written through: drop what this XCD's L2 and the CU's L1 may
    // hold of them (an invalidate, no write-back) and read them like any kernel would, cached
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const CsrGlobal csr{a.rowptr, a.col};
#else
    const CsrCoherent csr{a.rowptr, a.col};
#endif
    for (int t = wg + GRID * wave; t < a.ntiles; t += GRID * (kCountThreads / 64)) {
      int ts, te, e0, e1;
      tile_cuts_wave(csr, n, t, a.ntiles, a.T, a.slack, lane, ts, te, e0, e1);
      if (lane == 0) {
        a.tile_ptr[t] = ts;
        if (t == a.ntiles - 1) a.tile_ptr[a.ntiles] = n;
      }
      tile_halo_wave<decltype(csr), !GCONV_COUNT_TILES_CACHED>(csr, a.dis, ts, te, e0, e1, t, a.lcol, a.slots, a.recs + t, info,
                                                                a.is, lane);
    }
  }
  GCONV_STAMP(5);
  // ---- the last workgroup to get here reports to the host and clears the info words ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    uint32_t* done = reinterpret_cast<uint32_t*>(a.state + kStateBar) + (kCountGroups + 1) * kBarLine;
    const uint32_t old = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_s = old == static_cast<uint32_t>(GRID - 1);
    if (last_s) __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (last_s && tid < 64) {
    int v = ldc(info + tid);
#if GCONV_COUNT_STAMPS
    if (tid == 32 + 6) v = static_cast<int32_t>(wall_clock64());
#endif
    if (tid == INFO_DONE) v = 1;
    stc(info + tid, 0);
    if (tid != INFO_DONE) a.host_info[tid] = v;
    __threadfence_system();   // the other 63 words and the zeroed state block first, then the completion mark
    if (tid == INFO_DONE) a.host_info[tid] = v;
    __threadfence_system();
  }
#undef GCONV_GRID_SYNC
#undef GCONV_STAMP
}

// library-owned state of the one-launch builders, one per device: the pinned words both report into,
// and counting_build_kernel's device block (see the comment above). Used under `mu`.
struct CountState {
  std::mutex mu;
  int32_t* state = nullptr;       // device: kStateWords words, zero between launches
  int32_t* host_info = nullptr;   // pinned, mapped: 64 words
  int32_t* host_info_dev = nullptr;   // its device address
  bool broken = true;             // the device block must be cleared before its next use

  gconv_status_t ensure(bool device_block) {
    if (!host_info) {
      GCONV_HIP(hipHostMalloc(reinterpret_cast<void**>(&host_info), 64 * sizeof(int32_t), hipHostMallocMapped));
      void* p = nullptr;
      GCONV_HIP(hipHostGetDevicePointer(&p, host_info, 0));
      host_info_dev = static_cast<int32_t*>(p);
    }
    if (device_block) {
      if (!state) GCONV_HIP(hipMalloc(reinterpret_cast<void**>(&state), kStateWords * sizeof(int32_t)));
      if (broken) {   // first use, or a launch that gave up: everything back to zero
        GCONV_HIP(hipMemset(state, 0, kStateWords * sizeof(int32_t)));
        broken = false;
      }
    }
    return GCONV_OK;
  }
  // Waits for the completion mark in pinned memory, looking at the stream now and then (a kernel that
  // gave up or faulted never writes it). No stream synchronisation once the mark has arrived: the
  // kernel is at its last instruction, its results are ordered on `stream` like any kernel's, and the
  // runtime's completion handling (5 - 6 us, hipStreamQuery as much as hipStreamSynchronize; a D2H
  // copy of the counters another 4) stays off the build's path. false: no mark.
  bool wait_mark(hipStream_t stream, int32_t* h) {
    volatile int32_t* hv = host_info;
    for (long spins = 0; !hv[INFO_DONE]; ++spins)
      if ((spins & 1023) == 1023 && hipStreamQuery(stream) != hipErrorNotReady) break;
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!hv[INFO_DONE]) return false;
    for (int i = 0; i < 64; ++i) h[i] = hv[i];
    return true;
  }
  void arm() {
    volatile int32_t* hv = host_info;
    hv[INFO_DONE] = 0;
    std::atomic_thread_fence(std::memory_order_release);
  }
};
static CountState* count_state() {
  static CountState states[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  return &states[dev];
}
// the state of the current device with its memory in place, or null: a one-launch builder whose few
// library-owned bytes cannot be allocated (device memory exhausted by the caller) is skipped, the radix
// pipeline — which needs nothing but the arena — builds the plan
static CountState* usable_state(bool device_block) {
  CountState* cs = count_state();
  if (!cs) return nullptr;
  std::lock_guard<std::mutex> lock(cs->mu);
  if (cs->ensure(device_block) != GCONV_OK) {
    (void)hipGetLastError();   // (the failed allocation's sticky error must not fail the next launch check)
    return nullptr;
  }
  return cs;
}

// How the counting builder may be launched on the current device. The fused launch spins at two grid
// barriers, so ALL its workgroups must be able to be resident together: grid <= CUs x the kernel's
// workgroups per CU (hipDeviceProp_t / the occupancy query, asked once per device) — a device too small
// for the wide grid takes the narrow one, one too small for that takes the radix pipeline at once, instead
// of finding out by a ~1 s spin. GCONV_COUNT_COOPERATIVE=1 launches it with hipLaunchCooperativeKernel (the
// runtime then checks the same thing per launch and refuses with an error): measured +9 us of wall time per
// plan (HIV-512 49.7 -> 59.0 us, HIV-4096 130 -> 143; profiles/r04_lab_k0_background.txt), so it stays a knob.
// Background builds (GCONV_BUILD_BACKGROUND: next to the caller's kernels, which may hold every CU for a
// while) must not wait inside a kernel. Three flavours, all producing the same plan (GCONV_BG_BUILD):
//   radix (default)  the multi-launch sort pipeline: 99 us of wall time per HIV-512 plan on an idle device,
//                    but the LEAST disturbance to the step it runs beside (0.205 ms/step; 0.191 with a cached plan)
//   split            this builder as one narrow launch (64 workgroups) per phase, no barrier: 48 us per plan,
//                    0.217 ms/step — a few 256-thread workgroups that sit on 64 CUs for ~10 us each cost the
//                    one-tile-per-CU kernels of the step a second round more often than a dozen 2 - 5 us kernels do
//   fused            the narrow grid with its two barriers in one launch: 48 us per plan, 0.222 ms/step
struct CountPlan { int grid; bool split, cooperative; };
// background builds of graphs this large take the split counting builder (2048 molecules = 52 k nodes: 0.4927 -> 0.4772
// ms/step against the radix pipeline; 1024 molecules = 26 k nodes: equal; 512: radix 0.2003 against 0.2037 —
// tools/lab/r05_prio2.sh)
constexpr int32_t kBgSplitMinNodes = 1 << 15;
static CountPlan count_plan(bool background, int32_t num_nodes) {
  struct Caps { int wide = -1, narrow = -1; };
  static Caps caps[16];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return {0, false, false};
  {
    std::lock_guard<std::mutex> lock(mu);
    if (caps[dev].wide < 0) {
      hipDeviceProp_t pr;
      int per_wide = 0, per_narrow = 0;
      if (hipGetDeviceProperties(&pr, dev) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_wide, counting_build_kernel<kCountGrid>, kCountThreads, 0) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_narrow, counting_build_kernel<kCountGridNarrow>, kCountThreads, 0) != hipSuccess) {
        (void)hipGetLastError();
        caps[dev].wide = caps[dev].narrow = 0;
      } else {
        caps[dev].wide = pr.multiProcessorCount * per_wide >= kCountGrid ? 1 : 0;
        caps[dev].narrow = pr.multiProcessorCount * per_narrow >= kCountGridNarrow ? 1 : 0;
      }
    }
  }
  const char* cv = std::getenv("GCONV_COUNT_COOPERATIVE");   // (read per call: tests and A/B runs)
  const bool coop = cv && cv[0] == '1';
  const char* bg = std::getenv("GCONV_BG_BUILD");   // (read per call: bench A/B)
  if (background) {
    if (bg && bg[0] == 's') return {kCountGridNarrow, true, false};   // split launches need no co-residency at all
    if (bg && bg[0] == 'w') return {kCountGrid, true, false};         // lab: the same three launches, 256 short-lived workgroups each
    if (bg && bg[0] == 'f') return {caps[dev].narrow ? kCountGridNarrow : 0, false, coop};
    // Round 5, next to the whole 4096-molecule batch's step (103 k nodes; tools/lab/r05_bg.sh, one box, medians over 4 s):
    // radix pipeline 0.9637 ms/step, the counting builder as three narrow launches 0.9400, three wide ones 0.9490, one
    // narrow launch 0.9626 (0.8459 with one cached plan); next to the 512-molecule shard's step the radix pipeline
    // stays the best neighbour (0.2157 against 0.2231 / 0.2398 / 0.2303). Hence by size (GCONV_BG_BUILD=radix forces);
    // the threshold was then placed between the 1024- and the 2048-molecule batch (kBgSplitMinNodes).
    if (!(bg && bg[0] == 'r') && num_nodes >= kBgSplitMinNodes) return {kCountGridNarrow, true, false};
    return {0, false, false};                          // radix pipeline (measured: least disturbance, see above)
  }
  if (caps[dev].wide) return {kCountGrid, false, coop};
  if (caps[dev].narrow) return {kCountGridNarrow, false, coop};
  return {0, false, false};
}

static bool counting_build_ok(int64_t E, int32_t n) {   // (env read per call: the tests compare the builders)
  const char* v = std::getenv("GCONV_NO_COUNTING_BUILD");
  const bool off = v != nullptr && v[0] != '\0' && v[0] != '0';
  static_assert(kCountMinNodes == kSmallMaxNodes + 1 && kCountMaxNodes == (1 << 18) && kEllWidth * 4 == 64 &&
                kCountMaxEdges == (static_cast<int64_t>(1) << 20), "counting_build_size_ok repeats these");
  return !off && counting_build_size_ok(E, n);
}

static bool small_build_ok(int64_t E, int32_t n) {   // (env read per call: the tests compare both builders)
  const char* v = std::getenv("GCONV_NO_SMALL_BUILD");
  const bool off = v != nullptr && v[0] != '\0' && v[0] != '0';
  return !off && n >= 1 && n <= kSmallMaxNodes && E <= kSmallMaxEdges;
}

}  // namespace gconv

using namespace gconv;

// size sort of the tile records (queued on `stream`, not waited for): node order for <= 256 tiles (one
// per persistent workgroup: no launch; the kernels know a tile's halo slot from the workgroup index,
// before its record has arrived), one ranking workgroup up to 2048 tiles, a stable radix sort beyond
template <class Layout>
static gconv_status_t sort_tiles(const Layout& L, char* arena, gconv_tile_t* recs_f, gconv_tile_t* recs_b,
                                 gconv_tile_t* tiles_f, gconv_tile_t* tiles_b, gconv_graph_t* g,
                                 hipStream_t stream) {
  const int T = 256;
  if (L.ntiles <= kNumCU) {
    g->tiles_fwd.tiles = recs_f;
    g->tiles_bwd.tiles = recs_b ? recs_b : recs_f;
  } else if (L.ntiles <= kSmallSortMax) {
    tile_sort_small(recs_f, recs_b, L.ntiles, tiles_f, tiles_b, stream);
    GCONV_LAUNCH_CHECK();
  } else {  // largest tile first; stable, so the order is a function of the graph alone
    uint32_t* key = reinterpret_cast<uint32_t*>(arena + L.tile_key);
    uint32_t* key_out = reinterpret_cast<uint32_t*>(arena + L.tile_key_out);
    uint32_t* id = reinterpret_cast<uint32_t*>(arena + L.tile_id);
    uint32_t* id_out = reinterpret_cast<uint32_t*>(arena + L.tile_id_out);
    tile_sort_keys_kernel<<<(L.ntiles + T - 1) / T, T, 0, stream>>>(recs_f, L.ntiles, key, id);
    GCONV_LAUNCH_CHECK();
    size_t cub_bytes = L.cub_bytes;
    GCONV_HIP(hipcub::DeviceRadixSort::SortPairs(arena + L.cub, cub_bytes, key, key_out, id, id_out, L.ntiles, 0,
                                                 kTileSizeBits, stream));
    tile_sort_gather_kernel<<<(L.ntiles + T - 1) / T, T, 0, stream>>>(recs_f, recs_b, id_out, L.ntiles, tiles_f, tiles_b);
    GCONV_LAUNCH_CHECK();
  }
  return GCONV_OK;
}

extern "C" int32_t gconv_tile_nodes(int32_t num_nodes) { return tile_nominal(num_nodes); }

extern "C" size_t gconv_graph_build_bytes(int64_t num_edges, int32_t num_nodes) {
  if (num_edges < 0 || num_nodes < 0) return 0;
  return build_layout(num_edges, num_nodes).total;
}

extern "C" gconv_status_t gconv_graph_build(const int64_t* edge_index, int64_t num_edges,
                                            int32_t num_nodes, int32_t normalize, void* arena_,
                                            size_t arena_bytes, gconv_graph_t* g,
                                            gconv_stream_t stream_) {
  return gconv_graph_build_ex(edge_index, num_edges, num_nodes, normalize, 0u, arena_, arena_bytes, g, stream_);
}

extern "C" gconv_status_t gconv_graph_build_ex(const int64_t* edge_index, int64_t num_edges,
                                               int32_t num_nodes, int32_t normalize, uint32_t flags,
                                               void* arena_, size_t arena_bytes, gconv_graph_t* g,
                                               gconv_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (num_edges < 0 || num_nodes < 0 || !arena_ || !g) return GCONV_E_INVALID_ARG;
  if (num_edges > 0 && !edge_index) return GCONV_E_INVALID_ARG;
  if (num_edges >= (static_cast<int64_t>(1) << 31) - 1) return GCONV_E_UNSUPPORTED;
  if (!aligned16(arena_)) return GCONV_E_ALIGNMENT;
  const BuildLayout L = build_layout(num_edges, num_nodes);
  if (arena_bytes < L.total) return GCONV_E_WORKSPACE;
  char* arena = static_cast<char*>(arena_);
  const int32_t n = num_nodes;
  const int64_t E = num_edges;

  auto slots = [&](const CsrSlots& c) {
    return CsrOut{reinterpret_cast<int32_t*>(arena + c.rowptr), reinterpret_cast<int32_t*>(arena + c.col),
                  reinterpret_cast<int32_t*>(arena + c.long_rows),
                  reinterpret_cast<int32_t*>(arena + c.long_seg_base),
                  reinterpret_cast<int32_t*>(arena + c.seg_long),
                  reinterpret_cast<uint32_t*>(arena + c.long_sync)};
  };
  struct TileOut { gconv_tile_t *recs, *tiles; int32_t *lcol, *halo; };
  auto tslots = [&](const CsrSlots& c) {
    return TileOut{reinterpret_cast<gconv_tile_t*>(arena + c.recs), reinterpret_cast<gconv_tile_t*>(arena + c.tiles),
                   reinterpret_cast<int32_t*>(arena + c.lcol), reinterpret_cast<int32_t*>(arena + c.halo)};
  };
  const CsrOut of = slots(L.f), ob = slots(L.b);
  const TileOut tf = tslots(L.f), tb = tslots(L.b);
  float* dis = normalize ? reinterpret_cast<float*>(arena + L.dis) : nullptr;
  int32_t* tile_ptr = reinterpret_cast<int32_t*>(arena + L.tile_ptr);
  int32_t* info = reinterpret_cast<int32_t*>(arena + L.info);
  // info words: [0..8] flags / maxima (enum above); forward counters at 16.., transposed at 24..
  const int FWD = 16, BWD = 24;
  const TileInfoSlots is{INFO_NOTILE, INFO_TILEMAX, INFO_TILEEDGES};
  const int slack = tile_slack(L.T, n);

  auto fill = [&](gconv_csr_t& c, const CsrOut& o, const int32_t* cnt) {
    c.num_nodes = n; c.nnz = cnt[0];
    c.rowptr = o.rowptr; c.col = o.col;
    c.n_long = cnt[1]; c.n_seg = cnt[2];
    c.long_rows = o.long_rows; c.long_seg_base = o.long_seg_base; c.seg_long = o.seg_long;
    c.long_sync = o.long_sync;
    c.near_edges = cnt[4];   // (radix builder: build_csr_kernel; the one-launch builders leave the word at 0)
  };
  auto fill_tiles = [&](const int32_t* h, bool symmetric) {
    g->tile_ptr = tile_ptr;
    g->tiles_fwd = gconv_tiling_t{tf.tiles, tf.lcol, tf.halo};
    g->tiles_bwd = symmetric ? g->tiles_fwd : gconv_tiling_t{tb.tiles, tb.lcol, tb.halo};
    g->n_tiles = (h[INFO_NOTILE] || n == 0) ? 0 : L.ntiles;
    g->tile_nodes_max = h[INFO_TILEMAX];
    g->tile_edges_max = h[INFO_TILEEDGES];
  };
  CountState* cs_small = small_build_ok(E, n) ? usable_state(false) : nullptr;
  if (cs_small) {   // ONE launch, counters straight into pinned memory (see small_graph_build_kernel)
    CountState* cs = cs_small;
    std::lock_guard<std::mutex> lock(cs->mu);
    cs->arm();
    SmallBuildArgs a{};
    a.ei = edge_index; a.E = static_cast<int32_t>(E); a.n = n; a.normalize = normalize ? 1 : 0;
    a.shift = index_bits(n); a.ntiles = L.ntiles; a.T = L.T; a.slack = slack;
    const int threads = 1024;
    a.nsort = threads;
    while (a.nsort < E) a.nsort <<= 1;
    a.rowptr_f = of.rowptr; a.col_f = of.col; a.long_rows_f = of.long_rows; a.long_seg_base_f = of.long_seg_base;
    a.seg_long_f = of.seg_long; a.long_sync_f = of.long_sync;
    a.rowptr_b = ob.rowptr; a.col_b = ob.col; a.long_rows_b = ob.long_rows; a.long_seg_base_b = ob.long_seg_base;
    a.seg_long_b = ob.seg_long; a.long_sync_b = ob.long_sync;
    a.dis = dis; a.tile_ptr = tile_ptr; a.info = cs->host_info_dev;
    a.recs_f = tf.recs; a.recs_b = tb.recs; a.tiles_f = tf.tiles; a.tiles_b = tb.tiles;
    a.lcol_f = tf.lcol; a.lcol_b = tb.lcol; a.halo_f = tf.halo; a.halo_b = tb.halo;
    const size_t lds = (static_cast<size_t>(a.nsort) + (n + 1) + 2 * static_cast<size_t>(n) + threads +
                        2 * L.ntiles + 1 + 64) * sizeof(int32_t);
#define GCONV_SMALL_BUILD(NR_, T_)                                                                     \
  do {                                                                                                 \
    if (lds > 48 * 1024)                                                                               \
      GCONV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&small_graph_build_kernel<NR_, T_>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds))); \
    small_graph_build_kernel<NR_, T_><<<1, T_, lds, stream>>>(a);                                      \
  } while (0)
    switch (a.nsort / 1024) {
      case 1: GCONV_SMALL_BUILD(1, 1024); break;
      case 2: GCONV_SMALL_BUILD(2, 1024); break;
      default: GCONV_SMALL_BUILD(4, 1024); break;
    }
#undef GCONV_SMALL_BUILD
    GCONV_LAUNCH_CHECK();
    int32_t h[64];
    if (!cs->wait_mark(stream, h)) {
      GCONV_HIP(hipStreamSynchronize(stream));
      return static_cast<gconv_status_t>(hipErrorUnknown);   // ended without reporting: cannot happen short of a fault
    }
    if (h[INFO_BAD]) {
      g->bad_index = 1;
      return GCONV_E_BAD_INDEX;
    }
    fill(g->fwd, of, h + 16);
    g->max_degree = h[16 + 3];
    g->symmetric = h[INFO_ASYM] ? 0 : 1;
    if (g->symmetric) g->bwd = g->fwd; else fill(g->bwd, ob, h + 24);
    g->dis = dis;
    fill_tiles(h, g->symmetric != 0);
    g->normalize = normalize ? 1 : 0;
    g->bad_index = 0;
    return GCONV_OK;
  }

  const int T = 256;
  int32_t h[64];
  // Which one-launch builder this device and this call can take (count_plan): the wide grid in-stream when
  // every workgroup can be resident at once, the narrow one as three barrier-free launches next to other
  // kernels (GCONV_BUILD_BACKGROUND), none on a device too small for either -> the radix pipeline.
  const CountPlan cp = counting_build_ok(E, n) ? count_plan((flags & GCONV_BUILD_BACKGROUND) != 0, n) : CountPlan{0, false, false};
  CountState* cs = cp.grid > 0 ? usable_state(true) : nullptr;
  if (cs) {   // one launch (see counting_build_kernel)
    std::lock_guard<std::mutex> lock(cs->mu);
    if (cs->broken && cs->ensure(true) != GCONV_OK) return GCONV_E_WORKSPACE;   // (re-clearing after a give-up)
    CountArgs a{};
    a.ei = edge_index; a.E = E; a.n = n; a.normalize = normalize ? 1 : 0;
    a.state = cs->state;
    a.ell = reinterpret_cast<int32_t*>(arena + L.ell);
    a.rowptr = of.rowptr; a.col = of.col; a.dis = dis;
    a.tile_ptr = tile_ptr; a.lcol = tf.lcol; a.slots = tf.halo; a.recs = tf.recs;
    a.host_info = cs->host_info_dev;
    a.ntiles = L.ntiles; a.T = L.T; a.slack = slack; a.fwd = FWD; a.is = is;
    cs->arm();
    bool launched = true;
    if (cp.split) {          // a launch per phase: nothing waits inside a kernel
      for (int ph = 1; ph <= 4; ph <<= 1) {
        a.phases = ph;
        if (cp.grid == kCountGrid) counting_build_kernel<kCountGrid, GCONV_COUNT_SPLIT_COH != 0><<<kCountGrid, kCountThreads, 0, stream>>>(a);
        else counting_build_kernel<kCountGridNarrow, GCONV_COUNT_SPLIT_COH != 0><<<kCountGridNarrow, kCountThreads, 0, stream>>>(a);
      }
    } else {
      a.phases = 7;
      if (cp.cooperative) {  // co-residency guaranteed by the runtime, or an immediate launch error
        void* args[] = {&a};
        const void* fn = cp.grid == kCountGrid ? reinterpret_cast<const void*>(&counting_build_kernel<kCountGrid>)
                                               : reinterpret_cast<const void*>(&counting_build_kernel<kCountGridNarrow>);
        if (hipLaunchCooperativeKernel(fn, dim3(cp.grid), dim3(kCountThreads), args, 0, stream) != hipSuccess) {
          (void)hipGetLastError();   // too large for this device right now: the radix pipeline builds the plan
          launched = false;
        }
      } else if (cp.grid == kCountGrid) {
        counting_build_kernel<kCountGrid><<<kCountGrid, kCountThreads, 0, stream>>>(a);
      } else {
        counting_build_kernel<kCountGridNarrow><<<kCountGridNarrow, kCountThreads, 0, stream>>>(a);
      }
    }
    if (launched) GCONV_LAUNCH_CHECK();
    if (!launched) {
    } else if (!cs->wait_mark(stream, h)) {
      GCONV_HIP(hipStreamSynchronize(stream));   // (a fault surfaces here)
      cs->broken = true;                       // gave up at a barrier: the radix pipeline below does the build
    } else {
#if GCONV_COUNT_STAMPS
      for (int i = 1; i <= 6; ++i) fprintf(stderr, "%d:%.2f ", i, (h[32 + i] - h[32 + i - 1]) * 0.01);
      fprintf(stderr, "us\n");
#endif
      if (h[INFO_BAD]) {
        g->bad_index = 1;
        return GCONV_E_BAD_INDEX;
      }
      if (!h[INFO_ASYM] && h[FWD + 3] <= kEllWidth) {   // symmetric, every row fitted its slots: done
        fill(g->fwd, of, h + FWD);
        g->max_degree = h[FWD + 3];
        g->symmetric = 1;
        g->bwd = g->fwd;
        g->dis = dis;
        fill_tiles(h, true);
        if (g->n_tiles > 0) {
          gconv_status_t ts = sort_tiles(L, arena, tf.recs, nullptr, tf.tiles, tb.tiles, g, stream);
          if (ts != GCONV_OK) return ts;
        }
        g->normalize = normalize ? 1 : 0;
        g->bad_index = 0;
        return GCONV_OK;
      }
      // a row of more than kEllWidth edges or a directed list: the radix pipeline does the whole build again
    }
  }

  GCONV_HIP(hipMemsetAsync(info, 0, 64 * sizeof(int32_t), stream));
  const int b = index_bits(n);
  const bool k32 = 2 * b <= 32;
  gconv_status_t st =
      k32 ? build_direction<uint32_t>(edge_index, E, n, normalize, 0, b, arena, L, of, dis, info, FWD, stream)
          : build_direction<uint64_t>(edge_index, E, n, normalize, 0, b, arena, L, of, dis, info, FWD, stream);
  if (st != GCONV_OK) return st;

  if (E > 0) {
    sym_check_kernel<<<static_cast<unsigned>((E + T - 1) / T), T, 0, stream>>>(
        of.rowptr, of.col, n, info + FWD, info + INFO_ASYM);
    GCONV_LAUNCH_CHECK();
  }
  if (n > 0) {   // tiles of the forward CSR: cuts, halos, local columns (one wave per tile)
    tile_build_kernel<true><<<(L.ntiles + 3) / 4, 256, 0, stream>>>(of.rowptr, of.col, dis, n, L.ntiles, L.T, slack,
                                                                    tile_ptr, tf.lcol, tf.halo, tf.recs, info, is);
    GCONV_LAUNCH_CHECK();
  }
  GCONV_HIP(hipMemcpyAsync(h, info, sizeof(h), hipMemcpyDeviceToHost, stream));
  GCONV_HIP(hipStreamSynchronize(stream));
  if (h[INFO_BAD]) {
    g->bad_index = 1;
    return GCONV_E_BAD_INDEX;
  }

  fill(g->fwd, of, h + FWD);
  g->max_degree = h[FWD + 3];
  g->symmetric = h[INFO_ASYM] ? 0 : 1;
  if (g->symmetric) {
    g->bwd = g->fwd;
  } else {  // rare path (directed edge lists): second sort for S^T, its tiles on the same cuts, second sync
    st = k32 ? build_direction<uint32_t>(edge_index, E, n, normalize, 1, b, arena, L, ob, nullptr, info, BWD, stream)
             : build_direction<uint64_t>(edge_index, E, n, normalize, 1, b, arena, L, ob, nullptr, info, BWD, stream);
    if (st != GCONV_OK) return st;
    if (n > 0) {
      tile_build_kernel<false><<<(L.ntiles + 3) / 4, 256, 0, stream>>>(ob.rowptr, ob.col, dis, n, L.ntiles, L.T, slack,
                                                                       tile_ptr, tb.lcol, tb.halo, tb.recs, info, is);
      GCONV_LAUNCH_CHECK();
    }
    GCONV_HIP(hipMemcpyAsync(h, info, sizeof(h), hipMemcpyDeviceToHost, stream));
    GCONV_HIP(hipStreamSynchronize(stream));
    fill(g->bwd, ob, h + BWD);
  }
  g->dis = dis;
  fill_tiles(h, g->symmetric != 0);
  if (g->n_tiles > 0) {
    st = sort_tiles(L, arena, tf.recs, g->symmetric ? nullptr : tb.recs, tf.tiles, tb.tiles, g, stream);
    if (st != GCONV_OK) return st;
  }
  g->normalize = normalize ? 1 : 0;
  g->bad_index = 0;
  return GCONV_OK;
}
