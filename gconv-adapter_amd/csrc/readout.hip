// The end of the fine-tune step's forward pass (what sits between the last adapter and
// `loss.backward()`), as three launches instead of ~25 of torch's:
//
//   graph readout    `self.pool(h, batch)` with global_mean_pool / global_add_pool
//                    /root/reference/src/models/inductive/gnn.py:187-195 (PyG: scatter over the batch vector);
//                    one workgroup per graph, the graph's rows found by a 64-ary search of the SORTED batch
//                    vector (PyG's collation emits it sorted), columns summed in row order — no atomics
//   its backward     d h[i] = d pooled[batch[i]] (/ count): one wave per node row
//   masked loss      /root/reference/scripts/finetune_inductive_learning.py:55-63 — targets in fp64, NaN = no
//                    label; criterion(valid_output.double(), valid_target) summed and divided by the number of
//                    valid targets; criterion = BCEWithLogitsLoss(reduction='none') (classification) or
//                    MSELoss(reduction='none') (regression: ESOL). One workgroup, two passes over the G x T
//                    logits (count + sum in a fixed order, then the gradient (sigmoid(x) - t) / count or
//                    2 (x - t) / count, stored in fp32 beside the loss): the backward pass is a scaling of it.
// On a 512-molecule shard each of torch's small kernels costs 4 - 8 us of device time and as much of host time:
// index_add_ 24 us, its zero fill, the division, the fp64 casts, the boolean-mask gather and its sort-based
// backward, log-sigmoid, the reductions — ~100 us of a 1.2 ms step.
#include <cmath>

#include "common.h"

namespace gconv {

// first index i in [0, n) with batch[i] >= g (n if none), by all 64 lanes of one wave: every round the lanes
// probe 64 evenly spaced positions of the remaining range (3 rounds up to 262,144 nodes)
__device__ __forceinline__ int lower_bound64(const int64_t* __restrict__ batch, int n, int64_t g, int lane) {
  // invariant: batch[lo - 1] < g (or lo == 0) and batch[hi] >= g (or hi == n): the answer is in [lo, hi]
  int lo = 0, hi = n;
  while (hi > lo) {
    const int span = hi - lo, step = (span + 63) / 64;   // pieces of `step` positions, one per lane
    const int pos = lo + min(span, (lane + 1) * step) - 1;   // the last position of this lane's piece
    const bool ge = lane * step < span && batch[pos] >= g;   // (lanes past the range: no piece)
    const unsigned long long m = __ballot(ge);
    if (m == 0) return hi;                               // every piece ends below g: so does position hi - 1
    const int first = __ffsll(static_cast<long long>(m)) - 1;
    // the piece of lane `first` ends at or above g, the one before it ends below: the answer is inside it
    hi = lo + min(span, (first + 1) * step) - 1;
    lo = lo + first * step;
  }
  return lo;
}

// out[g, :] = sum (or mean) of x[i, :] over the graph's rows; counts[g] = number of rows
template <int NP>
__global__ __launch_bounds__(256) void segment_pool_kernel(const float* __restrict__ x,
                                                           const int64_t* __restrict__ batch, int n, int d,
                                                           int mean, float* __restrict__ out,
                                                           int32_t* __restrict__ counts) {
  __shared__ int seg[2];
  __shared__ int badflag;
  __shared__ f32x4 red[3][64 * NP];
  const int g = blockIdx.x, lane = threadIdx.x & 63, wave = wave_id();
  if (threadIdx.x == 0) badflag = 0;
  if (wave < 2) {
    const int b = lower_bound64(batch, n, static_cast<int64_t>(g) + wave, lane);
    if (lane == 0) seg[wave] = b;
  }
  __syncthreads();
  const int r0 = seg[0], r1 = seg[1];
  const int nq = d >> 2;
  f32x4 a[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) a[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  // wave w: rows r0 + w, r0 + w + 4, ... added in that order; kU of them are requested before the first is added
  // (a molecule of 222 atoms would otherwise be 56 dependent round trips for its workgroup — and the kernel)
  constexpr int kU = 8;
  // `batch` must be sorted (PyG's collation order): the segment [r0, r1) found by the search then holds exactly the
  // rows of graph g. Checked on the way — every row this wave adds must carry g; graph 0 / the last graph also that
  // no row lies in front of / behind all segments — at the price of one broadcast 8-byte load per row: a violation
  // turns the graph's output row into NaN (the loss shows it without any host synchronisation) and raises the
  // flag word counts[num_graphs] for `validate=True` (round-4 advisor finding: silently wrong sums).
  bool bad = false;
  for (int row = r0 + wave; row < r1; row += 4 * kU) {
    f32x4 v[kU][NP];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (row + 4 * u < r1) bad |= batch[row + 4 * u] != static_cast<int64_t>(g);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int q = lane + 64 * p;
        v[u][p] = q < nq && row + 4 * u < r1 ? ldg4(x + static_cast<size_t>(row + 4 * u) * d + 4 * q)
                                            : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u)
#pragma unroll
      for (int p = 0; p < NP; ++p) a[p] += v[u][p];
  }
  if (g == 0) bad |= r0 != 0;                                    // rows in front of every segment (ids < 0, or disorder)
  if (g == static_cast<int>(gridDim.x) - 1) bad |= r1 != n;      // rows behind every segment (ids >= size, or disorder)
  if (bad) badflag = 1;                                          // (benign race: every writer stores 1)
  if (wave > 0) {
#pragma unroll
    for (int p = 0; p < NP; ++p) red[wave - 1][lane + 64 * p] = a[p];
  }
  __syncthreads();
  if (wave == 0) {
    const bool poisoned = badflag != 0;
    const float scale = poisoned ? __builtin_nanf("") : (mean && r1 > r0 ? 1.f / static_cast<float>(r1 - r0) : 1.f);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int q = lane + 64 * p;
      f32x4 t = a[p];
      for (int w = 0; w < 3; ++w) t += red[w][lane + 64 * p];   // waves added in wave order
      if (q < nq) stg4(out + static_cast<size_t>(g) * d + 4 * q, t * scale);
    }
    if (lane == 0 && counts) {
      counts[g] = r1 - r0;
      if (poisoned) counts[gridDim.x] = 1;   // the flag word behind the counts (zeroed by the launcher)
    }
  }
}

// grad_x[i, :] = grad_out[batch[i], :] (/ counts[batch[i]])
template <int NP>
__global__ __launch_bounds__(256) void segment_pool_bwd_kernel(const float* __restrict__ go,
                                                               const int64_t* __restrict__ batch,
                                                               const int32_t* __restrict__ counts, int n, int d,
                                                               int num_graphs, int mean,
                                                               float* __restrict__ gx) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave_id();
  if (row >= n) return;
  const int64_t g = batch[row];
  const bool ok = g >= 0 && g < num_graphs;              // (a node outside [0, size): no graph, no gradient)
  const float scale = ok && mean ? 1.f / static_cast<float>(max(counts[g], 1)) : 1.f;
  const int nq = d >> 2;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int q = lane + 64 * p;
    if (q < nq) {
      const f32x4 v = ok ? ldg4(go + static_cast<size_t>(g) * d + 4 * q) * scale : f32x4{0.f, 0.f, 0.f, 0.f};
      stg4(gx + static_cast<size_t>(row) * d + 4 * q, v);
    }
  }
}

// kind 0: BCE with logits, kind 1: squared error; fp64 arithmetic on fp32 logits (the reference casts the valid
// outputs to double). out[0] = sum / count, out[1] = count (as a double); grad[i] = d out[0] / d logits[i]
__global__ __launch_bounds__(1024) void masked_loss_kernel(const float* __restrict__ logits,
                                                           const double* __restrict__ target, long long m,
                                                           int kind, double* __restrict__ out,
                                                           float* __restrict__ grad) {
  __shared__ double red_s[16];
  __shared__ long long red_c[16];
  __shared__ double total[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s = 0.0;
  long long c = 0;
  for (long long i = tid; i < m; i += 1024) {            // thread t: elements t, t + 1024, ... (fixed order)
    const double t = target[i];
    if (t != t) continue;                                 // NaN: no label
    const double x = static_cast<double>(logits[i]);
    if (kind == 0) s += fmax(x, 0.0) - x * t + log1p(exp(-fabs(x)));
    else s += (x - t) * (x - t);
    ++c;
  }
  for (int off = 32; off > 0; off >>= 1) {                // (xor butterfly: every lane ends with the wave's sum)
    s += __shfl_xor(s, off);
    c += __shfl_xor(c, off);
  }
  if (lane == 0) { red_s[wave] = s; red_c[wave] = c; }
  __syncthreads();
  if (tid == 0) {
    double ts = 0.0;
    long long tc = 0;
    for (int w = 0; w < 16; ++w) { ts += red_s[w]; tc += red_c[w]; }
    total[0] = tc > 0 ? ts / static_cast<double>(tc) : 0.0;
    total[1] = static_cast<double>(tc);
    out[0] = total[0];
    out[1] = total[1];
  }
  __syncthreads();
  const double cnt = total[1];
  const double inv = cnt > 0.0 ? 1.0 / cnt : 0.0;
  for (long long i = tid; i < m; i += 1024) {
    const double t = target[i];
    double gi = 0.0;
    if (t == t) {
      const double x = static_cast<double>(logits[i]);
      gi = kind == 0 ? (1.0 / (1.0 + exp(-x)) - t) * inv : 2.0 * (x - t) * inv;
    }
    grad[i] = static_cast<float>(gi);
  }
}

// The same loss for MANY logits (a multi-task head: ToxCast, 617 tasks x 4096 graphs = 2.5 M; round-4 advisor
// finding: one workgroup walked them all, twice, in fp64). Stage 1: kLossBlocks workgroups, block b sums the elements
// b * 1024 + t, (b + kLossBlocks) * 1024 + t, ... (fixed assignment and order), writes {sum, count} to the workspace
// with write-through stores; the workgroup whose ticket is last adds the partials in block order, writes the loss and
// the count, and resets the ticket (fixed order: the same bits every run). Stage 2 (second launch): the gradient.
constexpr int kLossBlocks = 256;
__global__ __launch_bounds__(1024) void masked_loss_partial_kernel(const float* __restrict__ logits,
                                                                   const double* __restrict__ target, long long m,
                                                                   int kind, double* __restrict__ out,
                                                                   double* __restrict__ part, unsigned* __restrict__ ticket) {
  __shared__ double red_s[16];
  __shared__ long long red_c[16];
  __shared__ int last_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s = 0.0;
  long long c = 0;
  for (long long i = static_cast<long long>(blockIdx.x) * 1024 + tid; i < m; i += static_cast<long long>(gridDim.x) * 1024) {
    const double t = target[i];
    if (t != t) continue;
    const double x = static_cast<double>(logits[i]);
    if (kind == 0) s += fmax(x, 0.0) - x * t + log1p(exp(-fabs(x)));
    else s += (x - t) * (x - t);
    ++c;
  }
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off);
    c += __shfl_xor(c, off);
  }
  if (lane == 0) { red_s[wave] = s; red_c[wave] = c; }
  __syncthreads();
  if (tid == 0) {
    double ts = 0.0;
    long long tc = 0;
    for (int w = 0; w < 16; ++w) { ts += red_s[w]; tc += red_c[w]; }
    __hip_atomic_store(part + 2 * blockIdx.x, ts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(part + 2 * blockIdx.x + 1, static_cast<double>(tc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_s = t == gridDim.x - 1;
  }
  __syncthreads();
  if (last_s && tid == 0) {
    double ts = 0.0, tc = 0.0;
    for (unsigned b = 0; b < gridDim.x; ++b) {
      ts += __hip_atomic_load(part + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tc += __hip_atomic_load(part + 2 * b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    out[0] = tc > 0.0 ? ts / tc : 0.0;
    out[1] = tc;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(256) void masked_loss_grad_kernel(const float* __restrict__ logits,
                                                               const double* __restrict__ target, long long m, int kind,
                                                               const double* __restrict__ out, float* __restrict__ grad) {
  const double cnt = out[1];
  const double inv = cnt > 0.0 ? 1.0 / cnt : 0.0;
  for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < m; i += static_cast<long long>(gridDim.x) * 256) {
    const double t = target[i];
    double gi = 0.0;
    if (t == t) {
      const double x = static_cast<double>(logits[i]);
      gi = kind == 0 ? (1.0 / (1.0 + exp(-x)) - t) * inv : 2.0 * (x - t) * inv;
    }
    grad[i] = static_cast<float>(gi);
  }
}

}  // namespace gconv

using namespace gconv;

#define GCONV_POOL_DISPATCH(KERNEL_CALL)                  \
  switch (ceil_div(hidden, 256)) {                        \
    case 1: { constexpr int NP = 1; KERNEL_CALL; } break; \
    case 2: { constexpr int NP = 2; KERNEL_CALL; } break; \
    case 3: { constexpr int NP = 3; KERNEL_CALL; } break; \
    default: { constexpr int NP = 4; KERNEL_CALL; } break; \
  }

static gconv_status_t pool_check(int32_t num_nodes, int32_t hidden, int32_t num_graphs) {
  if (num_nodes < 0 || num_graphs < 0 || hidden <= 0) return GCONV_E_INVALID_ARG;
  if (hidden % 4 != 0 || hidden > 1024) return GCONV_E_UNSUPPORTED;
  return GCONV_OK;
}

extern "C" gconv_status_t gconv_segment_pool_f32(const float* x, const int64_t* batch, int32_t num_nodes,
                                                 int32_t hidden, int32_t num_graphs, int32_t mean, float* out,
                                                 int32_t* counts, gconv_stream_t stream_) {
  const gconv_status_t st = pool_check(num_nodes, hidden, num_graphs);
  if (st != GCONV_OK) return st;
  if (num_graphs == 0) return GCONV_OK;
  if (!out || (num_nodes > 0 && (!x || !batch))) return GCONV_E_INVALID_ARG;
  if (!aligned16(x) || !aligned16(out)) return GCONV_E_ALIGNMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  // counts[num_graphs] = "batch is not sorted / holds ids outside [0, num_graphs)" flag (include/gconv_adapter.h)
  if (counts) GCONV_HIP(hipMemsetAsync(counts + num_graphs, 0, sizeof(int32_t), stream));
  GCONV_POOL_DISPATCH((segment_pool_kernel<NP><<<num_graphs, 256, 0, stream>>>(x, batch, num_nodes, hidden, mean,
                                                                               out, counts)))
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

extern "C" gconv_status_t gconv_segment_pool_bwd_f32(const float* grad_out, const int64_t* batch,
                                                     const int32_t* counts, int32_t num_nodes, int32_t hidden,
                                                     int32_t num_graphs, int32_t mean, float* grad_x,
                                                     gconv_stream_t stream_) {
  const gconv_status_t st = pool_check(num_nodes, hidden, num_graphs);
  if (st != GCONV_OK) return st;
  if (num_nodes == 0) return GCONV_OK;
  if (!grad_out || !batch || !grad_x || (mean && !counts)) return GCONV_E_INVALID_ARG;
  if (!aligned16(grad_out) || !aligned16(grad_x)) return GCONV_E_ALIGNMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GCONV_POOL_DISPATCH((segment_pool_bwd_kernel<NP><<<ceil_div(num_nodes, 4), 256, 0, stream>>>(
      grad_out, batch, counts, num_nodes, hidden, num_graphs, mean, grad_x)))
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

extern "C" gconv_status_t gconv_masked_loss_f64(const float* logits, const double* target, int64_t count,
                                                int32_t kind, double* loss_and_count, float* grad_logits,
                                                gconv_stream_t stream_) {
  if (count < 0 || (kind != 0 && kind != 1) || !loss_and_count) return GCONV_E_INVALID_ARG;
  if (count > 0 && (!logits || !target || !grad_logits)) return GCONV_E_INVALID_ARG;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  masked_loss_kernel<<<1, 1024, 0, stream>>>(logits, target, count, kind, loss_and_count, grad_logits);
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

// (256 partial pairs + the ticket word; must read all zero in its LAST 64 bytes before the first call — the ticket
//  goes back to zero at the end of every call)
extern "C" size_t gconv_masked_loss_workspace_bytes(int64_t count) {
  (void)count;
  return static_cast<size_t>(kLossBlocks) * 2 * sizeof(double) + 64;
}

extern "C" gconv_status_t gconv_masked_loss_ws_f64(const float* logits, const double* target, int64_t count,
                                                   int32_t kind, double* loss_and_count, float* grad_logits,
                                                   void* workspace, size_t workspace_bytes, gconv_stream_t stream_) {
  if (count < 0 || (kind != 0 && kind != 1) || !loss_and_count) return GCONV_E_INVALID_ARG;
  if (count > 0 && (!logits || !target || !grad_logits)) return GCONV_E_INVALID_ARG;
  if (!workspace || workspace_bytes < gconv_masked_loss_workspace_bytes(count)) return GCONV_E_WORKSPACE;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  double* part = static_cast<double*>(workspace);
  unsigned* ticket = reinterpret_cast<unsigned*>(part + 2 * kLossBlocks);
  const int blocks = static_cast<int>(count / 1024 + 1 < kLossBlocks ? count / 1024 + 1 : kLossBlocks);
  masked_loss_partial_kernel<<<blocks, 1024, 0, stream>>>(logits, target, count, kind, loss_and_count, part, ticket);
  GCONV_LAUNCH_CHECK();
  if (count > 0) {
    const int gblocks = static_cast<int>((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
    masked_loss_grad_kernel<<<gblocks, 256, 0, stream>>>(logits, target, count, kind, loss_and_count, grad_logits);
    GCONV_LAUNCH_CHECK();
  }
  return GCONV_OK;
}
