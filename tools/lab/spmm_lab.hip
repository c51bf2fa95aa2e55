// Experimental SpMM variants for the S-arxiv roofline work (round 2). Not part of the product
// library: built by tools/lab/run_spmm_lab.py into tools/lab/spmm_lab.so and driven through ctypes.
//   v1: row kernel with an XCD-contiguous block -> row-block mapping
//   v2: column-phase blocked persistent kernel — a wave owns 8*S rows whose accumulators stay in
//       registers while the whole chip sweeps K column ranges in step, so that at any time every
//       gather of an XCD falls into one 21.8 MB / K slice of H (L2-resident)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr int kOob = 0x7ffffff0;

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, static_cast<int>(bytes), 0x00020000);
}
__device__ __forceinline__ f32x4 buf_ld4(rsrc_t r, int off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ int buf_ld1i(rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0); }
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)); }

constexpr int F = 32;
constexpr int LONG = 64;

// ---------------------------------------------------------------- v1: row kernel, optional remap
__global__ __launch_bounds__(256) void v1_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                 const float* __restrict__ dis, const float* __restrict__ H,
                                                 float* __restrict__ Y, int n, int remap) {
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int sub = lane & 7, grp = lane >> 3;
  int b = blockIdx.x;
  if (remap) {
    const int per = (gridDim.x + 7) >> 3;
    b = (b & 7) * per + (b >> 3);
  }
  const int row = (b * 4 + wave) * 8 + grp;
  const int rowc = min(row, n - 1);
  const int e0 = rowptr[rowc], e1 = rowptr[rowc + 1];
  const bool mine = row < n && (e1 - e0) <= LONG;
  const int deg = mine ? e1 - e0 : 0;
  int nbat = (deg + 7) >> 3;
  for (int off = 32; off > 0; off >>= 1) nbat = max(nbat, __shfl_xor(nbat, off));
  nbat = __builtin_amdgcn_readfirstlane(nbat);
  const int elast = max(e1 - 1, 0);
  int cn = col[min(e0 + sub, elast)];
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  for (int it = 0; it < nbat; ++it) {
    const int cc = cn;
    const int eb = e0 + 8 * it;
    cn = col[min(eb + 8 + sub, elast)];
    f32x4 h[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int c = __shfl(cc, grp * 8 + u);
      c = (eb + u < e0 + deg) ? c : rowc;
      h[u] = *reinterpret_cast<const f32x4*>(H + static_cast<size_t>(c) * F + 4 * sub);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) h[u] = (eb + u < e0 + deg) ? h[u] : f32x4{0, 0, 0, 0};
    acc0 += (h[0] + h[1]) + (h[2] + h[3]);
    acc1 += (h[4] + h[5]) + (h[6] + h[7]);
  }
  if (mine) {
    const size_t off = static_cast<size_t>(row) * F + 4 * sub;
    const float di = dis[row];
    f32x4 v = ((acc0 + acc1) + *reinterpret_cast<const f32x4*>(H + off)) * di;
    v *= di;
    *reinterpret_cast<f32x4*>(Y + off) = v;
  }
}

// ---------------------------------------------------------------- v2: column-phase blocked
template <int S, int K>
__global__ __launch_bounds__(256) void v2_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                 const float* __restrict__ dis, const float* __restrict__ H,
                                                 float* __restrict__ Y, int n, int nnz, int remap) {
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int sub = lane & 7, grp = lane >> 3;
  int b = blockIdx.x;
  if (remap) {
    const int per = (gridDim.x + 7) >> 3;
    b = (b & 7) * per + (b >> 3);
  }
  const int row0 = (b * 4 + wave) * (8 * S);
  const rsrc_t rH = make_rsrc(H, static_cast<uint32_t>(n) * F * 4u);
  const rsrc_t rC = make_rsrc(col, static_cast<uint32_t>(nnz) * 4u);

  int ecur[S], eend[S], cwin[S];
  f32x4 acc[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int row = row0 + s * 8 + grp;
    const int rc = min(row, n - 1);
    int e0 = rowptr[rc], e1 = rowptr[rc + 1];
    if (row >= n || e1 - e0 > LONG) e1 = e0;
    ecur[s] = e0;
    eend[s] = e1;
    cwin[s] = buf_ld1i(rC, ((e0 & ~7) + sub) * 4);
    acc[s] = f32x4{0, 0, 0, 0};
  }
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    const int hi = (k == K - 1) ? 0x7fffffff : static_cast<int>((static_cast<long long>(k + 1) * n) / K);
    for (;;) {
      int c[S];
      bool act[S];
      bool any = false;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        c[s] = __shfl(cwin[s], grp * 8 + (ecur[s] & 7));
        act[s] = ecur[s] < eend[s] && c[s] < hi;
        any |= act[s];
      }
      if (!__any(any)) break;
      f32x4 h[S];
#pragma unroll
      for (int s = 0; s < S; ++s) h[s] = buf_ld4(rH, act[s] ? c[s] * (F * 4) + sub * 16 : kOob);
#pragma unroll
      for (int s = 0; s < S; ++s) {
        ecur[s] += act[s] ? 1 : 0;
        const bool reload = act[s] && (ecur[s] & 7) == 0;
        const int w = buf_ld1i(rC, reload ? (ecur[s] + sub) * 4 : kOob);
        cwin[s] = reload ? w : cwin[s];
      }
#pragma unroll
      for (int s = 0; s < S; ++s) acc[s] += h[s];
    }
  }
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int row = row0 + s * 8 + grp;
    if (row < n) {
      const int e0 = rowptr[row], e1 = rowptr[row + 1];
      if (e1 - e0 <= LONG) {
        const size_t off = static_cast<size_t>(row) * F + 4 * sub;
        const float di = dis[row];
        f32x4 v = (acc[s] + *reinterpret_cast<const f32x4*>(H + off)) * di;
        v *= di;
        *reinterpret_cast<f32x4*>(Y + off) = v;
      }
    }
  }
}

// v1s: rows taken from a schedule (sched[i] = row handled by row slot i; slot -> block -> XCD)
__global__ __launch_bounds__(256) void v1s_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                  const float* __restrict__ dis, const float* __restrict__ H,
                                                  float* __restrict__ Y, const int* __restrict__ sched, int n) {
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int sub = lane & 7, grp = lane >> 3;
  const int slot = (blockIdx.x * 4 + wave) * 8 + grp;
  const int row = slot < n ? sched[slot] : -1;
  const int rowc = max(row, 0);
  const int e0 = rowptr[rowc], e1 = rowptr[rowc + 1];
  const bool mine = row >= 0 && (e1 - e0) <= LONG;
  const int deg = mine ? e1 - e0 : 0;
  int nbat = (deg + 7) >> 3;
  for (int off = 32; off > 0; off >>= 1) nbat = max(nbat, __shfl_xor(nbat, off));
  nbat = __builtin_amdgcn_readfirstlane(nbat);
  const int elast = max(e1 - 1, 0);
  int cn = col[min(e0 + sub, elast)];
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  for (int it = 0; it < nbat; ++it) {
    const int cc = cn;
    const int eb = e0 + 8 * it;
    cn = col[min(eb + 8 + sub, elast)];
    f32x4 h[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int c = __shfl(cc, grp * 8 + u);
      c = (eb + u < e0 + deg) ? c : rowc;
      h[u] = *reinterpret_cast<const f32x4*>(H + static_cast<size_t>(c) * F + 4 * sub);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) h[u] = (eb + u < e0 + deg) ? h[u] : f32x4{0, 0, 0, 0};
    acc0 += (h[0] + h[1]) + (h[2] + h[3]);
    acc1 += (h[4] + h[5]) + (h[6] + h[7]);
  }
  if (mine) {
    const size_t off = static_cast<size_t>(row) * F + 4 * sub;
    const float di = dis[row];
    f32x4 v = ((acc0 + acc1) + *reinterpret_cast<const f32x4*>(H + off)) * di;
    v *= di;
    *reinterpret_cast<f32x4*>(Y + off) = v;
  }
}

extern "C" int lab_v1s(const int* rowptr, const int* col, const float* dis, const float* H, float* Y,
                       const int* sched, int n, void* stream) {
  const int grid = (n + 31) / 32;
  v1s_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(rowptr, col, dis, H, Y, sched, n);
  return static_cast<int>(hipGetLastError());
}

extern "C" int lab_v1(const int* rowptr, const int* col, const float* dis, const float* H, float* Y, int n,
                      int remap, void* stream) {
  const int grid = (n + 31) / 32;
  v1_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(rowptr, col, dis, H, Y, n, remap);
  return static_cast<int>(hipGetLastError());
}

template <int S>
static int launch_v2(const int* rowptr, const int* col, const float* dis, const float* H, float* Y, int n, int nnz,
                     int K, int remap, hipStream_t st) {
  const int rows_per_block = 4 * 8 * S;
  int grid = (n + rows_per_block - 1) / rows_per_block;
  grid = (grid + 7) / 8 * 8;
  switch (K) {
    case 4: v2_kernel<S, 4><<<grid, 256, 0, st>>>(rowptr, col, dis, H, Y, n, nnz, remap); break;
    case 8: v2_kernel<S, 8><<<grid, 256, 0, st>>>(rowptr, col, dis, H, Y, n, nnz, remap); break;
    case 16: v2_kernel<S, 16><<<grid, 256, 0, st>>>(rowptr, col, dis, H, Y, n, nnz, remap); break;
    case 32: v2_kernel<S, 32><<<grid, 256, 0, st>>>(rowptr, col, dis, H, Y, n, nnz, remap); break;
    case 64: v2_kernel<S, 64><<<grid, 256, 0, st>>>(rowptr, col, dis, H, Y, n, nnz, remap); break;
    case 1: v2_kernel<S, 1><<<grid, 256, 0, st>>>(rowptr, col, dis, H, Y, n, nnz, remap); break;
    default: return -1;
  }
  return static_cast<int>(hipGetLastError());
}

extern "C" int lab_v2(const int* rowptr, const int* col, const float* dis, const float* H, float* Y, int n, int nnz,
                      int S, int K, int remap, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (S) {
    case 2: return launch_v2<2>(rowptr, col, dis, H, Y, n, nnz, K, remap, st);
    case 4: return launch_v2<4>(rowptr, col, dis, H, Y, n, nnz, K, remap, st);
    case 8: return launch_v2<8>(rowptr, col, dis, H, Y, n, nnz, K, remap, st);
    case 12: return launch_v2<12>(rowptr, col, dis, H, Y, n, nnz, K, remap, st);
    case 16: return launch_v2<16>(rowptr, col, dis, H, Y, n, nnz, K, remap, st);
  }
  return -1;
}
