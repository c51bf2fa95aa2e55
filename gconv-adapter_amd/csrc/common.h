// Shared device/host helpers for the gfx950 GConv-Adapter kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/gconv_adapter.h"

#define GCONV_HIP(expr)                                         \
  do {                                                          \
    hipError_t _e = (expr);                                     \
    if (_e != hipSuccess) return static_cast<gconv_status_t>(_e); \
  } while (0)

#define GCONV_LAUNCH_CHECK() GCONV_HIP(hipGetLastError())

namespace gconv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
// The tile-plan FORMAT constant: a graph plan is cut for 256 persistent workgroups (tile_nominal below, the
// node-order rule of the tile list) on every device, so plans do not depend on where they were built. What a
// LAUNCH sizes by is the current device's compute-unit count, device_cus() — 256 on MI355X.
constexpr int kNumCU = 256;

// Compute units of the calling thread's current device (hipDeviceAttributeMultiprocessorCount, cached per
// device; kNumCU when the query fails). Grids of the persistent / capped launches come from here.
inline int device_cus() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return kNumCU;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n > 0) return n;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    n = kNumCU;
  }
  cache[dev].store(n, std::memory_order_relaxed);
  return n;
}

// What a persistent launch sizes its grid by: the device's CUs, never more than the plan format's 256 (the
// backward workspaces hold one partial row per workgroup, sized for at most kNumCU of them).
inline int launch_cus() { const int n = device_cus(); return n < kNumCU ? n : kNumCU; }

// "Once per device" for per-kernel attributes (hipFuncSetAttribute is a per-device setting; a process-wide
// `static bool` would opt the kernel in on the first device only — round-4 advisor finding). Racing threads
// may both run the body: it is idempotent.
struct DeviceOnce {
  std::atomic<uint64_t> done{0};
  bool need() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;   // unknown device: always set
    return (done.load(std::memory_order_relaxed) & (1ull << d)) == 0;
  }
  void mark() {
    int d = 0;
    if (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) done.fetch_or(1ull << d, std::memory_order_relaxed);
  }
};

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Nominal nodes per tile of an n-node graph (include/gconv_adapter.h, "Tiles"): about n / 256, so
// that every one of the 256 persistent workgroups gets ONE tile; never below GCONV_TILE_NODES (one
// MFMA block of rows) and, beyond 256 x GCONV_TILE_NODES_MAX nodes, n / (256 k) for the smallest k
// that keeps it under that maximum (k tiles of equal size per workgroup). With one tile per
// workgroup the largest tile IS the kernel's length, so cuts stay within T / 4 of their nominal
// position (and go through a molecule if there is no boundary that close); with several, the
// workgroups even out by dealing themselves tiles largest first, and a cut takes the nearest
// edge-free boundary up to a whole tile (at most 64 nodes) below — halos are then rare. Round 2 used 32 (64 from 16384 nodes on) and cut only at
// edge-free boundaries; libraries built at 32 / 48 / 64 / 96 / 128 were compared then: with the
// largest molecule as one tile the 512-molecule shard did not care (46.7 .. 47.6 us), larger
// batches gained from fewer chains per workgroup (4096 molecules: 181 -> 162 us at 64).
// A cut may lie up to tile_slack(T) nodes below its nominal position (on an edge-free boundary).
// Measured this round (tools/lab/tile_sweep.sh, 5 adapters' worth of one forward + backward):
// 512 molecules (one tile per workgroup) 47.0 us with round 2's tiles -> 39.9 at T = N / 256, slack
// T / 4 (T / 2 .. T / 6: 39.2 .. 40.1); 4096 molecules 156.3 -> 154.7 at T = 101 (4 tiles per workgroup,
// slack T) against 170.7 at T = 58 / slack T / 4 (7 per workgroup, a third of them with a halo);
// 32 molecules 24.6 -> 23.6 at T = 16.
// graphs of at least this many nodes get gconv_csr_t::near_edges counted and may take the SpMM's XCD-contiguous rows
constexpr int kXcdRowsMinNodes = 1 << 15;
int tile_nominal(int n);                       // graph_prep.hip (reads GCONV_TILE_NODES_OVERRIDE / _MAX)
int tile_slack(int T, int n);                  // how far below its nominal position a cut may sit
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int padded_rank(int r) {
  if (r < 1 || r > 64) return 0;
  int f = 4;
  while (f < r) f <<= 1;
  return f;
}

__device__ __forceinline__ f32x4 ldg4(const float* p) {
  return *reinterpret_cast<const f32x4*>(p);
}
// streamed-once output rows (never re-read by the writing kernel): non-temporal, so that they do not
// push the rows that ARE re-read out of L2 (GCONV_STREAM_STORES_NT=0: plain stores, A/B)
#ifndef GCONV_STREAM_STORES_NT
#define GCONV_STREAM_STORES_NT 1
#endif
__device__ __forceinline__ void stg4_stream(float* p, f32x4 v) {
#if GCONV_STREAM_STORES_NT
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
  *reinterpret_cast<f32x4*>(p) = v;
#endif
}
__device__ __forceinline__ void stg4(float* p, f32x4 v) {
  *reinterpret_cast<f32x4*>(p) = v;
}

__device__ __forceinline__ float act_fwd(int act, float h) {
  if (act == GCONV_ACT_RELU) return h > 0.f ? h : 0.f;
  if (act == GCONV_ACT_SILU) return h / (1.f + expf(-h));
  return h;
}

// derivative of SiLU at pre-activation h
__device__ __forceinline__ float silu_grad(float h) {
  float sig = 1.f / (1.f + expf(-h));
  return sig * (1.f + h * (1.f - sig));
}

// Buffer-resource access (raw buffers, 32-bit byte offsets, hardware range check): a load
// whose offset is >= the descriptor's byte count returns 0 and such a store is dropped. The
// tile kernels build one descriptor per tile and let the range check handle ragged node
// blocks and column tails — no clamped addresses, no operand masks, no branches around
// stores, so every lane issues the same number of memory operations and the compiler can
// count vmcnt exactly. Descriptors must be built from wave-uniform values only.
typedef int i32x4 __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr int kBufOob = 0x40000000;  // added to an offset to push it out of any tile's range

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, static_cast<int>(bytes), 0x00020000);
}
__device__ __forceinline__ f32x4 buf_ld4(rsrc_t r, int byte_off) {
  const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
  return __builtin_bit_cast(f32x4, v);
}
// Cache policy of the streamed-once accesses of the tile kernels (aux bit 1 = nt): the output
// rows (out, dX) are written once and never re-read by the kernel; written with the default
// policy they push the rows that ARE re-read (X between phases 1 and 4, G likewise) and the
// small hot arrays out of L2. nt stores: HIV-512 forward 17.2 -> 16.0 us, HIV-4096 backward
// 129 -> 114 us.
#ifndef GCONV_STORE_AUX
#define GCONV_STORE_AUX 2
#endif
// (nt on the rows' final READS was tried as well — backward 32.9 -> 36.4 us — and stays off)
#ifndef GCONV_LAST_USE_LOAD_AUX
#define GCONV_LAST_USE_LOAD_AUX 0
#endif
__device__ __forceinline__ f32x4 buf_ld4_last(rsrc_t r, int byte_off) {   // a row's final read
  const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, GCONV_LAST_USE_LOAD_AUX);
  return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ void buf_st4(rsrc_t r, int byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, byte_off, 0, GCONV_STORE_AUX);
}
// the same with the policy as a template argument (the fused tile kernels choose per launch: their phase-4 loops
// exist twice): NT = non-temporal (aux bit 1), else the default policy
template <bool NT>
__device__ __forceinline__ void buf_st4_p(rsrc_t r, int byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, byte_off, 0, NT ? 2 : 0);
}
template <bool NT>
__device__ __forceinline__ void buf_st1_p(rsrc_t r, int byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, byte_off, 0, NT ? 2 : 0);
}
__device__ __forceinline__ float buf_ld1(rsrc_t r, int byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ __forceinline__ void buf_st1(rsrc_t r, int byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, byte_off, 0, GCONV_STORE_AUX);
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// 16 independent 4x4 outer-product accumulations (v_mfma_f32_4x4x1_16b_f32), exact fp32:
//   D_b[i][j] += A_b[i] * B_b[j] for the 16 blocks b.
//   A: lane l holds A_(l/4)[l%4];  B: lane l holds B_(l/4)[l%4];  D: lane l, reg v = D_(l/4)[v][l%4]
// (layout verified on gfx950 by tools/mfma4_probe.hip). With the same A in every block it
// multiplies a 4-vector by 64 per-lane scalars: M = 4 wastes nothing at rank 4 or 8, where the
// 16x16x4 form pads the rank dimension to 16.
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
#ifdef GCONV_LAB_NO_MFMA   // lab build (wrong results): one fma keeps the operands live, the matrix pipe stays idle —
  c[0] += a * b;           // the upper bound of what ANY cheaper matrix arithmetic could return (LAB.md, round 5)
  return c;
#else
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
#endif
}

// Wave index inside the workgroup as a provably wave-uniform (SGPR) value. Without the
// readfirstlane hipcc treats `threadIdx.x >> 6` as divergent, turns every branch that
// depends on it into an exec-masked one and fences each load under it with vmcnt(0).
__device__ __forceinline__ int wave_id() {
  return __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
}

// D(16x16) += A(16x4) * B(4x16), exact fp32.
//   A: lane l holds A[l & 15][l >> 4];  B: lane l holds B[l >> 4][l & 15]
//   D: lane l, reg m holds D[4 * (l >> 4) + m][l & 15]
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
#ifdef GCONV_LAB_NO_MFMA
  c[0] += a * b;
  return c;
#else
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

}  // namespace gconv
