// Fused tile-local adapter forward / backward for block-diagonal graphs (molecule batches).
//
// Graph prep cuts the node range into tiles of about N / 256 nodes (tiles.h): on a boundary
// between two molecules where there is one nearby, else through a component, and then the
// tile carries its two-hop halo — the rank-width rows of the nodes just outside that its own
// rows depend on are recomputed by the tile itself, so tiles exchange nothing. ONE workgroup
// runs the whole adapter for its tile — the three rank-width intermediates (H0', H2', P,
// resp. dP', dH1', T) never leave LDS, the grid-wide dependencies of the four-launch path
// become lds_barrier(), and X is read from HBM once (its second read, for the
// residual, comes out of L2 a few microseconds later). Phase 1 runs on own + halo rows (n0),
// phase 2 on own + 1-hop halo rows (n1), phases 3 / 4 on the own rows:
//
//   forward   phase 1  H0' = dis . (X_t Wd^T)                   MFMA, A layout, W_d frags in LDS
//             phase 2  H2' = dis . act(dis . (A0 H0' + H0') + bd)      LDS gathers
//             phase 3  P   = dis . (A0 H2' + H2')                      LDS gathers
//             phase 4  out = s (P Wu^T + bu + X_t)              MFMA, L layout
//   backward  phase 1  dP' = dis . s (G_t Wu) ; sum (G Wu) . P  MFMA, A layout
//             phase 2  dH1' = dis . (S^T dP) . act' ; dbd
//             phase 3  T = S^T dH1
//             phase 4  dX_t = T Wd + s G_t ; dWd^T, dWu^T, colsum(G), sum G.X   4x4x1 MFMA,
//                      one column per lane, T / P rank-major in LDS
//
// Same math, same epilogues and the same deterministic B5 finalize as the four-launch
// path (adapter.hip); replaces /root/reference/src/finetune/gconv_adapter.py:80-108 and
// its autograd backward for the batches the inductive script feeds it
// (scripts/finetune_inductive_learning.py:43-67).
#include <hip/hip_ext.h>

#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "fused_tile.h"
#include "tiles.h"

#ifndef GCONV_BWD_WAVES
#define GCONV_BWD_WAVES 2
#endif
// Workgroups per CU (lab knob): 2 = half-size workgroups (forward and backward:
// one wave per 64-column chunk, 4 at least), registers capped for 3 waves per SIMD, the backward
// with ONE (G, X) pair in flight, grid of 2 x CUs — so that two workgroups share a CU and the
// latency-bound phases of one run under the bandwidth-bound phase 4 of the other. The product
// build is 1 (measured, slower).
#ifndef GCONV_TILE_WGPC
#define GCONV_TILE_WGPC 1
#endif
// Wave priority of the tile kernels' waves when a workgroup walks several tiles (s_setprio at kernel entry): a
// graph-build wave of the prefetch worker that shares a SIMD with two backward waves then yields its issue slots to
// them. Round 5, whole 4096-molecule batch, fresh plan per step (tools/lab/r05_prio.sh): 0.8959 / 0.8858 -> 0.8601 /
// 0.8686 ms (one cached plan: 0.809 either way; the 512-molecule shard, one tile per workgroup, stays at 0).
#ifndef GCONV_TILE_PRIO
#define GCONV_TILE_PRIO 3
#endif
#ifndef GCONV_TILE_PRIO_MIN_PASSES   // (lab: 0 = every launch)
#define GCONV_TILE_PRIO_MIN_PASSES 1
#endif
#ifndef GCONV_TILE_STAGGER_DEFAULT
#define GCONV_TILE_STAGGER_DEFAULT 1
#endif

// Diagnostic build (-DGCONV_TILE_STAMPS): per-workgroup phase time stamps (100 MHz wall clock)
// into a side buffer that nothing else reads; never compiled into the product library.
#ifdef GCONV_TILE_STAMPS
#define TILE_STAMP(k) do { if (threadIdx.x == 0) a.stamps[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#define TILE_STAMP_NT(v) do { if (threadIdx.x == 0) a.stamps[blockIdx.x * 16 + 7] = (v); } while (0)
#else
#define TILE_STAMP(k)
#define TILE_STAMP_NT(v)
#endif

namespace gconv {

namespace {

__device__ __forceinline__ f32x4 zero4f() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// Workgroup barrier for LDS hand-offs only. hipcc's __syncthreads() is a workgroup-scope fence
// plus s_barrier and drains vmcnt as well: every global request in flight — the phase-4 tiles
// requested two phases early, a neighbour wave's refills — would have to land at each of the
// five barriers of a tile. Nothing in these kernels passes data between waves through global
// memory, so only LDS traffic (lgkmcnt) has to be complete.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int FP>
__device__ __forceinline__ int rank_k_(int s, int t) {
  return FP >= 16 ? 16 * (s >> 2) + 4 * t + (s & 3) : 4 * s + t;
}

// A fragments of an LDS rank-width tile Z[cap, FP] for local row `row` (clamped by caller)
template <int FP>
__device__ __forceinline__ void lds_rank_frag(const float* Z, int row, int t, bool ok,
                                              float (&frag)[FP / 4]) {
  const float* p = Z + row * FP;
  if constexpr (FP >= 16) {
#pragma unroll
    for (int q = 0; q < FP / 16; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + 16 * q + 4 * t);
      frag[4 * q + 0] = ok ? v[0] : 0.f; frag[4 * q + 1] = ok ? v[1] : 0.f;
      frag[4 * q + 2] = ok ? v[2] : 0.f; frag[4 * q + 3] = ok ? v[3] : 0.f;
    }
  } else {
#pragma unroll
    for (int s = 0; s < FP / 4; ++s) frag[s] = ok ? p[4 * s + t] : 0.f;
  }
}

// Keeps a loaded value opaque to the optimiser. `cond ? loaded : 0` is otherwise turned back
// into a branch around the load (one exec-masked load + vmcnt(0) per value: the kernel
// prologues were chains of 17 serialized L2 round trips, ~4 us per launch).
__device__ __forceinline__ float keep(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ f32x4 keep(f32x4 v) { asm volatile("" : "+v"(v)); return v; }

// Stage Wt(k, c) (k < d, c < r) into MFMA fragment order [NQ][NCB][64] float4: element
// (k = 16 q + 4 kq + t, c = 16 cb + n) sits at float ((q * NCB + cb) * 64 + n + 16 kq) * 4 + t,
// so the 4 consecutive k of one (c, k / 4) group are one 16-byte LDS word. Groups are handled
// kStageBatch at a time per thread, all requests of a batch in flight before the first write.
//   DOWN: Wt(k, c) = W_down[c][k]  (a group is one aligned float4 of a W_down row)
//   else: Wt(k, c) = W_up[k][c]    (a group is 4 scalars, stride r)
constexpr int kStageBatch = 4;
template <int NCB, int NQ, bool DOWN>
__device__ __forceinline__ void stage_weights(f32x4* wl, const float* __restrict__ W, int d, int r) {
  constexpr int kgroups = NQ * 4, cpad = NCB * 16, total = kgroups * cpad;
  const int stride = blockDim.x;
  for (int base = threadIdx.x; base < total; base += kStageBatch * stride) {
    f32x4 v[kStageBatch];
#pragma unroll
    for (int u = 0; u < kStageBatch; ++u) {
      const int g = min(base + u * stride, total - 1);
      const int c = g / kgroups, k4 = g % kgroups;
      const int cc = min(c, r - 1), kk = min(4 * k4, d - 4);
      if (DOWN) {
        v[u] = ldg4(W + static_cast<size_t>(cc) * d + kk);
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[u][t] = W[static_cast<size_t>(kk + t) * r + cc];
      }
    }
#pragma unroll
    for (int u = 0; u < kStageBatch; ++u) {
      const int g = base + u * stride;
      const int c = g / kgroups, k4 = g % kgroups;
      const bool ok = c < r && 4 * k4 < d;
      const f32x4 val = keep(v[u]);
      const int q = k4 >> 2, kq = k4 & 3, cb = c >> 4, ln = (c & 15) + 16 * kq;
      if (g < total) wl[(q * NCB + cb) * 64 + ln] = ok ? val : zero4f();
    }
  }
}

// Offset that lies outside every buffer the tile kernels address through a descriptor (their
// byte counts stay below it: tile_path_ok); what is added to it (a k-step inside a row) cannot
// overflow 32 bits.
constexpr int kRowOob = 0x7FFF0000;

// One tile as the kernels see it (wave-uniform) + its halo slot. The kernels read the first 16
// bytes of a gconv_tile_t only ({first, count, n1, n0}: a record prefetched for the next pass
// sits in VGPRs for a whole tile — the forward kernel has none to spare) and pick up slot / e0 with
// a second, short-lived request at the top of the pass; the edge counts come from the row
// offsets the CSR staging requests anyway.
struct TileView {
  int ts, nt, n1, n0, e0;
  const int32_t* hb;   // the tile's halo slot (tiles.h)
};
__device__ __forceinline__ i32x4 tile_rec_load(const gconv_tile_t* tiles, int idx) {
  return *reinterpret_cast<const i32x4*>(tiles + idx);
}
__device__ __forceinline__ TileView tile_view(const gconv_tile_t* tiles, int idx, const i32x4& r0,
                                              const int32_t* halo) {
  TileView v;
  const int2 r1 = *reinterpret_cast<const int2*>(&tiles[idx].slot);   // {slot, e0}
  v.ts = uniform(r0[0]); v.nt = uniform(r0[1]); v.n1 = uniform(r0[2]); v.n0 = uniform(r0[3]);
  v.e0 = uniform(r1.y);
  v.hb = halo + static_cast<size_t>(uniform(r1.x)) * kHaloSlot;
  return v;
}

// Entry of the size-sorted tile list that workgroup w = blockIdx.x runs in pass `pass` (may lie past the end of the
// list: the ragged pass has no tile for some workgroups). Boustrophedon: even passes deal forwards, odd passes
// backwards. `rev` (wave-uniform 0 / 1): this workgroup takes ITS passes in the opposite order, smallest tile first.
// HALF the workgroups do (tile_walk_reversed): the chip's workgroups are then not all in the same phase of equally
// large tiles (round 5, in the step: -3.4 ... -3.8 %; rotated starts in 2 or 4 groups, or other choices of the half,
// did no better: profiles/r05_lab_tile_step_policies.txt). Which half alternates between consecutive adapters of a
// model (gconv_adapter_params_t::walk), so that a workgroup starts a kernel with the tile it ended the previous one with.
// (pure arithmetic on purpose: with ?: hipcc emits scalar BRANCHES here, basic-block boundaries in the kernels'
//  prologues, and splits the kernel-argument loads into three dependent batches in front of the first request)
__device__ __forceinline__ int tile_order_index(int pass, int npass, int rev) {
  const int G = gridDim.x, w = blockIdx.x;
  const int p = pass + rev * (npass - 1 - 2 * pass);          // rev ? npass - 1 - pass : pass
  return p * G + w + (p & 1) * (G - 1 - 2 * w);                // odd pass: (p + 1) G - 1 - w, even pass: p G + w
}
// a.stagger: 0 = nobody, 1 = the odd workgroups, 2 = the even ones
__device__ __forceinline__ int tile_walk_reversed(int stagger) {
  return static_cast<int>(stagger > 0) & (((static_cast<int>(blockIdx.x) ^ stagger) & 1) ^ 1);
}

// graph node of the tile's local row lr (< n0): own rows are contiguous, halo rows come from the slot
__device__ __forceinline__ int tile_node(const TileView& tv, int lr) {
  const int hid = tv.hb[min(max(lr - tv.nt, 0), kHaloNodes - 1)];   // (requested for every lr: no branch)
  return lr < tv.nt ? tv.ts + lr : hid;
}

// Phase 1 of both directions: dst[local row][c] = row_scale * sval * (M Wt) for the tile's n0
// rows (own rows, then its halo nodes). Returns this lane's partial of sum(acc * Pmat) over the
// OWN rows (backward scalar gradient).
// `rm` is a descriptor over ALL rows of M (n * d floats): a lane whose row lies past the tile's
// n0 rows, and a k-step past the row end, is pushed out of its range and reads zeros.
//
// A wave owns the 16-row blocks wave, wave + nwaves, ...; the k-steps of a block sit in a
// register ring: all of them are requested before the first MFMA, and while block b is
// multiplied each freed ring slot is refilled with block b + nwaves (the refill variant of
// the block body is a separate straight-line copy, selected by a wave-uniform branch, so the
// compiler can count vmcnt exactly and never waits for more than it needs). `mid` runs once,
// right after the first block's requests are in flight: the caller's small dependent loads
// (CSR slice, dis) ride under the X / G tile latency instead of in front of it.
template <int FP, int NQ, bool HAS_DIS, bool WANT_P, class Mid>
__device__ __forceinline__ float tile_rowsA(rsrc_t rm, const float* __restrict__ anyrow,
                                            const f32x4* wl, float* dst,
                                            const float* __restrict__ dis,
                                            const float* __restrict__ Pmat, float sval,
                                            const TileView& tv, int d, int rowb, int nwaves, int wave,
                                            int lane, Mid&& mid) {
  constexpr int NCB = FP >= 16 ? FP / 16 : 1;
  const int node = lane & 15, kq = lane >> 4;
  const int ts = tv.ts, nt = tv.nt, n0 = tv.n0;
  const int nb = ceil_div(n0, 16);
  const float* hdis = reinterpret_cast<const float*>(tv.hb + kHaloDisOff);
  float ds_acc = 0.f;
  int b = wave;
  if (b >= nb) {  // wave-uniform: no block for this wave in this tile
    mid();
    return 0.f;
  }
  // ring of R k-steps (4 VGPRs each): the whole row up to d = 304 (d = 192 for rank 17..32,
  // whose accumulators are twice as large), half a row beyond. Slot q % R serves k-steps
  // q and q + R of a block, then k-step q % R of the wave's next block.
  constexpr int R = (NQ > 19 || (FP >= 32 && NQ > 12)) ? (NQ + 1) / 2 : NQ;
  f32x4 x[R];
  // The per-k-step offsets are loop invariant; left alone the compiler hoists all NQ of them
  // into registers for the whole kernel (+40 VGPRs at d = 300). Laundering kq makes it
  // recompute them: two VALU ops per 16-byte load.
  auto launder = [](int v) { asm volatile("" : "+v"(v)); return v; };
  auto koff = [&](int rowoff, int q, int kqv) {  // a k-step past the row end reads zeros
    return 16 * q + 4 * kqv < d ? rowoff + 64 * q + 16 * kqv : kRowOob;
  };
  // byte offset of this lane's row of node block `blk` in M: own rows are contiguous from ts,
  // halo rows come from the slot's node list (only a block that reaches past the own rows of a
  // tile WITH a halo pays that look-up; wave-uniform branch)
  auto rowoff = [&](int blk) {
    const int lr = 16 * blk + node;
    int g = ts + lr;
    if (n0 > nt && 16 * blk + 16 > nt) {
      const int hid = tv.hb[min(max(lr - nt, 0), kHaloNodes - 1)];
      g = lr < nt ? g : hid;
    }
    return lr < n0 ? g * rowb : kRowOob;
  };
  {
    const int ro = rowoff(b), kqv = launder(kq);
#pragma unroll
    for (int q = 0; q < R; ++q) x[q] = buf_ld4(rm, koff(ro, q, kqv));
  }
  mid();
  const float* pmat = WANT_P ? (Pmat ? Pmat : anyrow) : nullptr;  // never read out of range
  auto block = [&](auto refill_tag, int blk) {
    constexpr bool REFILL = decltype(refill_tag)::value;
    // epilogue operands first (vmcnt retires in order: they sit between this block's tiles,
    // requested one block ago, and the refills requested below)
    float dv[4], pv[NCB][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int lrc = min(16 * blk + 4 * kq + m, n0 - 1);
      // own rows: dis of the graph; halo rows: the slot's copy (ONE request, from a selected address)
      dv[m] = HAS_DIS ? *(lrc < nt ? dis + ts + lrc : hdis + (lrc - nt)) : 1.f;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
        pv[cb][m] = WANT_P ? pmat[static_cast<size_t>(ts + min(lrc, nt - 1)) * FP + min(16 * cb + node, FP - 1)] : 0.f;
    }
    const int rc = rowoff(blk);
    int rn = 0;
    if constexpr (REFILL) rn = rowoff(blk + nwaves);
    const int kqv = launder(kq);
    f32x4 acc[NCB][2];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb][0] = acc[cb][1] = zero4f();
    // W fragments: LDS reads one k-step ahead of their MFMAs; the scheduling fences keep the
    // compiler from hoisting all NQ reads (4 VGPRs each) to the top of the block
    f32x4 w[NCB], wn[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) w[cb] = wn[cb] = wl[cb * 64 + lane];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      // operand copy behind an opaque asm: without it the compiler gives the refill below a
      // second set of ring registers (and copies the whole ring at the loop head)
      f32x4 xv = x[q % R];
      asm volatile("" : "+v"(xv));
      if (q + 1 < NQ) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) wn[cb] = wl[((q + 1) * NCB + cb) * 64 + lane];
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        acc[cb][0] = mfma16(xv[0], w[cb][0], acc[cb][0]);
        acc[cb][1] = mfma16(xv[1], w[cb][1], acc[cb][1]);
        acc[cb][0] = mfma16(xv[2], w[cb][2], acc[cb][0]);
        acc[cb][1] = mfma16(xv[3], w[cb][3], acc[cb][1]);
      }
      if (q + R < NQ) x[q % R] = buf_ld4(rm, koff(rc, q + R, kqv));
      else if constexpr (REFILL) x[q % R] = buf_ld4(rm, koff(rn, q % R, kqv));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) w[cb] = wn[cb];
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const f32x4 res = acc[cb][0] + acc[cb][1];
      const int c = 16 * cb + node;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int lr = 16 * blk + 4 * kq + m;
        if (lr < n0 && c < FP) {
          float v = res[m];
          if (WANT_P && Pmat && lr < nt) ds_acc += v * pv[cb][m];
          v *= sval * dv[m];
          dst[lr * FP + c] = v;
        }
      }
    }
  };
  while (b + nwaves < nb) {
    block(std::true_type{}, b);
    b += nwaves;
  }
  block(std::false_type{}, b);
  return ds_acc;
}

// Phases 2 / 3: one lane group (FP/4 lanes) per tile row, neighbour rows gathered from LDS.
//   sum = sum over stored edges of src[col - ts]; v = NORM ? dis * (sum + src[row]) : sum
// `rp` / `cl` / `dl`: the tile's CSR slice (local row offsets, local column ids) and its
// dis values, staged in LDS during phase 1
template <int FP, bool NORM>
__device__ __forceinline__ f32x4 tile_row_sum(const float* src, const int32_t* rp,
                                              const int32_t* cl, const float* dl, int lrow,
                                              int sub) {
  const int e0 = rp[lrow], e1 = rp[lrow + 1];
  f32x4 acc = zero4f();
  for (int e = e0; e < e1; ++e)
    acc += *reinterpret_cast<const f32x4*>(src + cl[e] * FP + 4 * sub);
  if (NORM) {
    acc += *reinterpret_cast<const f32x4*>(src + lrow * FP + 4 * sub);
    acc *= dl[lrow];
  }
  return acc;
}

// Split-phase staging of a tile's local CSR and dis values into LDS: `request` (before the
// phase-1 tile loads) asks for the own rows' offsets, dis and the halo slot's small arrays,
// `advance` (the phase-1 `mid` hook: the big tile loads are in flight by then) asks for the column
// ids and parks what has arrived; `finish` parks the column ids. The barrier that ends phase 1
// publishes everything. Local rows: own rows 0 .. nt - 1 (columns already tile-local in `lcol`),
// then the halo1 rows nt .. n1 - 1 (rowptr and columns from the slot). First blockDim elements in
// registers, the rest (own parts of tiles with more rows / edges than threads) by plain loops in
// `finish`; the halo parts never exceed a workgroup (64 rows, 256 edges; blockDim >= 256).
template <bool NORM>
struct TileCsrStage {
  const int32_t* __restrict__ rowptr;
  const int32_t* __restrict__ lcol;
  const float* __restrict__ dis;
  int32_t *rp, *cl;
  float* dl;
  TileView tv;
  int rpv, hrpv, clv, e1v, hev, ne, nhe;
  float dv;
  __device__ __forceinline__ void request() {
    const int tid = threadIdx.x, nt = tv.nt;
    rpv = rowptr[tv.ts + min(tid, nt)];
    e1v = rowptr[tv.ts + nt];                                   // (same word in every lane)
    hrpv = tv.hb[kHaloRpOff + min(tid, kHaloNodes)];
    hev = tv.hb[kHaloRpOff + tv.n1 - nt];                       // stored edges of the halo1 rows
    const float* hdis = reinterpret_cast<const float*>(tv.hb + kHaloDisOff);
    dv = NORM ? *(tid < nt ? dis + tv.ts + tid : hdis + min(tid - nt, kHaloNodes - 1)) : 1.f;
  }
  // column id e of the tile's local edge list: own rows' edges first, then the halo1 rows'
  __device__ __forceinline__ int col_at(int e) const {
    const int ec = min(e, max(ne + nhe - 1, 0));
    return *(ec < ne ? lcol + tv.e0 + ec : tv.hb + kHaloColOff + (ec - ne));
  }
  __device__ __forceinline__ void advance() {
    const int tid = threadIdx.x, nt = tv.nt;
    ne = uniform(e1v) - tv.e0;
    nhe = min(uniform(hev), kHaloEdges);
    clv = col_at(tid);
    if (tid <= nt) rp[tid] = rpv - tv.e0;
    if (tid >= 1 && tid <= tv.n1 - nt) rp[nt + tid] = ne + hrpv;   // (rp[nt] = ne from the own side)
    if (NORM && tid < tv.n0) dl[tid] = dv;
  }
  __device__ __forceinline__ void finish() {
    const int tid = threadIdx.x, nt = tv.nt;
    if (tid < ne + nhe) cl[tid] = clv;
    for (int e = tid + blockDim.x; e < ne + nhe; e += blockDim.x) cl[e] = col_at(e);
    for (int i = tid + blockDim.x; i <= nt; i += blockDim.x) rp[i] = rowptr[tv.ts + i] - tv.e0;
    if (NORM) {   // rows of a tile with more rows than threads
      const float* hdis = reinterpret_cast<const float*>(tv.hb + kHaloDisOff);
      for (int i = tid + blockDim.x; i < tv.n0; i += blockDim.x) dl[i] = i < nt ? dis[tv.ts + i] : hdis[i - nt];
    }
  }
};

}  // namespace

// ===========================================================================
// forward — persistent workgroups walk the tiles (weights staged / loaded once)
// ===========================================================================
template <int FP, int NQ, bool NORM, bool NT>
#if GCONV_TILE_WGPC >= 2
__global__ __launch_bounds__(320, 3) void tile_fwd_kernel(TileFwdArgs a) {
#else
__global__ __launch_bounds__((FP >= 32 || NQ > 19) ? 512 : 640) void tile_fwd_kernel(TileFwdArgs a) {
#endif
  extern __shared__ f32x4 lds4[];
  constexpr int NCB = FP >= 16 ? FP / 16 : 1;
  constexpr int KS = FP / 4;
  constexpr int LPR = FP / 4;
  if (GCONV_TILE_PRIO && a.npass > GCONV_TILE_PRIO_MIN_PASSES) __builtin_amdgcn_s_setprio(GCONV_TILE_PRIO);
  TILE_STAMP(8);
  f32x4* wl = lds4;                                              // [NQ][NCB][64]
  float* bufA = reinterpret_cast<float*>(lds4 + NQ * NCB * 64);  // [cap][FP]  H0', later P
  float* bufB = bufA + a.cap * FP;                               // [cap][FP]  H2'
  float* bufC = bufB + a.cap * FP;                               // (backward only: P tile)
  int32_t* rp = reinterpret_cast<int32_t*>(bufC + a.cap * FP + 8 * (FP + 2));  // [cap + 1]
  float* dl = reinterpret_cast<float*>(rp + a.cap + 16);         // [cap]  dis of the tile
  int32_t* cl = reinterpret_cast<int32_t*>(dl + a.cap);          // [ecap]

  const int lane = threadIdx.x & 63, wave = wave_id();
  const int nwaves = blockDim.x >> 6;
  const int jj = lane & 15, t = lane >> 4;
  const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR, ngrp = blockDim.x / LPR;

  // Every first-touch load of a launch misses L2 (the previous kernel streamed 30+ MB through
  // it): the prologue requests ALL of them back to back — register fragments, scalar, the first
  // tile's record — and only then stages the LDS weights, so it pays one cold round trip, not four.
  TILE_STAMP(9);
  // (a BUFFER load, counted in vmcnt like the rest of the prologue: as a scalar load this value's cold round trip was
  //  waited for with lgkmcnt(0) together with the kernel arguments, IN FRONT of every other request — round 5: the
  //  walk-order arithmetic below made hipcc place it there, +0.4 us per kernel on one-tile-per-workgroup batches;
  //  through a VGPR address it became a flat load, which drains vmcnt AND lgkmcnt wherever it is waited for)
  const float sraw = buf_ld1(make_rsrc(a.scalar ? a.scalar : a.w_down, 4u), 0);
  // this workgroup's walk: passes [pass0, pend). Only the ragged pass — the LAST one of the common order, hence the
  // FIRST one of a reversed walk — may have no tile for it: it is cut off here, so that the loop below has no
  // "skip" path (a request on such a path gave hipcc's vmcnt bookkeeping a pessimistic join: vmcnt(0) at the top
  // of every tile, +0.3 us per kernel, round 5)
  const int rev = tile_walk_reversed(a.stagger);
  const int ragged = static_cast<int>(tile_order_index(rev ? 0 : a.npass - 1, a.npass, rev) >= a.n_tiles);
  const int pass0 = rev & ragged, pend = a.npass - ((rev ^ 1) & ragged);
  i32x4 rec = tile_rec_load(a.tiles, min(tile_order_index(pass0, a.npass, rev), a.n_tiles - 1));   // (grid <= n_tiles)
#ifdef GCONV_TILE_STAMPS
  if (threadIdx.x == 0) {   // where this workgroup runs: HW_ID (wave / simd / cu / sh / se) and XCC_ID
    a.stamps[blockIdx.x * 16 + 12] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    a.stamps[blockIdx.x * 16 + 13] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  }
#endif

  // phase-4 ownership: a wave owns ONE 64-column chunk (W_up fragments in registers) and
  // every nslots-th node block of a tile (nchunks <= nwaves is guaranteed by the launcher)
  const int nchunks = ceil_div(a.d, 64);
  const int nslots = nwaves / nchunks;
  const int J = wave % nchunks, slot = wave / nchunks;
  const bool active = slot < nslots;
  const int colbase = 64 * J + 4 * jj;
  const bool colvalid = colbase < a.d;
  const int colc = min(colbase, a.d - 4);
  // all requests first, the selects (behind keep()) afterwards: one round trip, not one each
  float wb[KS][4];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = rank_k_<FP>(s, t);
#pragma unroll
    for (int c = 0; c < 4; ++c) wb[s][c] = a.w_up[static_cast<size_t>(colc + c) * a.r + min(k, a.r - 1)];
  }
  f32x4 bu = ldg4((a.b_up ? a.b_up : a.w_down) + colc);          // (any readable address if absent)
  f32x4 bdv;                                                      // b_down entries of this lane group
#pragma unroll
  for (int k = 0; k < 4; ++k) bdv[k] = (a.b_down ? a.b_down : a.w_down)[min(4 * sub + k, a.r - 1)];
  TILE_STAMP(10);
  stage_weights<NCB, NQ, true>(wl, a.w_down, a.d, a.r);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = rank_k_<FP>(s, t);
#pragma unroll
    for (int c = 0; c < 4; ++c) wb[s][c] = (colvalid && k < a.r) ? keep(wb[s][c]) : 0.f;
  }
  bu = (a.b_up && colvalid) ? keep(bu) : zero4f();
  bdv = keep(bdv);
  const float sval = a.scalar ? keep(sraw) : 1.f;
  TILE_STAMP(11);
#pragma unroll
  for (int k = 0; k < 4; ++k) bdv[k] = (a.b_down && 4 * sub + k < a.r) ? bdv[k] : 0.f;
  lds_barrier();

  TILE_STAMP(0);
  // Tiles are dealt from the size-sorted list boustrophedon: pass 0 gives workgroup w the w-th largest tile,
  // pass 1 the (2 G - 1 - w)-th, ... — the big tiles end up alone. a.stagger: workgroups with an odd index walk
  // THEIR tiles in the opposite order (smallest first), so that the chip's workgroups are not all in the same
  // phase of a tile at the same time (tile_order_index below).
  for (int pass = pass0; pass < pend; ++pass) {   // (every pass in [pass0, pend) has a tile for this workgroup)
    const int idx = tile_order_index(pass, a.npass, rev);
    const int nidx = min(max(tile_order_index(pass + 1, a.npass, rev), 0), a.n_tiles - 1);
    // ORDER MATTERS (vmcnt retires in order): tile_view's short request for {slot, e0} — a line the record's own
    // request has already pulled in — goes out BEFORE the next pass's record, a cold miss; issued behind it, every
    // tile waited for that miss at its top (+0.6 us per kernel on one-tile-per-workgroup batches, round 5)
    const TileView tv = tile_view(a.tiles, idx, rec, a.halo);      // the tile of this pass
    const int ts = tv.ts, nt = tv.nt;
    rec = tile_rec_load(a.tiles, nidx);   // the next pass's record is requested now (clamped: always a valid entry)
    if (nt <= 0) continue;  // uniform
    TILE_STAMP_NT(nt);
    TILE_STAMP(1);
    // Row pitch in bytes, re-derived per tile behind an opaque asm: the byte offsets built on
    // it are tile invariant, and hoisted out of this loop they would be kept (and spilled)
    // for the whole kernel instead of being recomputed with one or two VALU ops.
    int rowb = 4 * a.d;
    asm volatile("" : "+s"(rowb));
    // phase-4 byte offset of this lane inside a tile: row 4t of a node block, its 4 columns;
    // a lane past the last column is pushed out of the tile's range (loads 0, stores nothing)
    const int voff4 = 4 * t * rowb + (colvalid ? 4 * colbase : kBufOob);
    // the tile's own rows of X and of out as range-checked buffers (phase 4); all of X (phase 1
    // also reads the halo nodes' rows)
    const rsrc_t rx = make_rsrc(a.x + static_cast<size_t>(ts) * a.d, static_cast<uint32_t>(nt) * rowb);
    const rsrc_t ro = make_rsrc(a.out + static_cast<size_t>(ts) * a.d, static_cast<uint32_t>(nt) * rowb);
    const rsrc_t rxa = make_rsrc(a.x, static_cast<uint32_t>(a.n) * rowb);

    // ---- phase 1: H0' -> bufA, n0 rows (the tile's local CSR and dis land in LDS meanwhile)
    TileCsrStage<NORM> csr{a.rowptr, a.lcol, a.dis, rp, cl, dl, tv};
    csr.request();
    tile_rowsA<FP, NQ, NORM, false>(rxa, a.x, wl, bufA, a.dis, nullptr, 1.f, tv, a.d, rowb, nwaves, wave,
                                    lane, [&]() { csr.advance(); });
    csr.finish();
    lds_barrier();
    TILE_STAMP(2);

    // phase 4's first two X tiles are requested now: their latency passes under phases 2 / 3.
    // Unconditional requests (a conditional prefetch makes the compiler drain vmcnt): a node
    // block past the tile lies outside the buffer and costs no traffic.
    const int nb4 = ceil_div(nt, 16);
    auto load_x = [&](int blk, f32x4 (&xq)[4]) {
#ifdef GCONV_LAB_FWD_NO_XREAD   // lab build (wrong results): what phase 4 costs without its re-read of X
#pragma unroll
      for (int m = 0; m < 4; ++m) xq[m] = zero4f();
      (void)blk;
#else
#pragma unroll
      for (int m = 0; m < 4; ++m) xq[m] = buf_ld4_last(rx, voff4 + (16 * blk + m) * rowb);
#endif
    };
    f32x4 xa[4], xb[4];
    load_x(slot, xa);
    load_x(slot + nslots, xb);

    // ---- phase 2: H2' -> bufB, n1 rows (own rows: + saved_h, saved_h1)
    for (int lr = grp; lr < tv.n1; lr += ngrp) {
      f32x4 v = tile_row_sum<FP, NORM>(bufA, rp, cl, dl, lr, sub) + bdv;
      const size_t off = static_cast<size_t>(ts + lr) * FP + 4 * sub;
      // a halo row's value is stored by the tile that owns it; a forward-only call (saved_h == NULL:
      // gconv_adapter_infer_f32, evaluation under no_grad) keeps nothing
      const bool own = lr < nt && a.saved_h != nullptr;
      if (a.act == GCONV_ACT_SILU && own) stg4(a.saved_h1 + off, v);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = act_fwd(a.act, v[k]);
      if (NORM) v *= dl[lr];
      *reinterpret_cast<f32x4*>(bufB + lr * FP + 4 * sub) = v;
      if (own) stg4(a.saved_h + off, v);
    }
    lds_barrier();

    TILE_STAMP(3);
    // ---- phase 3: P -> bufA (+ saved_p)
    for (int lr = grp; lr < nt; lr += ngrp) {
      const f32x4 v = tile_row_sum<FP, NORM>(bufB, rp, cl, dl, lr, sub);
      *reinterpret_cast<f32x4*>(bufA + lr * FP + 4 * sub) = v;
      if (a.saved_p) stg4(a.saved_p + static_cast<size_t>(ts + lr) * FP + 4 * sub, v);
    }
    lds_barrier();

    TILE_STAMP(4);
    // ---- phase 4: out = s (P Wu^T + bu + X); two X tiles per wave in flight. Everything
    // requested so far has landed (the phases in between waited on younger loads): saying so
    // gives the loop below one known state on entry and around its back edge.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    // NT (template argument, chosen per launch: tile_store_nt): `out` with non-temporal or with default-policy
    // stores. `out` is the NEXT kernel's X: on the kernel alone nt wins (round 1: nothing re-reads the output
    // there); in the step, with several tiles per workgroup, the default policy does (round 5: whole batch -4 %).
    if (active && slot < nb4) {
      auto step = [&](int b, f32x4 (&xv)[4], int refill) {
        float pa[KS];
        lds_rank_frag<FP>(bufA, min(16 * b + jj, nt - 1), t, 16 * b + jj < nt, pa);
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = zero4f();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = mfma16(pa[s], wb[s][c], acc[c]);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = sval * (acc[c][m] + bu[c] + (a.skip ? xv[m][c] : 0.f));
          buf_st4_p<NT>(ro, voff4 + (16 * b + m) * rowb, o);  // rows / columns past the tile: dropped
        }
        load_x(refill, xv);
      };
      for (int b = slot;;) {
        step(b, xa, b + 2 * nslots);
        b += nslots;
        if (b >= nb4) break;
        step(b, xb, b + 2 * nslots);
        b += nslots;
        if (b >= nb4) break;
      }
    }
    lds_barrier();  // bufA / bufB are reused by the next tile
    TILE_STAMP(5);
  }
}

// ===========================================================================
// backward — persistent workgroups; the dW^T tiles, colsum(G), bias gradient and the
// scalar-gradient pieces stay in registers across all tiles of a workgroup, so ONE
// partial row per workgroup reaches HBM and B5 is a single launch.
// ===========================================================================
template <int FP, int NQ, bool NORM, bool NEED_DX, bool NT>
#if GCONV_TILE_WGPC >= 2
__global__ __launch_bounds__(320, 3) void tile_bwd_kernel(TileBwdArgs a) {
#else
__global__ __launch_bounds__(512, GCONV_BWD_WAVES) void tile_bwd_kernel(TileBwdArgs a) {
#endif
  extern __shared__ f32x4 lds4[];
  constexpr int NCB = FP >= 16 ? FP / 16 : 1;
  constexpr int KS = FP / 4;
  constexpr int LPR = FP / 4;
  if (GCONV_TILE_PRIO && a.npass > GCONV_TILE_PRIO_MIN_PASSES) __builtin_amdgcn_s_setprio(GCONV_TILE_PRIO);
  TILE_STAMP(8);
  f32x4* wl = lds4;                                              // [NQ][NCB][64]  W_up^T frags
  float* bufA = reinterpret_cast<float*>(lds4 + NQ * NCB * 64);  // [cap][FP]  dP', later T
  float* bufB = bufA + a.cap * FP;                               // [cap][FP]  dH1'
  float* bufC = bufB + a.cap * FP;                               // [FP][cap]  P tile, transposed
  float* bufD = bufC + a.cap * FP;                               // [FP][cap]  T, transposed
  float* red = bufD + a.cap * FP;                                // [nwaves][FP + 2] scratch
  int32_t* rp = reinterpret_cast<int32_t*>(red + 8 * (FP + 2));  // [cap + 1]
  float* dl = reinterpret_cast<float*>(rp + a.cap + 16);         // [cap]  dis of the tile
  int32_t* cl = reinterpret_cast<int32_t*>(dl + a.cap);          // [ecap]

  const int lane = threadIdx.x & 63, wave = wave_id();
  const int nwaves = blockDim.x >> 6;
  const int pid = blockIdx.x;
  // (prologue: all first-touch requests back to back, LDS staging after them — see forward)
  // (a BUFFER load, counted in vmcnt like the rest of the prologue: as a scalar load this value's cold round trip was
  //  waited for with lgkmcnt(0) together with the kernel arguments, IN FRONT of every other request — round 5: the
  //  walk-order arithmetic below made hipcc place it there, +0.4 us per kernel on one-tile-per-workgroup batches;
  //  through a VGPR address it became a flat load, which drains vmcnt AND lgkmcnt wherever it is waited for)
  const float sraw = buf_ld1(make_rsrc(a.scalar ? a.scalar : a.w_down, 4u), 0);
  // this workgroup's walk: passes [pass0, pend). Only the ragged pass — the LAST one of the common order, hence the
  // FIRST one of a reversed walk — may have no tile for it: it is cut off here, so that the loop below has no
  // "skip" path (a request on such a path gave hipcc's vmcnt bookkeeping a pessimistic join: vmcnt(0) at the top
  // of every tile, +0.3 us per kernel, round 5)
  const int rev = tile_walk_reversed(a.stagger);
  const int ragged = static_cast<int>(tile_order_index(rev ? 0 : a.npass - 1, a.npass, rev) >= a.n_tiles);
  const int pass0 = rev & ragged, pend = a.npass - ((rev ^ 1) & ragged);
  i32x4 rec = tile_rec_load(a.tiles, min(tile_order_index(pass0, a.npass, rev), a.n_tiles - 1));   // (grid <= n_tiles)
#ifdef GCONV_TILE_STAMPS
  if (threadIdx.x == 0) {
    a.stamps[blockIdx.x * 16 + 12] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    a.stamps[blockIdx.x * 16 + 13] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  }
#endif
  const int nchunks = ceil_div(a.d, 64);
  const int jj = lane & 15, t = lane >> 4;
  const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR, ngrp = blockDim.x / LPR;

  // phase-4 ownership: wave J < nchunks is the primary owner of 64-column chunk J (its dW
  // accumulators live in that wave's registers for the whole kernel). Up to kExtra of the
  // remaining waves co-own chunks, last chunk first, each with accumulators of its own: the
  // co-owners of a chunk split its node blocks (slot, nslots) and are summed into the primary
  // owner once, at the end. At d = 300 that is waves 0..3 -> chunks 0..3 and waves 4..7 ->
  // chunk 4: 1.25 chunks of MFMA work on each of the CU's four SIMDs instead of 2 on one.
  constexpr int kExtra = FP >= 32 ? 1 : 3;
  const int extras = min(kExtra, nwaves - nchunks);
  const bool primary = wave < nchunks;
  const bool owner = wave < nchunks + extras;
  const int ex = wave - nchunks;                                    // co-owner index
  const int J = primary ? wave : (owner ? nchunks - 1 - ex % nchunks : 0);
  const int jr = nchunks - 1 - J;
  const int slot = primary ? 0 : 1 + ex / nchunks;
  const int nslots = 1 + (extras > jr ? (extras - 1 - jr) / nchunks + 1 : 0);
  // Phase 4 works in the "column per lane" layout of the 4x4x1 MFMA: lane l of the wave that
  // owns chunk J holds column 64 J + l of a row (256 contiguous bytes per row and instruction),
  // the rank-width operand (4 of the FP values of a node) is the same in all 16 blocks.
  constexpr int MT = FP / 4;            // 4-row tiles of the rank dimension
  const int l4 = lane & 3;
  const int col1 = 64 * J + lane;
  const bool colvalid = owner && col1 < a.d;
  const int colc = min(col1, a.d - 1);
  float wd[NEED_DX ? FP : 1];          // W_down[k][col] of this lane's column
  if constexpr (NEED_DX) {
#pragma unroll
    for (int k = 0; k < FP; ++k) wd[k] = a.w_down[static_cast<size_t>(min(k, a.r - 1)) * a.d + colc];
  }
  stage_weights<NCB, NQ, false>(wl, a.w_up, a.d, a.r);
  const float sval = a.scalar ? keep(sraw) : 1.f;
  if constexpr (NEED_DX) {
#pragma unroll
    for (int k = 0; k < FP; ++k) wd[k] = (colvalid && k < a.r) ? keep(wd[k]) : 0.f;
  }
  f32x4 awd[MT], awu[MT];              // dWd^T[4 mt + v][col], dWu^T[4 mt + v][col] in reg v
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) awd[mt] = awu[mt] = zero4f();
  float cs = 0.f;                      // colsum(G)[col]
  f32x4 bgrad = zero4f();
  float ds1 = 0.f, dsx = 0.f;
  lds_barrier();

  TILE_STAMP(0);
  // tiles are dealt from the size-sorted list boustrophedon (see the forward kernel; a.stagger: odd workgroups
  // walk their tiles smallest first — their partial sums then add the same tiles in the opposite order: fixed per
  // launch geometry, so still the same bits run to run)
  for (int pass = pass0; pass < pend; ++pass) {   // (every pass in [pass0, pend) has a tile for this workgroup)
    const int idx = tile_order_index(pass, a.npass, rev);
    const int nidx = min(max(tile_order_index(pass + 1, a.npass, rev), 0), a.n_tiles - 1);
    // ORDER MATTERS (vmcnt retires in order): tile_view's short request for {slot, e0} — a line the record's own
    // request has already pulled in — goes out BEFORE the next pass's record, a cold miss; issued behind it, every
    // tile waited for that miss at its top (+0.6 us per kernel on one-tile-per-workgroup batches, round 5)
    const TileView tv = tile_view(a.tiles, idx, rec, a.halo);      // the tile of this pass
    const int ts = tv.ts, nt = tv.nt;
    rec = tile_rec_load(a.tiles, nidx);   // the next pass's record is requested now (clamped: always a valid entry)
    if (nt <= 0) continue;  // uniform
    TILE_STAMP_NT(nt);
    TILE_STAMP(1);
    // Row pitch in bytes, re-derived per tile behind an opaque asm: the byte offsets built on
    // it are tile invariant, and hoisted out of this loop they would be kept (and spilled)
    // for the whole kernel instead of being recomputed with one or two VALU ops.
    int rowb = 4 * a.d;
    asm volatile("" : "+s"(rowb));
    // phase-4 byte offset of this lane's column inside a row; a lane past the last column, and
    // every lane of a wave that owns no chunk, is pushed out of the tile's range (loads 0
    // without traffic, stores nothing)
    const int voff1 = colvalid ? 4 * col1 : kBufOob;
    // the tile's own rows of G, X and dX as range-checked buffers (phase 4); all of G (phase 1
    // also reads the halo nodes' rows)
    const uint32_t tbytes = static_cast<uint32_t>(nt) * rowb;
    const rsrc_t rg = make_rsrc(a.g + static_cast<size_t>(ts) * a.d, tbytes);
    const rsrc_t rx = make_rsrc(a.x + static_cast<size_t>(ts) * a.d, tbytes);
    const rsrc_t rdx = make_rsrc(NEED_DX ? a.dx + static_cast<size_t>(ts) * a.d : nullptr, NEED_DX ? tbytes : 0);
    const rsrc_t rga = make_rsrc(a.g, static_cast<uint32_t>(a.n) * rowb);

    // ---- phase 1: dP' = dis . s (G Wu) -> bufA, n0 rows ; ds1 += sum over the own rows (G Wu) . P
    // (the tile's local CSR and its P rows land in LDS meanwhile)
    TileCsrStage<NORM> csr{a.rowptr, a.lcol, a.dis, rp, cl, dl, tv};
    csr.request();
    // phase 2's saved activation of this lane group's first row, requested a phase early: the
    // row's graph node now (a halo row's comes from the slot), the row itself once the phase-1
    // tile requests are out (`mid`)
    const float* hsel = a.act == GCONV_ACT_SILU ? a.saved_h1 : a.saved_h;
    // (the halo id is only REQUESTED here: selecting the node — a use of the loaded value — in front of
    // the phase-1 tile requests put one more cold round trip, ~1.3 us, in front of every backward tile)
    const int lr_first = min(grp, tv.n1 - 1);
    const int hid_first = tv.hb[min(max(lr_first - nt, 0), kHaloNodes - 1)];
    f32x4 hfirst;
    const int np4 = nt * LPR;  // the P tile: np4 float4, the first blockDim of them via registers
    const f32x4 pfirst =
        ldg4(a.saved_p + static_cast<size_t>(ts) * FP + 4 * min(static_cast<int>(threadIdx.x), np4 - 1));
    ds1 += tile_rowsA<FP, NQ, NORM, true>(rga, a.g, wl, bufA, a.dis, a.want_ds ? a.saved_p : nullptr, sval, tv,
                                          a.d, rowb, nwaves, wave, lane, [&]() {
                                            csr.advance();
                                            const int gfirst = lr_first < nt ? ts + lr_first : hid_first;
                                            hfirst = ldg4(hsel + static_cast<size_t>(gfirst) * FP + 4 * sub);
                                          });
    csr.finish();
    // P goes to LDS transposed (rank-major): phase 4 reads 4 consecutive nodes of one rank row
    // with one 16-byte LDS load; rows between nt and the end of the last node block are zeros
    const int np4pad = ceil_div(nt, 16) * 16 * LPR;
    for (int i = threadIdx.x; i < np4pad; i += blockDim.x) {
      f32x4 pv = zero4f();
      if (i < np4) pv = i == static_cast<int>(threadIdx.x) ? pfirst : ldg4(a.saved_p + static_cast<size_t>(ts) * FP + 4 * i);
      const int lr = i / LPR, sb = i % LPR;
#pragma unroll
      for (int k = 0; k < 4; ++k) bufC[(4 * sb + k) * a.cap + lr] = pv[k];
    }
    lds_barrier();

    TILE_STAMP(2);
    // phase 4's first two (G, X) tile pairs are requested now: their latency passes under
    // phases 2 / 3. Unconditional requests (a conditional prefetch makes the compiler drain
    // vmcnt): a node block past the tile lies outside the buffers and costs no traffic.
    const int nb4 = ceil_div(nt, 16);
    auto load_gx = [&](int blk, float (&gq)[16], float (&xq)[16]) {
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        const int off = voff1 + (16 * blk + n) * rowb;
#ifdef GCONV_LAB_BWD_NO_GREAD   // lab build (wrong results): what phase 4 costs without its re-read of G
        gq[n] = 0.f;
#else
        gq[n] = buf_ld1(rg, off);
#endif
        xq[n] = buf_ld1(rx, off);
      }
    };
    constexpr bool DEEP = FP < 32 && GCONV_TILE_WGPC < 2;  // rank 17..32: no registers left for the second pair
    float ga[16], xa[16], gb[DEEP ? 16 : 1], xb[DEEP ? 16 : 1];
    load_gx(slot, ga, xa);
    if constexpr (DEEP) load_gx(slot + nslots, gb, xb);

    // ---- phase 2: dH1' = dis . (S^T dP . act') -> bufB, n1 rows ; bias gradient (own rows)
    for (int lr = grp; lr < tv.n1; lr += ngrp) {
      f32x4 v = tile_row_sum<FP, NORM>(bufA, rp, cl, dl, lr, sub);
      f32x4 hv = hfirst;
      if (lr != grp)   // tiles with more rows than lane groups
        hv = ldg4(hsel + static_cast<size_t>(tile_node(tv, lr)) * FP + 4 * sub);
      if (a.act == GCONV_ACT_RELU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = hv[k] > 0.f ? v[k] : 0.f;
      } else if (a.act == GCONV_ACT_SILU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= silu_grad(hv[k]);
      }
      if (lr < nt) bgrad += v;   // a halo row's share belongs to the tile that owns it
      if (NORM) v *= dl[lr];
      *reinterpret_cast<f32x4*>(bufB + lr * FP + 4 * sub) = v;
    }
    lds_barrier();

    TILE_STAMP(3);
    // ---- phase 3: T = S^T dH1 -> bufA (row-major: the dX operand) and bufD (rank-major:
    // the dW operand; rows up to the end of the last node block are zeros)
    for (int lr = grp; lr < ceil_div(nt, 16) * 16; lr += ngrp) {
      f32x4 v = zero4f();
      if (lr < nt) {
        v = tile_row_sum<FP, NORM>(bufB, rp, cl, dl, lr, sub);
        *reinterpret_cast<f32x4*>(bufA + lr * FP + 4 * sub) = v;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) bufD[(4 * sub + k) * a.cap + lr] = v[k];
    }
    lds_barrier();

    TILE_STAMP(4);
    // ---- phase 4 (owner waves): dX = T Wd + s G ; dWd^T += T^T X ; dWu^T += P^T G ;
    //      colsum(G) ; sum G.X — two (G, X) tile pairs per wave in flight
    // (everything requested so far has landed — the phases in between waited on younger
    // loads; saying so gives the loop one known state on entry and around its back edge)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    if (owner && slot < nb4) {   // (NT: dX with non-temporal / default-policy stores — see the forward kernel)
      const float skipf = a.skip ? sval : 0.f;
      auto step = [&](int b, float (&gv)[16], float (&xv)[16], int refill) {
        // dWd^T += T^T X, dWu^T += P^T G: one node per MFMA, the node's 4 rank values of tile mt
        // (the same in all 16 blocks) times this lane's column of the node's row
#pragma unroll
        for (int ng = 0; ng < 4; ++ng) {   // 4 nodes per 16-byte LDS read of a rank row
          f32x4 tq[MT], pq[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            tq[mt] = *reinterpret_cast<const f32x4*>(bufD + (4 * mt + l4) * a.cap + 16 * b + 4 * ng);
            pq[mt] = *reinterpret_cast<const f32x4*>(bufC + (4 * mt + l4) * a.cap + 16 * b + 4 * ng);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int n = 4 * ng + i;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              awd[mt] = mfma4(tq[mt][i], xv[n], awd[mt]);
              awu[mt] = mfma4(pq[mt][i], gv[n], awu[mt]);
            }
            cs += gv[n];
            dsx += gv[n] * xv[n];
          }
        }
        if constexpr (NEED_DX) {
          // dX[node][col] = sum_k T[node][k] Wd[k][col] + s G: 4 nodes per accumulator
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int lrc = min(16 * b + 4 * g4 + l4, nt - 1);   // (rows past the tile: dropped below)
            f32x4 tk[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) tk[mt] = *reinterpret_cast<const f32x4*>(bufA + lrc * FP + 4 * mt);
            f32x4 acc = zero4f();
#pragma unroll
            for (int k = 0; k < FP; ++k) acc = mfma4(tk[k >> 2][k & 3], wd[k], acc);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int n = 4 * g4 + v;
              buf_st1_p<NT>(rdx, voff1 + (16 * b + n) * rowb, acc[v] + skipf * gv[n]);
            }
          }
        }
        load_gx(refill, gv, xv);
      };
      if constexpr (DEEP) {
        for (int b = slot;;) {
          step(b, ga, xa, b + 2 * nslots);
          b += nslots;
          if (b >= nb4) break;
          step(b, gb, xb, b + 2 * nslots);
          b += nslots;
          if (b >= nb4) break;
        }
      } else {
        for (int b = slot; b < nb4; b += nslots) step(b, ga, xa, b + nslots);
      }
    }
    lds_barrier();  // bufA / bufB are reused by the next tile
    TILE_STAMP(5);
  }

  // ---- co-owners hand their accumulators to the chunk's primary owner through LDS (the
  // tile buffers are dead: the loop ends on a barrier); summed in slot order
  {
    constexpr int kSlab = (2 * FP + 1) * 64;  // floats per co-owner wave
    float* slabs = reinterpret_cast<float*>(lds4);
    if (owner && !primary) {
      float* sl = slabs + ex * kSlab + lane;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          sl[(4 * mt + v) * 64] = awd[mt][v];
          sl[(FP + 4 * mt + v) * 64] = awu[mt][v];
        }
      sl[2 * FP * 64] = cs;
    }
    lds_barrier();
    if (primary) {
      for (int s = 1; s < nslots; ++s) {
        const float* sl = slabs + ((s - 1) * nchunks + jr) * kSlab + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            awd[mt][v] += sl[(4 * mt + v) * 64];
            awu[mt][v] += sl[(FP + 4 * mt + v) * 64];
          }
        cs += sl[2 * FP * 64];
      }
    }
    lds_barrier();  // the slabs may overlap the reduction scratch used below
  }

  // ---- one partial row per workgroup (primary owners): lane = column, register = rank row
  // b_up . colsum(G) of this wave's chunk joins the scalar-gradient partial
  float bdot = 0.f;
  if (primary && colvalid && a.b_up) bdot = a.b_up[col1] * cs;
  if (primary && colvalid) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const size_t o = (static_cast<size_t>(pid) * FP + 4 * mt + v) * a.d + col1;
        a.part_wd[o] = awd[mt][v];
        a.part_wu[o] = awu[mt][v];
      }
    a.part_cs[static_cast<size_t>(pid) * a.d + col1] = cs;
  }
  // bias gradient and d(scalar): lanes -> wave -> workgroup, fixed order
#pragma unroll
  for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) bgrad[k] += __shfl_xor(bgrad[k], off);
  }
  float dsum = ds1 + (a.skip ? dsx : 0.f) + bdot;
  for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off);
  if (lane < LPR) {
#pragma unroll
    for (int k = 0; k < 4; ++k) red[wave * (FP + 2) + 4 * lane + k] = bgrad[k];
  }
  if (lane == 0) red[wave * (FP + 2) + FP] = dsum;
  lds_barrier();
  if (threadIdx.x < FP) {
    float v = 0.f;
    for (int w = 0; w < nwaves; ++w) v += red[w * (FP + 2) + threadIdx.x];
    a.part_bd[static_cast<size_t>(pid) * FP + threadIdx.x] = v;
  }
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int w = 0; w < nwaves; ++w) v += red[w * (FP + 2) + FP];
    a.part_ds[pid] = v;
  }
  TILE_STAMP(6);
}

// ===========================================================================
// launchers
// ===========================================================================
#if !defined(GCONV_TILE_ONLY_FP) || GCONV_TILE_ONLY_FP == 0
static int tile_nq(int d) {
  static const int kNq[] = {2, 4, 8, 12, 16, 19, 24, 32};
  const int need = ceil_div(d, 16);
  for (int v : kNq)
    if (need <= v) return v;
  return 0;
}

// Workgroup shapes. One workgroup per CU, walking its tiles; a tile's time is a chain of
// memory latencies, so a workgroup carries as many waves as its registers allow:
//   forward   nchunks * nslots waves (a wave owns a 64-column chunk and every nslots-th node
//             block in phase 4), up to 10 (8 for rank 17..32 and for d > 304, whose kernels
//             need more VGPRs)
//   backward  8 waves (the dW accumulators pin a chunk to one wave: nchunks owner waves)
static int tile_threads_fwd(int d, int fp) {
#if GCONV_TILE_WGPC >= 2
  const int nchunks = ceil_div(d, 64), cap = 5;
#else
  const int nchunks = ceil_div(d, 64), cap = (fp >= 32 || tile_nq(d) > 19) ? 8 : 10;
#endif
  int nslots = cap / nchunks;
  if (nslots < 1) nslots = 1;
  int w = nchunks * nslots;
  if (w < 4) w = 4;
  return 64 * w;
}
static int tile_threads_bwd(int d) {
  int w = ceil_div(d, 64);
  if (w < (GCONV_TILE_WGPC >= 2 ? 4 : 8)) w = GCONV_TILE_WGPC >= 2 ? 4 : 8;
  return 64 * w;
}

constexpr size_t kTileMaxLds = (GCONV_TILE_WGPC >= 2 ? 72 : 144) * 1024;

size_t tile_lds_bytes(int d, int fp, int tile_nodes_max, int tile_edges_max) {
  const int nq = tile_nq(d), ncb = fp >= 16 ? fp / 16 : 1;
  const int cap = ceil_div(tile_nodes_max > 0 ? tile_nodes_max : 1, 16) * 16;
  const size_t tiles =
      static_cast<size_t>(nq) * ncb * 64 * sizeof(f32x4) +              // weight fragments
      static_cast<size_t>(4) * cap * fp * sizeof(float) +                // rank-width tiles (backward: 4)
      8 * (fp + 2) * sizeof(float) +                                     // reduction scratch
      (static_cast<size_t>(2) * cap + 16 + (tile_edges_max > 0 ? tile_edges_max : 1)) * 4 + 64;  // rp, dl, cl
  // the backward's co-owner slabs reuse the same space after the last tile
  const size_t slabs = static_cast<size_t>(fp >= 32 ? 1 : 3) * (2 * fp + 1) * 64 * sizeof(float);
  return tiles > slabs ? tiles : slabs;
}

bool tile_path_ok(const gconv_graph_t* g, int d, int fp) {
  if (g->n_tiles <= 0 || !g->tiles_fwd.tiles || !g->tiles_fwd.lcol || !g->tiles_fwd.halo || !g->tiles_bwd.tiles ||
      !g->tiles_bwd.lcol || !g->tiles_bwd.halo)
    return false;
  // phase 1 addresses ALL rows of X / G through one buffer descriptor with 32-bit byte offsets
  if (static_cast<int64_t>(g->fwd.num_nodes) * d * 4 > 0x7FFF0000LL) return false;
  if (tile_nq(d) == 0 || fp > 32 || ceil_div(d, 64) > (GCONV_TILE_WGPC >= 2 ? 5 : 8)) return false;
  // wide rows at high rank do not fit the register budget of the fused kernels (they would
  // spill); the four-launch path is the better one there anyway
  if ((fp >= 32 && tile_nq(d) > 19) || (fp >= 16 && tile_nq(d) > 24)) return false;
  static const bool off = getenv("GCONV_NO_TILE_PATH") != nullptr;  // debugging / A-B knob
  if (off) return false;
  // one workgroup per CU: the tile state may take most of the CU's 160 KB (round 1 stopped at 64 KB,
  // which sent a batch with one 150-atom molecule at rank 16 down the multi-launch path)
  return tile_lds_bytes(d, fp, g->tile_nodes_max, g->tile_edges_max) <= kTileMaxLds;
}
#endif  // host side

#define GCONV_TILE_LDS_OPT_IN(KERNEL)                                                          \
  do {                                                                                         \
    if (lds > 48 * 1024)                                                                       \
      GCONV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL),                    \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds))); \
  } while (0)

template <int FP, int NQ>
static gconv_status_t launch_tile_fwd_t(const TileFwdArgs& a, int grid, int threads, size_t lds,
                                        hipStream_t stream, hipEvent_t done) {
  // `done` (gconv_arm_completion_event): recorded by the kernel's OWN completion signal — an event recorded behind the
  // launch is a barrier packet in front of the stream's next kernel (6.5 us of idle GPU in the whole-batch step)
#define GCONV_TILE_FWD_LAUNCH(NORM_, NT_)                                                                      \
  do {                                                                                                         \
    GCONV_TILE_LDS_OPT_IN((tile_fwd_kernel<FP, NQ, NORM_, NT_>));                                              \
    if (done)                                                                                                  \
      hipExtLaunchKernelGGL((tile_fwd_kernel<FP, NQ, NORM_, NT_>), dim3(grid), dim3(threads), lds, stream,     \
                            nullptr, done, 0, a);                                                              \
    else                                                                                                       \
      tile_fwd_kernel<FP, NQ, NORM_, NT_><<<grid, threads, lds, stream>>>(a);                                  \
  } while (0)
  if (a.dis) {
    if (a.store_nt) GCONV_TILE_FWD_LAUNCH(true, true); else GCONV_TILE_FWD_LAUNCH(true, false);
  } else {
    if (a.store_nt) GCONV_TILE_FWD_LAUNCH(false, true); else GCONV_TILE_FWD_LAUNCH(false, false);
  }
#undef GCONV_TILE_FWD_LAUNCH
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

template <int FP, int NQ>
static gconv_status_t launch_tile_bwd_t(const TileBwdArgs& a, int grid, int threads, size_t lds,
                                        hipStream_t stream) {
#define GCONV_TILE_BWD_LAUNCH(NORM_, DX_, NT_)                                       \
  do {                                                                               \
    GCONV_TILE_LDS_OPT_IN((tile_bwd_kernel<FP, NQ, NORM_, DX_, NT_>));               \
    tile_bwd_kernel<FP, NQ, NORM_, DX_, NT_><<<grid, threads, lds, stream>>>(a);     \
  } while (0)
  // (without dX there is no streamed output: one policy)
  if (a.dis) {
    if (!a.dx) GCONV_TILE_BWD_LAUNCH(true, false, true);
    else if (a.store_nt) GCONV_TILE_BWD_LAUNCH(true, true, true);
    else GCONV_TILE_BWD_LAUNCH(true, true, false);
  } else {
    if (!a.dx) GCONV_TILE_BWD_LAUNCH(false, false, true);
    else if (a.store_nt) GCONV_TILE_BWD_LAUNCH(false, true, true);
    else GCONV_TILE_BWD_LAUNCH(false, true, false);
  }
#undef GCONV_TILE_BWD_LAUNCH
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

// One entry point per padded rank: every (Fp, NQ) pair below is a compiled kernel. The build
// compiles this file once per rank (-DGCONV_TILE_ONLY_FP=4|8|16|32: the kernels of that rank and
// its entry points) plus once for the host side (=0), so that the five pieces compile in
// parallel; without the define everything lands in one translation unit.
#define GCONV_TILE_NQ_SWITCH(FN, FP_, ARGS)                                             \
  switch (nq) {                                                                         \
    case 2:  return FN<FP_, 2> ARGS;  case 4:  return FN<FP_, 4> ARGS;                  \
    case 8:  return FN<FP_, 8> ARGS;  case 12: return FN<FP_, 12> ARGS;                 \
    case 16: return FN<FP_, 16> ARGS; case 19: return FN<FP_, 19> ARGS;                 \
    case 24: return FN<FP_, 24> ARGS; case 32: return FN<FP_, 32> ARGS;                 \
    default: return GCONV_E_UNSUPPORTED;                                                \
  }
template <int FP>
gconv_status_t launch_tile_fwd_fp(const TileFwdArgs& a, int nq, int grid, int threads, size_t lds, hipStream_t stream,
                                 hipEvent_t done);
template <int FP>
gconv_status_t launch_tile_bwd_fp(const TileBwdArgs& a, int nq, int grid, int threads, size_t lds, hipStream_t stream);
#define GCONV_TILE_DEFINE_FP(FP_)                                                                          \
  template <>                                                                                              \
  gconv_status_t launch_tile_fwd_fp<FP_>(const TileFwdArgs& a, int nq, int grid, int threads, size_t lds,  \
                                         hipStream_t stream, hipEvent_t done) {                            \
    GCONV_TILE_NQ_SWITCH(launch_tile_fwd_t, FP_, (a, grid, threads, lds, stream, done))                    \
  }                                                                                                        \
  template <>                                                                                              \
  gconv_status_t launch_tile_bwd_fp<FP_>(const TileBwdArgs& a, int nq, int grid, int threads, size_t lds,  \
                                         hipStream_t stream) {                                             \
    GCONV_TILE_NQ_SWITCH(launch_tile_bwd_t, FP_, (a, grid, threads, lds, stream))                          \
  }
#if !defined(GCONV_TILE_ONLY_FP)
GCONV_TILE_DEFINE_FP(4) GCONV_TILE_DEFINE_FP(8) GCONV_TILE_DEFINE_FP(16) GCONV_TILE_DEFINE_FP(32)
#elif GCONV_TILE_ONLY_FP > 0
GCONV_TILE_DEFINE_FP(GCONV_TILE_ONLY_FP)
#endif

#if !defined(GCONV_TILE_ONLY_FP) || GCONV_TILE_ONLY_FP == 0
#define GCONV_TILE_DISPATCH(FN, ARGS)                                                   \
  switch (a.fp) {                                                                       \
    case 4: return FN<4> ARGS;   case 8: return FN<8> ARGS;                             \
    case 16: return FN<16> ARGS; case 32: return FN<32> ARGS;                           \
    default: return GCONV_E_UNSUPPORTED;                                                \
  }

// Half the workgroups walk their tiles smallest first (tile_order_index), with at least two passes only; `walk`
// (gconv_adapter_params_t::walk for the forward, its complement for the backward) picks the half.
// GCONV_TILE_STAGGER=0 switches it off (A/B knob, read once).
static int tile_stagger(int n_tiles, int grid, int walk) {
  static const char* v = getenv("GCONV_TILE_STAGGER");
  const int mode = v ? atoi(v) : GCONV_TILE_STAGGER_DEFAULT;
  if (mode == 0 || n_tiles < (mode >= 2 ? grid + 1 : 2 * grid)) return 0;   // (mode 2: lab — also with one ragged second pass)
  return walk ? 2 : 1;   // the odd / the even workgroups (tile_walk_reversed)
}

// Output rows (out, dX) non-temporal? For small batches yes, as since round 1 (512 molecules, 16 MB per [N, d]
// tensor: the step is 1 % faster with nt). From 24 MB on the default policy: the next kernel reads these rows, and the
// step runs faster with them left in the caches — 1024 molecules (31 MB) -3.8 %, 2048 -7.7 %, 4096 -3.4 ... -4.2 %
// (tools/lab/r05_midsize.sh, r05_storepolicy.sh); the kernel ALONE is slower that way, which is all round 1 measured.
// GCONV_TILE_STORE_NT=0 / 1 forces.
constexpr int64_t kStoreDefaultPolicyBytes = 24 << 20;
static int tile_store_nt(int n, int d) {
  static const char* v = getenv("GCONV_TILE_STORE_NT");
  if (v) return v[0] == '1';
  return static_cast<int64_t>(n) * d * 4 >= kStoreDefaultPolicyBytes ? 0 : 1;
}

int tile_grid(int n_tiles) {  // persistent: one workgroup per CU
  const int kMax = launch_cus() * (GCONV_TILE_WGPC >= 2 ? 2 : 1);
  return n_tiles < kMax ? (n_tiles > 0 ? n_tiles : 1) : kMax;
}

gconv_status_t launch_tile_fwd(TileFwdArgs a, int n_tiles, int tile_nodes_max,
                               int tile_edges_max, hipStream_t stream, void* done_event) {
  const int nq = tile_nq(a.d);
  a.n_tiles = n_tiles;
  n_tiles = tile_grid(n_tiles);
  a.stagger = tile_stagger(a.n_tiles, n_tiles, a.stagger);   // (in: the caller's walk bit)
  a.npass = ceil_div(a.n_tiles, n_tiles);
  a.store_nt = tile_store_nt(a.n, a.d);
  a.cap = ceil_div(tile_nodes_max > 0 ? tile_nodes_max : 1, 16) * 16;
  const size_t lds = tile_lds_bytes(a.d, a.fp, tile_nodes_max, tile_edges_max);
  const int threads = tile_threads_fwd(a.d, a.fp);
  GCONV_TILE_DISPATCH(launch_tile_fwd_fp, (a, nq, n_tiles, threads, lds, stream, static_cast<hipEvent_t>(done_event)))
}

gconv_status_t launch_tile_bwd(TileBwdArgs a, int n_tiles, int tile_nodes_max,
                               int tile_edges_max, hipStream_t stream) {
  const int nq = tile_nq(a.d);
  a.n_tiles = n_tiles;
  n_tiles = tile_grid(n_tiles);
  a.stagger = tile_stagger(a.n_tiles, n_tiles, a.stagger);   // (in: the caller's walk bit)
  a.npass = ceil_div(a.n_tiles, n_tiles);
  a.store_nt = tile_store_nt(a.n, a.d);
  a.cap = ceil_div(tile_nodes_max > 0 ? tile_nodes_max : 1, 16) * 16;
  const size_t lds = tile_lds_bytes(a.d, a.fp, tile_nodes_max, tile_edges_max);
  const int threads = tile_threads_bwd(a.d);
  GCONV_TILE_DISPATCH(launch_tile_bwd_fp, (a, nq, n_tiles, threads, lds, stream))
}
#endif  // host side

}  // namespace gconv
