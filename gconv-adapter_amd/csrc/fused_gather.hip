// K2' + K3 and B3 + B4 of the general path as ONE launch each (gfx950).
//
// Replaces, together with spmm.hip / gemm.hip: the second `propagate` of the adapter
// (conv_up of /root/reference/src/finetune/gconv_adapter.py:92, reassociated to rank r), the
// residual / bias / scalar tail (:94-95,105-106) and, in the backward, the transposed
// propagation that feeds dX / dW_down.
//
// Why fuse: P = S H2' is needed row by row by the up-projection (a per-row dependency, unlike
// K2 -> K2', which needs ALL rows of H2'). A workgroup requests its X tile, gathers the P rows
// of the same nodes meanwhile, parks them in LDS (and in HBM for the backward) and multiplies:
// one launch and one kernel boundary less, which is what a Cora / PubMed-sized graph is made of
// (forward 21 -> 15 us). On graphs with hub rows the separate launches stay faster at every size
// — see fused_gather_enabled(). Rows longer than GCONV_LONG_ROW edges keep their segment path (a small launch of
// the SpMM's segment blocks in front); every other row is summed by row_gather_sum — the same
// code, hence the same bits, as the stand-alone SpMM.
#include <cstdlib>

#include "fused_gather.h"
#include "spmm.h"

namespace gconv {

// Read per call: tests and tools flip the knobs inside one process. Measured (tools/lab/
// fuse_threshold.py, tools/fusebench.py): on uniform low-degree graphs of Cora / PubMed size the
// launch that disappears is what counts (forward 21 -> 15 us, backward 5 + 2 -> 4 + 1 launches); on
// power-law graphs (S-arxiv's generator at 8 k ... 170 k nodes, d = 128) the separate launches win
// at every size (8 k nodes: 29 against 34 us forward; 170 k: 117 against 125) — a workgroup waits
// at its barrier for the wave with the longest rows, a stand-alone SpMM wave only delays itself.
// On uniform graphs (tools/lab/fuse_uniform.py) the fused launches win while the [N, d] streams are
// small — d = 32 up to 65 k nodes, d = 128 up to ~20 k, d = 300 up to ~8 k — and lose beyond
// (d = 300, 40 k nodes: 45 -> 50 us forward, 70 -> 80 backward): one launch less is worth a few
// microseconds, not a pipeline stall per super-block. Hence: no row above GCONV_LONG_ROW edges AND
// N * d <= 2.5 M elements.
bool fused_gather_enabled(const gconv_csr_t* graph, int hidden) {
  const char* v = std::getenv("GCONV_NO_FUSED_GATHER");
  if (!(v == nullptr || v[0] == '\0' || v[0] == '0')) return false;
  const char* m = std::getenv("GCONV_FUSED_GATHER_MAX_NODES");
  if (m) return graph->num_nodes <= std::atoi(m);
  return graph->n_long == 0 && static_cast<int64_t>(graph->num_nodes) * hidden <= 2500000;
}

static int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return v ? std::atoi(v) : dflt;
}

// Gathers the rank-width rows of one node super-block (SB = 16 * nslots rows starting at sb0)
// into the LDS tile `tile` ([SB][FP + 4]); short rows are also stored to `store` when STORE.
//   W = waves of the workgroup; every wave takes RPW rows per pass.
template <int FP, bool NORM, bool STORE>
__device__ __forceinline__ void gather_block_rows(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                  const float* __restrict__ dis, const float* __restrict__ H,
                                                  float* store, float* tile, int sb0, int SB, int n,
                                                  int w, int W, int lane) {
  constexpr int LPR = FP / 4, RPW = kWave / LPR, PST = FP + 4;
  const int sub = lane % LPR, grp = lane / LPR;
  for (int pass = 0; pass * W * RPW < SB; ++pass) {   // uniform
    const int ridx = (pass * W + w) * RPW + grp;
    const int row = sb0 + ridx;
    const bool inb = ridx < SB && row < n;
    const int rowc = min(row, n - 1);
    const int e0 = rowptr[rowc], e1 = rowptr[rowc + 1];
    const bool is_long = (e1 - e0) > GCONV_LONG_ROW;
    const int deg = (inb && !is_long) ? e1 - e0 : 0;
    const f32x4 sum = row_gather_sum<FP>(H, col, e0, e1, deg, rowc, sub, grp, false, nullptr);
    const size_t off = static_cast<size_t>(rowc) * FP + 4 * sub;
    f32x4 v = sum;
    if (NORM) v = (sum + ldg4(H + off)) * dis[rowc];   // == spmm_epilogue<F, true, EPI_STORE>
    if (__any(inb && is_long)) {                       // rare: completed by the segment launch
      const f32x4 pl = ldg4(store + off);
      v = is_long ? pl : v;
    }
    if (inb) {
      if (STORE && !is_long) stg4(store + off, v);
      *reinterpret_cast<f32x4*>(tile + ridx * PST + 4 * sub) = v;
    }
  }
}

// A fragments of the LDS tile for node row `r` of the super-block: frag[s] = tile[r][rank_k(s, t)]
template <int FP>
__device__ __forceinline__ void tile_rank_frag(const float* tile, int r, int t, float (&frag)[FP / 4]) {
  constexpr int PST = FP + 4;
  const float* p = tile + r * PST;
  if constexpr (FP >= 16) {
#pragma unroll
    for (int q = 0; q < FP / 16; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + 16 * q + 4 * t);
      frag[4 * q + 0] = v[0]; frag[4 * q + 1] = v[1];
      frag[4 * q + 2] = v[2]; frag[4 * q + 3] = v[3];
    }
  } else {
#pragma unroll
    for (int s = 0; s < FP / 4; ++s) frag[s] = p[4 * s + t];
  }
}

// ===========================================================================
// K2' + K3
// ===========================================================================
template <int FP, bool NORM, int TB>
__global__ __launch_bounds__(TB) void up_gather_kernel(UpGatherArgs a, int nchunks, int nslots) {
  extern __shared__ float ptile[];   // [16 nslots][FP + 4]
  constexpr int KS = FP / 4;
  const UpArgs& u = a.up;
  const int lane = threadIdx.x & 63, w = wave_id();
  const int J = w % nchunks, slot = w / nchunks;
  const int jj = lane & 15, t = lane >> 4;
  const int colbase = 64 * J + 4 * jj;
  const bool colvalid = colbase < u.d;
  const float sval = u.scalar ? u.scalar[0] : 1.f;
  const int W = nchunks * nslots, SB = 16 * nslots;

  // B fragments: wb[s][c] = W_up[colbase + c][rank_k(s, t)]
  float wb[KS][4];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = rank_k<FP>(s, t);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      wb[s][c] = (colvalid && k < u.r) ? u.W_up[static_cast<size_t>(colbase + c) * u.r + k] : 0.f;
  }
  f32x4 bu = f32x4{0.f, 0.f, 0.f, 0.f};
  if (colvalid && u.b_up) bu = ldg4(u.b_up + colbase);

  const int nsb = ceil_div(u.n, SB);
  for (int sb = blockIdx.x; sb < nsb; sb += gridDim.x) {
    const int i0 = sb * SB + slot * 16;
    // the [16, 64] X tile of this wave first: it streams from HBM while the P rows are gathered
    f32x4 xv[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) xv[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (u.skip) load_tile_L(u.X, i0, t, u.n, u.d, colbase, colvalid, xv);   // uniform branch
    gather_block_rows<FP, NORM, true>(a.rowptr, a.col, a.dis, a.H, a.P, ptile, sb * SB, SB, u.n, w, W, lane);
    __syncthreads();
    float pa[KS];
    tile_rank_frag<FP>(ptile, slot * 16 + jj, t, pa);
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = mfma16(pa[s], wb[s][c], acc[c]);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = i0 + 4 * t + m;
      if (colvalid && row < u.n) {
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = sval * (acc[c][m] + bu[c] + xv[m][c]);
        stg4(u.out + static_cast<size_t>(row) * u.d + colbase, o);
      }
    }
    __syncthreads();   // the tile is rewritten by the next super-block
  }
}

template <int FP, bool NORM>
static gconv_status_t launch_up_gather_t(const UpGatherArgs& a, int grid, int threads, size_t lds, int nchunks,
                                         int nslots, hipStream_t stream) {
  if (threads <= 256) up_gather_kernel<FP, NORM, 256><<<grid, threads, lds, stream>>>(a, nchunks, nslots);
  else if (threads <= 512) up_gather_kernel<FP, NORM, 512><<<grid, threads, lds, stream>>>(a, nchunks, nslots);
  else up_gather_kernel<FP, NORM, 1024><<<grid, threads, lds, stream>>>(a, nchunks, nslots);
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

gconv_status_t launch_up_gather(const UpGatherArgs& a, hipStream_t stream) {
  const UpArgs& u = a.up;
  if (u.n == 0) return GCONV_OK;
  const int nchunks = ceil_div(u.d, 64);
  if (nchunks > 16) return GCONV_E_UNSUPPORTED;
  const int nslots = chunk_slots(nchunks, 4);
  const int SB = 16 * nslots, nsb = ceil_div(u.n, SB);
  // persistent: a few workgroups per CU so that one workgroup's gather latency hides behind
  // the others' streams (GCONV_UPG_WGPC: measured at S-arxiv)
  const int wgpc = env_int("GCONV_UPG_WGPC", 3);
  int grid = nsb < wgpc * launch_cus() ? nsb : wgpc * launch_cus();
  const int threads = 64 * nchunks * nslots;
  const size_t lds = static_cast<size_t>(SB) * (u.fp + 4) * sizeof(float);
#define GCONV_UPG_CASE(FP_)                                                                         \
  case FP_:                                                                                         \
    return a.dis ? launch_up_gather_t<FP_, true>(a, grid, threads, lds, nchunks, nslots, stream)    \
                 : launch_up_gather_t<FP_, false>(a, grid, threads, lds, nchunks, nslots, stream);
  switch (u.fp) {
    GCONV_UPG_CASE(4) GCONV_UPG_CASE(8) GCONV_UPG_CASE(16) GCONV_UPG_CASE(32) GCONV_UPG_CASE(64)
  }
#undef GCONV_UPG_CASE
  return GCONV_E_UNSUPPORTED;
}

// ===========================================================================
// B3 + B4
// ===========================================================================
template <int FP, bool NORM, bool NEED_DX, int TB>
__global__ __launch_bounds__(TB) void bwd_gather_kernel(BwdGatherArgs g, int nchunks, int nslots) {
  extern __shared__ float smem[];   // [16 nslots][FP + 4] T tile | B4's slot reduction buffer
  constexpr int KS = FP / 4;
  constexpr int NCB = FP >= 16 ? FP / 16 : 1;
  constexpr int PST = FP + 4;
  const BwdMainArgs& a = g.main;
  const int lane = threadIdx.x & 63, w = wave_id();
  const int J = w % nchunks, slot = w / nchunks;
  const int jj = lane & 15, t = lane >> 4;
  const int colbase = 64 * J + 4 * jj;
  const bool colvalid = colbase < a.d;
  const float sval = a.scalar ? a.scalar[0] : 1.f;
  const int W = nchunks * nslots, SB = 16 * nslots;
  float* ttile = smem;
  float* red = smem + SB * PST;

  // B fragments for dX: wd[s][c] = W_down[rank_k(s, t)][colbase + c]
  float wd[NEED_DX ? KS : 1][4];
  if (NEED_DX) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = rank_k<FP>(s, t);
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (colvalid && k < a.r) v = ldg4(a.W_down + static_cast<size_t>(k) * a.d + colbase);
#pragma unroll
      for (int c = 0; c < 4; ++c) wd[s][c] = v[c];
    }
  }
  f32x4 awd[NCB][4], awu[NCB][4];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int c = 0; c < 4; ++c) awd[cb][c] = awu[cb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
  float dsx = 0.f;

  const int nsb = ceil_div(a.n, SB);
  for (int sb = blockIdx.x; sb < nsb; sb += gridDim.x) {
    const int i0 = sb * SB + slot * 16;
    // this wave's G / X tiles and P fragments first: they fly while the T rows are gathered
    f32x4 gv[4], xv[4];
    load_tile_L(a.G, i0, t, a.n, a.d, colbase, colvalid, gv);
    load_tile_L(a.X, i0, t, a.n, a.d, colbase, colvalid, xv);
    float ap[4][NCB];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = i0 + 4 * t + m;
      const size_t rb = static_cast<size_t>(min(row, a.n - 1)) * FP;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int rc = 16 * cb + jj;
        const float vp = a.P[rb + min(rc, FP - 1)];
        ap[m][cb] = (row < a.n && rc < FP) ? vp : 0.f;
      }
    }
    gather_block_rows<FP, NORM, false>(g.rowptr, g.col, g.dis, g.H, const_cast<float*>(g.T), ttile, sb * SB, SB,
                                       a.n, w, W, lane);
    __syncthreads();
    float at[4][NCB];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = i0 + 4 * t + m;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int rc = 16 * cb + jj;
        const float vt = ttile[(slot * 16 + 4 * t + m) * PST + min(rc, FP - 1)];
        at[m][cb] = (row < a.n && rc < FP) ? vt : 0.f;   // rows past n were never written
      }
    }
    float ta[KS];
    if constexpr (NEED_DX) tile_rank_frag<FP>(ttile, slot * 16 + jj, t, ta);

    // contraction over the 16 nodes: k-step m covers nodes i0 + 4 t + m  (== bwd_main_kernel)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          awd[cb][c] = mfma16(at[m][cb], xv[m][c], awd[cb][c]);
          awu[cb][c] = mfma16(ap[m][cb], gv[m][c], awu[cb][c]);
        }
      }
      cs += gv[m];
      if (a.skip) dsx = dot4_acc(gv[m], xv[m], dsx);
    }
    if (NEED_DX) {
      f32x4 acc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = mfma16(ta[s], wd[s][c], acc[c]);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int row = i0 + 4 * t + m;
        if (colvalid && row < a.n) {
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = acc[c][m] + (a.skip ? sval * gv[m][c] : 0.f);
          stg4(a.dX + static_cast<size_t>(row) * a.d + colbase, o);
        }
      }
    }
    __syncthreads();   // the T tile is rewritten by the next super-block
  }
  bwd_main_tail<FP>(a, awd, awu, cs, dsx, red, nchunks, nslots, J, slot, jj, t, lane, w, colbase, colvalid);
}

template <int FP, bool NORM>
static gconv_status_t launch_bwd_gather_t(const BwdGatherArgs& a, const BwdMainGeom& g, hipStream_t stream) {
  const int threads = 64 * g.nchunks * g.nslots;
  const size_t lds = static_cast<size_t>(16 * g.nslots) * (FP + 4) * sizeof(float) +
                     bwd_main_lds_bytes(g.nchunks, g.nslots, FP);
  if (lds > 160 * 1024) return GCONV_E_UNSUPPORTED;
#define GCONV_BWG_LAUNCH(DX, TB_)                                                                      \
  do {                                                                                                 \
    if (lds > 48 * 1024)                                                                               \
      GCONV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_gather_kernel<FP, NORM, DX, TB_>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds))); \
    bwd_gather_kernel<FP, NORM, DX, TB_><<<g.grid, threads, lds, stream>>>(a, g.nchunks, g.nslots);    \
  } while (0)
  if (a.main.dX) {
    if (threads <= 256) GCONV_BWG_LAUNCH(true, 256);
    else if (threads <= 512) GCONV_BWG_LAUNCH(true, 512);
    else GCONV_BWG_LAUNCH(true, 1024);
  } else {
    if (threads <= 256) GCONV_BWG_LAUNCH(false, 256);
    else if (threads <= 512) GCONV_BWG_LAUNCH(false, 512);
    else GCONV_BWG_LAUNCH(false, 1024);
  }
#undef GCONV_BWG_LAUNCH
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

gconv_status_t launch_bwd_gather(const BwdGatherArgs& a, hipStream_t stream) {
  const BwdMainGeom g = bwd_main_geom(a.main.n, a.main.d);
  if (g.nchunks > 16) return GCONV_E_UNSUPPORTED;
#define GCONV_BWG_CASE(FP_)                                                              \
  case FP_:                                                                              \
    return a.dis ? launch_bwd_gather_t<FP_, true>(a, g, stream) : launch_bwd_gather_t<FP_, false>(a, g, stream);
  switch (a.main.fp) {
    GCONV_BWG_CASE(4) GCONV_BWG_CASE(8) GCONV_BWG_CASE(16) GCONV_BWG_CASE(32) GCONV_BWG_CASE(64)
  }
#undef GCONV_BWG_CASE
  return GCONV_E_UNSUPPORTED;
}

}  // namespace gconv
