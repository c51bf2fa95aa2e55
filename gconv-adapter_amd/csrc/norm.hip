// Next to the path: the adapter's `layer_norm` tail on the device.
//
// With normalization='layer_norm' the reference finishes the adapter with
//   out = LayerNorm(out) ; out = out * scalar        (gconv_adapter.py:101-102, 105-106)
// i.e. two more element-wise passes over [N, d] forward and four to five backward (LayerNorm's
// dx and its dgamma / dbeta reduction, the scalar multiply's two products and its reduction).
// Here: ONE pass forward (row statistics in registers, normalise, affine, scale) and one pass +
// a small fixed-order reduction backward. One wave per row, a row of d floats as float4 lanes
// (d = 300: 75 lanes, two passes); the backward's waves are persistent over rows so the column
// sums for dgamma / dbeta (and the scalar's gradient) accumulate in registers and leave the
// workgroup once, as one partial row. No atomics: results are bitwise reproducible.
#include "common.h"

namespace gconv {

constexpr int kLnPasses = 4;   // hidden <= 1024

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

template <int NP>
__global__ __launch_bounds__(256) void ln_scale_fwd_kernel(const float* __restrict__ z,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ scalar,
                                                           float* __restrict__ out,
                                                           float* __restrict__ stats, int n, int d,
                                                           float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave_id();
  if (row >= n) return;   // wave-uniform
  const int nq = d >> 2;
  const float s = scalar ? scalar[0] : 1.f;
  f32x4 v[NP];
  float sum = 0.f;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int q = lane + 64 * p;
    v[p] = q < nq ? ldg4(z + static_cast<size_t>(row) * d + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    sum += (v[p][0] + v[p][1]) + (v[p][2] + v[p][3]);
  }
  const float mean = wave_sum(sum) / static_cast<float>(d);
  float sq = 0.f;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int q = lane + 64 * p;
    if (q < nq) {
      const f32x4 c = v[p] - mean;
      sq += (c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]);
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / static_cast<float>(d) + eps);   // biased variance, as torch
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int q = lane + 64 * p;
    if (q < nq) {
      f32x4 y = (v[p] - mean) * rstd;
      if (gamma) y = y * ldg4(gamma + 4 * q);
      if (beta) y = y + ldg4(beta + 4 * q);
      stg4(out + static_cast<size_t>(row) * d + 4 * q, y * s);
    }
  }
  if (lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
}

// partial row layout: [dgamma (d) | dbeta (d) | dscalar (1)], one row per workgroup
template <int NP>
__global__ __launch_bounds__(256) void ln_scale_bwd_kernel(const float* __restrict__ g,
                                                           const float* __restrict__ z,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ scalar,
                                                           float* __restrict__ dz,
                                                           float* __restrict__ part, int n, int d) {
  __shared__ f32x4 red[3][64 * NP];   // waves 1..3 hand their column sums to wave 0
  __shared__ float red_s[4];
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int nq = d >> 2;
  const float s = scalar ? scalar[0] : 1.f;
  f32x4 gam[NP], bet[NP], dgam[NP], dbet[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int q = lane + 64 * p;
    gam[p] = (gamma && q < nq) ? ldg4(gamma + 4 * q) : f32x4{1.f, 1.f, 1.f, 1.f};
    bet[p] = (beta && q < nq) ? ldg4(beta + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    dgam[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    dbet[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float ds = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < n; row += gridDim.x * 4) {   // wave-uniform
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    f32x4 xh[NP], dyg[NP];
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int q = lane + 64 * p;
      if (q < nq) {
        const f32x4 gv = ldg4(g + static_cast<size_t>(row) * d + 4 * q);
        xh[p] = (ldg4(z + static_cast<size_t>(row) * d + 4 * q) - mean) * rstd;
        const f32x4 y = xh[p] * gam[p] + bet[p];      // LayerNorm output (before the scalar)
        const f32x4 dy = gv * s;
        ds += (gv[0] * y[0] + gv[1] * y[1]) + (gv[2] * y[2] + gv[3] * y[3]);
        dgam[p] += dy * xh[p];
        dbet[p] += dy;
        dyg[p] = dy * gam[p];
        a += (dyg[p][0] + dyg[p][1]) + (dyg[p][2] + dyg[p][3]);
        const f32x4 t = dyg[p] * xh[p];
        b += (t[0] + t[1]) + (t[2] + t[3]);
      } else {
        xh[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        dyg[p] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    a = wave_sum(a) / static_cast<float>(d);
    b = wave_sum(b) / static_cast<float>(d);
    if (dz) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int q = lane + 64 * p;
        if (q < nq) stg4(dz + static_cast<size_t>(row) * d + 4 * q, (dyg[p] - a - xh[p] * b) * rstd);
      }
    }
  }
  // one partial row per workgroup, waves added in wave order
  ds = wave_sum(ds);
  if (wave > 0) {
#pragma unroll
    for (int p = 0; p < NP; ++p) red[wave - 1][lane + 64 * p] = dgam[p];
  }
  if (lane == 0) red_s[wave] = ds;
  __syncthreads();
  float* prow = part + static_cast<size_t>(blockIdx.x) * (2 * d + 1);
  if (wave == 0) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int q = lane + 64 * p;
      f32x4 t = dgam[p];
      for (int w = 0; w < 3; ++w) t += red[w][lane + 64 * p];
      if (q < nq) {
        for (int k = 0; k < 4; ++k) prow[4 * q + k] = t[k];
      }
    }
    if (lane == 0) prow[2 * d] = (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]);
  }
  __syncthreads();
  if (wave > 0) {
#pragma unroll
    for (int p = 0; p < NP; ++p) red[wave - 1][lane + 64 * p] = dbet[p];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int q = lane + 64 * p;
      f32x4 t = dbet[p];
      for (int w = 0; w < 3; ++w) t += red[w][lane + 64 * p];
      if (q < nq) {
        for (int k = 0; k < 4; ++k) prow[d + 4 * q + k] = t[k];
      }
    }
  }
}

// column sums of the partial rows in a fixed order: a block owns 16 columns, its 16 row groups
// walk the partial rows 16 apart (4 loads in flight each) and are added in group order
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ part, int nparts,
                                                          int d, float* __restrict__ g_gamma,
                                                          float* __restrict__ g_beta,
                                                          float* __restrict__ g_scalar) {
  __shared__ float red[16][17];
  const int width = 2 * d + 1;
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < width) {
    int r = rg;
    for (; r + 48 < nparts; r += 64) {
      a0 += part[static_cast<size_t>(r) * width + c];
      a1 += part[static_cast<size_t>(r + 16) * width + c];
      a2 += part[static_cast<size_t>(r + 32) * width + c];
      a3 += part[static_cast<size_t>(r + 48) * width + c];
    }
    for (; r < nparts; r += 16) a0 += part[static_cast<size_t>(r) * width + c];
  }
  red[rg][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rg != 0 || c >= width) return;
  float v = 0.f;
  for (int k = 0; k < 16; ++k) v += red[k][cl];
  if (c < d) {
    if (g_gamma) g_gamma[c] = v;
  } else if (c < 2 * d) {
    if (g_beta) g_beta[c - d] = v;
  } else if (g_scalar) {
    g_scalar[0] = v;
  }
}

static int ln_bwd_grid(int n) { return min(ceil_div(n > 0 ? n : 1, 4), launch_cus() * 2); }

}  // namespace gconv

using namespace gconv;

static gconv_status_t ln_check(int32_t n, int32_t d) {
  if (n < 0) return GCONV_E_INVALID_ARG;
  if (d <= 0 || d % 4 != 0 || d > 256 * kLnPasses) return GCONV_E_UNSUPPORTED;
  return GCONV_OK;
}

extern "C" size_t gconv_layer_norm_scale_bwd_workspace_bytes(int32_t num_nodes, int32_t hidden) {
  if (ln_check(num_nodes, hidden) != GCONV_OK) return 0;
  return align_up(static_cast<size_t>(ln_bwd_grid(num_nodes)) * (2 * hidden + 1) * sizeof(float) + 16, 256);
}

extern "C" gconv_status_t gconv_layer_norm_scale_fwd_f32(const float* z, const float* gamma,
                                                         const float* beta, const float* scalar,
                                                         float* out, float* stats, int32_t num_nodes,
                                                         int32_t hidden, float eps,
                                                         gconv_stream_t stream_) {
  const gconv_status_t st = ln_check(num_nodes, hidden);
  if (st != GCONV_OK) return st;
  if (num_nodes == 0) return GCONV_OK;
  if (!z || !out || !stats) return GCONV_E_INVALID_ARG;
  if (!aligned16(z) || !aligned16(out) || (gamma && !aligned16(gamma)) || (beta && !aligned16(beta)))
    return GCONV_E_ALIGNMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int grid = ceil_div(num_nodes, 4);
  switch (ceil_div(hidden, 256)) {
    case 1: ln_scale_fwd_kernel<1><<<grid, 256, 0, stream>>>(z, gamma, beta, scalar, out, stats, num_nodes, hidden, eps); break;
    case 2: ln_scale_fwd_kernel<2><<<grid, 256, 0, stream>>>(z, gamma, beta, scalar, out, stats, num_nodes, hidden, eps); break;
    case 3: ln_scale_fwd_kernel<3><<<grid, 256, 0, stream>>>(z, gamma, beta, scalar, out, stats, num_nodes, hidden, eps); break;
    default: ln_scale_fwd_kernel<4><<<grid, 256, 0, stream>>>(z, gamma, beta, scalar, out, stats, num_nodes, hidden, eps); break;
  }
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

extern "C" gconv_status_t gconv_layer_norm_scale_bwd_f32(const float* grad_out, const float* z,
                                                         const float* stats, const float* gamma,
                                                         const float* beta, const float* scalar,
                                                         float* grad_z, float* grad_gamma,
                                                         float* grad_beta, float* grad_scalar,
                                                         int32_t num_nodes, int32_t hidden,
                                                         void* workspace, size_t workspace_bytes,
                                                         gconv_stream_t stream_) {
  const gconv_status_t st = ln_check(num_nodes, hidden);
  if (st != GCONV_OK) return st;
  if (!workspace || workspace_bytes < gconv_layer_norm_scale_bwd_workspace_bytes(num_nodes, hidden))
    return GCONV_E_WORKSPACE;
  if (num_nodes > 0 && (!grad_out || !z || !stats)) return GCONV_E_INVALID_ARG;
  if (!aligned16(grad_out) || !aligned16(z) || (grad_z && !aligned16(grad_z)) ||
      (gamma && !aligned16(gamma)) || (beta && !aligned16(beta)) || !aligned16(workspace))
    return GCONV_E_ALIGNMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  float* part = static_cast<float*>(workspace);
  const int grid = num_nodes > 0 ? ln_bwd_grid(num_nodes) : 0;
  if (grid > 0) {
    switch (ceil_div(hidden, 256)) {
      case 1: ln_scale_bwd_kernel<1><<<grid, 256, 0, stream>>>(grad_out, z, stats, gamma, beta, scalar, grad_z, part, num_nodes, hidden); break;
      case 2: ln_scale_bwd_kernel<2><<<grid, 256, 0, stream>>>(grad_out, z, stats, gamma, beta, scalar, grad_z, part, num_nodes, hidden); break;
      case 3: ln_scale_bwd_kernel<3><<<grid, 256, 0, stream>>>(grad_out, z, stats, gamma, beta, scalar, grad_z, part, num_nodes, hidden); break;
      default: ln_scale_bwd_kernel<4><<<grid, 256, 0, stream>>>(grad_out, z, stats, gamma, beta, scalar, grad_z, part, num_nodes, hidden); break;
    }
    GCONV_LAUNCH_CHECK();
  }
  if (grad_gamma || grad_beta || grad_scalar) {
    ln_finalize_kernel<<<ceil_div(2 * hidden + 1, 16), 256, 0, stream>>>(part, grid, hidden, grad_gamma, grad_beta,
                                                                         grad_scalar);
    GCONV_LAUNCH_CHECK();
  }
  return GCONV_OK;
}

// =============================================================================================
// BatchNorm1d over [N, d] node features with batch statistics (the backbone's
// `batch_norm_layers[layer](h)` after every GIN layer, /root/reference/src/models/inductive/
// gnn.py:108-110, trainable under the freeze rule gnn.py:236-239; also the adapter's own
// `batch_norm` tail, gconv_adapter.py:98-100) — optionally with the ReLU that follows it
// (gnn.py:127-128) folded in. torch's channels-last kernels take 72 + 10 us forward and
// 88 + 15 us backward for a [13104, 300] batch (28 % of the whole fine-tune step); the data is
// 15.7 MB. Here: one pass for the column sums (persistent waves, a lane owns 4 columns per pass and
// accumulates over the rows its wave walks; sums are taken around row 0 so that a large mean does
// not cancel), a small fixed-order reduction, one pass to apply; the same three steps backward.
// =============================================================================================
namespace gconv {

enum { BN_SUMS_STATS = 0, BN_SUMS_GRAD = 1, BN_SUMS_LOCAL = 2 };   // LOCAL: one rank's share of a sharded batch

// part[block][0:d] = sum_rows A, part[block][d:2d] = sum_rows B with
//   STATS: A = x - x[0], B = (x - x[0])^2        GRAD: A = g', B = g' * xhat   (g' = g where y > 0 if relu)
template <int NP, int MODE>
__global__ __launch_bounds__(256) void bn_colsums_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ g,
                                                         const float* __restrict__ y,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ rstd,
                                                         float* __restrict__ part, int n, int d) {
  __shared__ f32x4 red[3][64 * NP];
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int nq = d >> 2;
  f32x4 c0[NP], c1[NP], a[NP], b[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int q = lane + 64 * p;
    const bool ok = q < nq;
    if (MODE == BN_SUMS_STATS) {
      c0[p] = ok ? ldg4(x + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};   // shift: row 0
      c1[p] = f32x4{1.f, 1.f, 1.f, 1.f};
    } else {
      c0[p] = ok ? ldg4(mean + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      c1[p] = ok ? ldg4(rstd + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    a[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    b[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // a wave's rows are added in row order; kU of them are requested before the first is added (a 512-molecule
  // batch gives a wave 6 - 7 rows: that many dependent round trips otherwise)
  constexpr int kU = NP <= 2 ? 4 : 2;
  const int stride = gridDim.x * 4;
  const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int row = blockIdx.x * 4 + wave; row < n; row += kU * stride) {   // wave-uniform
    f32x4 xv[kU][NP], gv[kU][NP], yv[kU][NP];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int r = row + u * stride;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int q = lane + 64 * p;
        const bool ok = q < nq && r < n;
        const size_t off = static_cast<size_t>(ok ? r : 0) * d + 4 * (ok ? q : 0);
        xv[u][p] = ok ? ldg4(x + off) : zero;
        if (MODE != BN_SUMS_STATS) {
          gv[u][p] = ok ? ldg4(g + off) : zero;
          yv[u][p] = ok && y ? ldg4(y + off) : f32x4{1.f, 1.f, 1.f, 1.f};
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (row + u * stride >= n) break;                    // (wave-uniform)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int q = lane + 64 * p;
        if (q < nq) {
          const f32x4 xc = (xv[u][p] - c0[p]) * c1[p];
          if (MODE == BN_SUMS_STATS) {
            a[p] += xc;
            b[p] += xc * xc;
          } else {
            f32x4 gg = gv[u][p];
            for (int k = 0; k < 4; ++k) gg[k] = yv[u][p][k] > 0.f ? gg[k] : 0.f;
            a[p] += gg;
            b[p] += gg * xc;
          }
        }
      }
    }
  }
  float* prow = part + static_cast<size_t>(blockIdx.x) * (2 * d);
#pragma unroll
  for (int half = 0; half < 2; ++half) {   // waves added in wave order, A then B
    if (wave > 0) {
#pragma unroll
      for (int p = 0; p < NP; ++p) red[wave - 1][lane + 64 * p] = half ? b[p] : a[p];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int q = lane + 64 * p;
        f32x4 t = half ? b[p] : a[p];
        for (int w = 0; w < 3; ++w) t += red[w][lane + 64 * p];
        if (q < nq) stg4(prow + half * d + 4 * q, t);
      }
    }
    __syncthreads();
  }
}

// reduces the partial rows (16 columns x 16 row groups per block, fixed order) and finishes:
//   STATS: mean, rstd (biased variance), running statistics (unbiased variance, torch's rule)
//   GRAD : dgamma = sum g' xhat, dbeta = sum g'
//   LOCAL: mean and M2 = sum (x - mean)^2 of THIS rank's rows (combined across ranks by bn_combine_kernel)
constexpr int kBnFinCl = 4, kBnFinRg = 64;   // columns x row groups of one bn_finalize block (256 threads)

template <int MODE>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nparts,
                                                          const float* __restrict__ x_row0, int n, int d,
                                                          float eps, float momentum,
                                                          float* __restrict__ out0, float* __restrict__ out1,
                                                          float* __restrict__ running_mean,
                                                          float* __restrict__ running_var,
                                                          long long* __restrict__ tracked = nullptr) {
  // 4 columns x 64 row groups per block: a thread adds nparts / 64 partial rows with all its loads
  // in flight at once, then a fixed-order tree over the row groups. (Round 1: 16 columns x 16 row
  // groups — 19 blocks for d = 300, each thread walking 32 rows two at a time: 7-8 us of dependent
  // loads per launch, twice per layer and direction.)
  __shared__ float red[2][kBnFinRg][kBnFinCl + 1];
  const int cl = threadIdx.x % kBnFinCl, rg = threadIdx.x / kBnFinCl;
  const int c = blockIdx.x * kBnFinCl + cl;
  float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
  if (c < d) {
    int r = rg;
    for (; r + kBnFinRg < nparts; r += 2 * kBnFinRg) {
      a0 += part[static_cast<size_t>(r) * 2 * d + c];
      b0 += part[static_cast<size_t>(r) * 2 * d + d + c];
      a1 += part[static_cast<size_t>(r + kBnFinRg) * 2 * d + c];
      b1 += part[static_cast<size_t>(r + kBnFinRg) * 2 * d + d + c];
    }
    for (; r < nparts; r += kBnFinRg) {
      a0 += part[static_cast<size_t>(r) * 2 * d + c];
      b0 += part[static_cast<size_t>(r) * 2 * d + d + c];
    }
  }
  red[0][rg][cl] = a0 + a1;
  red[1][rg][cl] = b0 + b1;
  __syncthreads();
  for (int w = kBnFinRg / 2; w >= 1; w >>= 1) {   // fixed tree: (k, k + w) pairs, halving
    if (rg < w) {
      red[0][rg][cl] += red[0][rg + w][cl];
      red[1][rg][cl] += red[1][rg + w][cl];
    }
    __syncthreads();
  }
  if (rg != 0 || c >= d) return;
  const float sa = red[0][0][cl], sb = red[1][0][cl];
  if (MODE == BN_SUMS_LOCAL) {
    const float m1 = n > 0 ? sa / static_cast<float>(n) : 0.f;   // mean - x[0]
    out0[c] = n > 0 ? x_row0[c] + m1 : 0.f;
    out1[c] = fmaxf(sb - sa * m1, 0.f);                           // M2
    if (c == 0) {   // the row count travels with the statistics, as bits (an all-gather does no arithmetic)
      int* cnt = reinterpret_cast<int*>(out1 + d);
      cnt[0] = n; cnt[1] = 0; cnt[2] = 0; cnt[3] = 0;
    }
  } else if (MODE == BN_SUMS_STATS) {
    const float inv_n = 1.f / static_cast<float>(n);
    const float m1 = sa * inv_n;                        // mean - x[0]
    const float var = fmaxf(sb * inv_n - m1 * m1, 0.f);  // biased
    const float mu = x_row0[c] + m1;
    out0[c] = mu;
    out1[c] = rsqrtf(var + eps);
    if (tracked && c == 0) *tracked += 1;   // nn.BatchNorm1d's num_batches_tracked, without a launch of its own
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    if (running_var) {
      const float unbiased = n > 1 ? var * static_cast<float>(n) / static_cast<float>(n - 1) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
  } else {
    if (out0) out0[c] = sb;   // dgamma
    if (out1) out1[c] = sa;   // dbeta
  }
}

// Sharded batch (one process per GPU, rows split by graph): every rank's (count, mean, M2) per column,
// as gathered by the caller's all-gather, merged in RANK ORDER with Chan's pairwise update — the
// same bits on every rank. gathered[r] = [mean d][M2 d][count as int32 bits, 3 pad words].
__global__ __launch_bounds__(256) void bn_combine_kernel(const float* __restrict__ gathered, int world, int d,
                                                         float eps, float momentum,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                         float* __restrict__ running_mean,
                                                         float* __restrict__ running_var,
                                                         int* __restrict__ total_count) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int stride = 2 * d + 4;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  long long total = 0;
  for (int r = 0; r < world; ++r) {
    const float* row = gathered + static_cast<size_t>(r) * stride;
    const int cnt = __float_as_int(row[2 * d]);
    total += cnt;
    if (cnt <= 0 || c >= d) continue;
    const float nr = static_cast<float>(cnt), nn = n + nr;
    const float delta = row[c] - mean;
    mean += delta * (nr / nn);
    m2 += row[d + c] + delta * delta * (n * nr / nn);
    n = nn;
  }
  if (c == 0) *total_count = static_cast<int>(total);
  if (c >= d) return;
  const float var = n > 0.f ? m2 / n : 0.f;   // biased
  mean_out[c] = mean;
  rstd_out[c] = rsqrtf(var + eps);
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
  if (running_var) {
    const float unbiased = n > 1.f ? m2 / (n - 1.f) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// y = [relu](gamma * (x - mean) * rstd + beta)
template <int NP>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       float* __restrict__ y, int n, int d, int relu) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave_id();
  if (row >= n) return;
  const int nq = d >> 2;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int q = lane + 64 * p;
    if (q < nq) {
      const size_t off = static_cast<size_t>(row) * d + 4 * q;
      f32x4 v = (ldg4(x + off) - ldg4(mean + 4 * q)) * ldg4(rstd + 4 * q);
      if (gamma) v = v * ldg4(gamma + 4 * q);
      if (beta) v = v + ldg4(beta + 4 * q);
      if (relu)
        for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
      stg4(y + off, v);
    }
  }
}

// dx = gamma * rstd * (g' - dbeta / n - xhat * dgamma / n)
template <int NP>
__global__ __launch_bounds__(256) void bn_dx_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                    const float* __restrict__ y,
                                                    const float* __restrict__ mean,
                                                    const float* __restrict__ rstd,
                                                    const float* __restrict__ gamma,
                                                    const float* __restrict__ dgamma,
                                                    const float* __restrict__ dbeta,
                                                    float* __restrict__ dx, int n, int d,
                                                    const int* __restrict__ total_count) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave_id();
  if (row >= n) return;
  const int nq = d >> 2;
  // a sharded batch divides by the rows of ALL ranks (dgamma / dbeta are then the all-reduced sums)
  const float inv_n = 1.f / static_cast<float>(total_count ? *total_count : n);
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int q = lane + 64 * p;
    if (q < nq) {
      const size_t off = static_cast<size_t>(row) * d + 4 * q;
      const f32x4 rs = ldg4(rstd + 4 * q);
      const f32x4 xh = (ldg4(x + off) - ldg4(mean + 4 * q)) * rs;
      f32x4 gv = ldg4(g + off);
      if (y) {
        const f32x4 yv = ldg4(y + off);
        for (int k = 0; k < 4; ++k) gv[k] = yv[k] > 0.f ? gv[k] : 0.f;
      }
      f32x4 v = (gv - ldg4(dbeta + 4 * q) * inv_n - xh * (ldg4(dgamma + 4 * q) * inv_n)) * rs;
      if (gamma) v = v * ldg4(gamma + 4 * q);
      stg4(dx + off, v);
    }
  }
}

static int bn_grid(int n) { return min(ceil_div(n > 0 ? n : 1, 4), launch_cus() * 2); }

// workspace: [partials: grid x 2d] [dgamma d] [dbeta d]
struct BnWs {
  float *part, *dgamma, *dbeta;
  size_t bytes;
};
static BnWs bn_carve(void* ws, int n, int d) {
  BnWs w;
  char* base = static_cast<char*>(ws);
  size_t off = 0;
  auto take = [&](size_t count) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += align_up(count * sizeof(float) + 16, 256);
    return p;
  };
  w.part = take(static_cast<size_t>(bn_grid(n)) * 2 * d);
  w.dgamma = take(d);
  w.dbeta = take(d);
  w.bytes = off;
  return w;
}

#define GCONV_BN_DISPATCH(KERNEL_CALL)                \
  switch (ceil_div(hidden, 256)) {                    \
    case 1: { constexpr int NP = 1; KERNEL_CALL; } break; \
    case 2: { constexpr int NP = 2; KERNEL_CALL; } break; \
    case 3: { constexpr int NP = 3; KERNEL_CALL; } break; \
    default: { constexpr int NP = 4; KERNEL_CALL; } break; \
  }

}  // namespace gconv

extern "C" size_t gconv_batch_norm_workspace_bytes(int32_t num_nodes, int32_t hidden) {
  if (ln_check(num_nodes, hidden) != GCONV_OK) return 0;
  return bn_carve(nullptr, num_nodes, hidden).bytes;
}

extern "C" gconv_status_t gconv_batch_norm_fwd_f32(const float* x, const float* gamma, const float* beta,
                                                   float* running_mean, float* running_var,
                                                   float momentum, float eps, int32_t relu, float* y,
                                                   float* save_mean, float* save_rstd,
                                                   int32_t num_nodes, int32_t hidden, void* workspace,
                                                   size_t workspace_bytes, gconv_stream_t stream_) {
  return gconv_batch_norm_fwd_tracked_f32(x, gamma, beta, running_mean, running_var, nullptr, momentum, eps, relu, y,
                                          save_mean, save_rstd, num_nodes, hidden, workspace, workspace_bytes,
                                          stream_);
}

extern "C" gconv_status_t gconv_batch_norm_fwd_tracked_f32(const float* x, const float* gamma, const float* beta,
                                                           float* running_mean, float* running_var,
                                                           int64_t* num_batches_tracked, float momentum, float eps,
                                                           int32_t relu, float* y, float* save_mean,
                                                           float* save_rstd, int32_t num_nodes, int32_t hidden,
                                                           void* workspace, size_t workspace_bytes,
                                                           gconv_stream_t stream_) {
  const gconv_status_t st = ln_check(num_nodes, hidden);
  if (st != GCONV_OK) return st;
  if (num_nodes < 2) return GCONV_E_INVALID_ARG;   // batch statistics need more than one row
  if (!x || !y || !save_mean || !save_rstd || !workspace) return GCONV_E_INVALID_ARG;
  const BnWs ws = bn_carve(workspace, num_nodes, hidden);
  if (workspace_bytes < ws.bytes) return GCONV_E_WORKSPACE;
  if (!aligned16(x) || !aligned16(y) || !aligned16(save_mean) || !aligned16(save_rstd) ||
      (gamma && !aligned16(gamma)) || (beta && !aligned16(beta)) || !aligned16(workspace))
    return GCONV_E_ALIGNMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int grid = bn_grid(num_nodes);
  GCONV_BN_DISPATCH((bn_colsums_kernel<NP, BN_SUMS_STATS><<<grid, 256, 0, stream>>>(
      x, nullptr, nullptr, nullptr, nullptr, ws.part, num_nodes, hidden)))
  GCONV_LAUNCH_CHECK();
  bn_finalize_kernel<BN_SUMS_STATS><<<ceil_div(hidden, kBnFinCl), 256, 0, stream>>>(
      ws.part, grid, x, num_nodes, hidden, eps, momentum, save_mean, save_rstd, running_mean, running_var,
      reinterpret_cast<long long*>(num_batches_tracked));
  GCONV_LAUNCH_CHECK();
  GCONV_BN_DISPATCH((bn_apply_kernel<NP><<<ceil_div(num_nodes, 4), 256, 0, stream>>>(
      x, save_mean, save_rstd, gamma, beta, y, num_nodes, hidden, relu)))
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

extern "C" gconv_status_t gconv_batch_norm_apply_f32(const float* x, const float* mean, const float* rstd,
                                                     const float* gamma, const float* beta, int32_t relu,
                                                     float* y, int32_t num_nodes, int32_t hidden,
                                                     gconv_stream_t stream_) {
  const gconv_status_t st = ln_check(num_nodes, hidden);
  if (st != GCONV_OK) return st;
  if (num_nodes == 0) return GCONV_OK;
  if (!x || !y || !mean || !rstd) return GCONV_E_INVALID_ARG;
  if (!aligned16(x) || !aligned16(y) || !aligned16(mean) || !aligned16(rstd) ||
      (gamma && !aligned16(gamma)) || (beta && !aligned16(beta)))
    return GCONV_E_ALIGNMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GCONV_BN_DISPATCH((bn_apply_kernel<NP><<<ceil_div(num_nodes, 4), 256, 0, stream>>>(
      x, mean, rstd, gamma, beta, y, num_nodes, hidden, relu)))
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

extern "C" gconv_status_t gconv_batch_norm_bwd_f32(const float* grad_y, const float* x, const float* y,
                                                   const float* save_mean, const float* save_rstd,
                                                   const float* gamma, float* grad_x, float* grad_gamma,
                                                   float* grad_beta, int32_t num_nodes, int32_t hidden,
                                                   void* workspace, size_t workspace_bytes,
                                                   gconv_stream_t stream_) {
  const gconv_status_t st = ln_check(num_nodes, hidden);
  if (st != GCONV_OK) return st;
  if (num_nodes < 2) return GCONV_E_INVALID_ARG;
  if (!grad_y || !x || !save_mean || !save_rstd || !workspace) return GCONV_E_INVALID_ARG;
  const BnWs ws = bn_carve(workspace, num_nodes, hidden);
  if (workspace_bytes < ws.bytes) return GCONV_E_WORKSPACE;
  if (!aligned16(grad_y) || !aligned16(x) || (y && !aligned16(y)) || !aligned16(save_mean) ||
      !aligned16(save_rstd) || (gamma && !aligned16(gamma)) || (grad_x && !aligned16(grad_x)) ||
      (grad_gamma && !aligned16(grad_gamma)) || (grad_beta && !aligned16(grad_beta)) || !aligned16(workspace))
    return GCONV_E_ALIGNMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int grid = bn_grid(num_nodes);
  GCONV_BN_DISPATCH((bn_colsums_kernel<NP, BN_SUMS_GRAD><<<grid, 256, 0, stream>>>(
      x, grad_y, y, save_mean, save_rstd, ws.part, num_nodes, hidden)))
  GCONV_LAUNCH_CHECK();
  // the column sums are needed by dx whether or not the caller wants them back
  float* dg = grad_gamma ? grad_gamma : ws.dgamma;
  float* db = grad_beta ? grad_beta : ws.dbeta;
  bn_finalize_kernel<BN_SUMS_GRAD><<<ceil_div(hidden, kBnFinCl), 256, 0, stream>>>(
      ws.part, grid, nullptr, num_nodes, hidden, 0.f, 0.f, dg, db, nullptr, nullptr);
  GCONV_LAUNCH_CHECK();
  if (grad_x) {
    GCONV_BN_DISPATCH((bn_dx_kernel<NP><<<ceil_div(num_nodes, 4), 256, 0, stream>>>(
        grad_y, x, y, save_mean, save_rstd, gamma, dg, db, grad_x, num_nodes, hidden, nullptr)))
    GCONV_LAUNCH_CHECK();
  }
  return GCONV_OK;
}

// ---- the same BatchNorm over a batch whose rows are sharded across ranks ("optional
// all_reduce of BN stats"; the reference trains the backbone's BatchNorm on WHOLE-batch statistics,
// gnn.py:56,109-110). The C ABI never calls a collective: forward and backward are cut where the
// exchange happens, the caller (norm.py) runs one small collective in between.
extern "C" size_t gconv_batch_norm_sync_words(int32_t hidden) { return 2 * static_cast<size_t>(hidden) + 4; }

extern "C" gconv_status_t gconv_batch_norm_local_stats_f32(const float* x, int32_t num_nodes, int32_t hidden,
                                                           float* local, void* workspace,
                                                           size_t workspace_bytes, gconv_stream_t stream_) {
  const gconv_status_t st = ln_check(num_nodes, hidden);
  if (st != GCONV_OK) return st;
  if (!local || !workspace || (num_nodes > 0 && !x)) return GCONV_E_INVALID_ARG;
  const BnWs ws = bn_carve(workspace, num_nodes, hidden);
  if (workspace_bytes < ws.bytes) return GCONV_E_WORKSPACE;
  if ((x && !aligned16(x)) || !aligned16(local) || !aligned16(workspace)) return GCONV_E_ALIGNMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int grid = bn_grid(num_nodes);
  if (num_nodes > 0) {
    GCONV_BN_DISPATCH((bn_colsums_kernel<NP, BN_SUMS_STATS><<<grid, 256, 0, stream>>>(
        x, nullptr, nullptr, nullptr, nullptr, ws.part, num_nodes, hidden)))
    GCONV_LAUNCH_CHECK();
  }
  bn_finalize_kernel<BN_SUMS_LOCAL><<<ceil_div(hidden, kBnFinCl), 256, 0, stream>>>(
      ws.part, num_nodes > 0 ? grid : 0, x, num_nodes, hidden, 0.f, 0.f, local, local + hidden, nullptr, nullptr);
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

extern "C" gconv_status_t gconv_batch_norm_fwd_sync_f32(const float* x, const float* gamma, const float* beta,
                                                        float* running_mean, float* running_var, float momentum,
                                                        float eps, int32_t relu, float* y, float* save_mean,
                                                        float* save_rstd, int32_t* total_count,
                                                        int32_t num_nodes, int32_t hidden, const float* gathered,
                                                        int32_t world, gconv_stream_t stream_) {
  const gconv_status_t st = ln_check(num_nodes, hidden);
  if (st != GCONV_OK) return st;
  if (!save_mean || !save_rstd || !total_count || !gathered || world < 1 || (num_nodes > 0 && (!x || !y)))
    return GCONV_E_INVALID_ARG;
  if ((x && !aligned16(x)) || (y && !aligned16(y)) || !aligned16(save_mean) || !aligned16(save_rstd) ||
      (gamma && !aligned16(gamma)) || (beta && !aligned16(beta)))
    return GCONV_E_ALIGNMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  bn_combine_kernel<<<ceil_div(hidden, 256), 256, 0, stream>>>(gathered, world, hidden, eps, momentum, save_mean,
                                                              save_rstd, running_mean, running_var, total_count);
  GCONV_LAUNCH_CHECK();
  if (num_nodes > 0) {
    GCONV_BN_DISPATCH((bn_apply_kernel<NP><<<ceil_div(num_nodes, 4), 256, 0, stream>>>(
        x, save_mean, save_rstd, gamma, beta, y, num_nodes, hidden, relu)))
    GCONV_LAUNCH_CHECK();
  }
  return GCONV_OK;
}

extern "C" gconv_status_t gconv_batch_norm_bwd_local_f32(const float* grad_y, const float* x, const float* y,
                                                         const float* save_mean, const float* save_rstd,
                                                         float* sums, int32_t num_nodes, int32_t hidden,
                                                         void* workspace, size_t workspace_bytes,
                                                         gconv_stream_t stream_) {
  const gconv_status_t st = ln_check(num_nodes, hidden);
  if (st != GCONV_OK) return st;
  if (!save_mean || !save_rstd || !sums || !workspace || (num_nodes > 0 && (!grad_y || !x)))
    return GCONV_E_INVALID_ARG;
  const BnWs ws = bn_carve(workspace, num_nodes, hidden);
  if (workspace_bytes < ws.bytes) return GCONV_E_WORKSPACE;
  if ((grad_y && !aligned16(grad_y)) || (x && !aligned16(x)) || (y && !aligned16(y)) || !aligned16(save_mean) ||
      !aligned16(save_rstd) || !aligned16(sums) || !aligned16(workspace))
    return GCONV_E_ALIGNMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int grid = bn_grid(num_nodes);
  if (num_nodes > 0) {
    GCONV_BN_DISPATCH((bn_colsums_kernel<NP, BN_SUMS_GRAD><<<grid, 256, 0, stream>>>(
        x, grad_y, y, save_mean, save_rstd, ws.part, num_nodes, hidden)))
    GCONV_LAUNCH_CHECK();
  }
  // sums = [dgamma d][dbeta d]: this rank's share (an empty shard contributes zeros)
  bn_finalize_kernel<BN_SUMS_GRAD><<<ceil_div(hidden, kBnFinCl), 256, 0, stream>>>(
      ws.part, num_nodes > 0 ? grid : 0, nullptr, num_nodes, hidden, 0.f, 0.f, sums, sums + hidden, nullptr, nullptr);
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}

extern "C" gconv_status_t gconv_batch_norm_bwd_dx_f32(const float* grad_y, const float* x, const float* y,
                                                      const float* save_mean, const float* save_rstd,
                                                      const float* gamma, const float* sums_all,
                                                      const int32_t* total_count, float* grad_x,
                                                      int32_t num_nodes, int32_t hidden, gconv_stream_t stream_) {
  const gconv_status_t st = ln_check(num_nodes, hidden);
  if (st != GCONV_OK) return st;
  if (num_nodes == 0) return GCONV_OK;
  if (!grad_y || !x || !save_mean || !save_rstd || !sums_all || !total_count || !grad_x) return GCONV_E_INVALID_ARG;
  if (!aligned16(grad_y) || !aligned16(x) || (y && !aligned16(y)) || !aligned16(save_mean) || !aligned16(save_rstd) ||
      (gamma && !aligned16(gamma)) || !aligned16(sums_all) || !aligned16(grad_x))
    return GCONV_E_ALIGNMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GCONV_BN_DISPATCH((bn_dx_kernel<NP><<<ceil_div(num_nodes, 4), 256, 0, stream>>>(
      grad_y, x, y, save_mean, save_rstd, gamma, sums_all, sums_all + hidden, grad_x, num_nodes, hidden, total_count)))
  GCONV_LAUNCH_CHECK();
  return GCONV_OK;
}
