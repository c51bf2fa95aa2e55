/* The C ABI of include/gconv_adapter.h used from plain C — no Python, no torch, no C++:
 * SURVEY.md §8(b)(ii) asks for "a plain C ABI usable without Python for the micro-bench".
 *
 *   gcc -std=c11 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/cabi_spmm_bench.c \
 *       -Lgconv-adapter_amd/lib -lgconv_adapter -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/gconv-adapter_amd/lib -Wl,-rpath,/opt/rocm/lib -o tools/cabi_spmm_bench
 *   tools/cabi_spmm_bench [nodes=170000] [pairs=600000] [width=32]
 *
 * Builds a power-law graph of S-arxiv's shape on the host (its own LCG: the shape, not the bits, of
 * gconv_adapter_amd.synth.arxiv_like), hands the COO list to gconv_graph_build, runs the normalized-
 * adjacency SpMM (the roofline kernel: what PyG's MessagePassing.propagate does at
 * /root/reference/src/finetune/gconv_adapter.py:92) through gconv_spmm_csr_f32 with HIP events around it,
 * checks a sample of output rows against a host loop over the edge list, and prints the launch time and the
 * GB/s at algorithmic bytes. Exit code 0 only if the check passes. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "gconv_adapter.h"

#define HIP(expr)                                                                    \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      return 2;                                                                      \
    }                                                                                \
  } while (0)
#define GCONV(expr)                                                                  \
  do {                                                                               \
    gconv_status_t s_ = (expr);                                                      \
    if (s_ != GCONV_OK) {                                                            \
      fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #expr, s_, gconv_status_string(s_)); \
      return 3;                                                                      \
    }                                                                                \
  } while (0)

static uint64_t lcg_state = 0x9E3779B97F4A7C15ull;
static double urand(void) {   /* [0, 1) */
  lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(lcg_state >> 11) / 9007199254740992.0;
}

int main(int argc, char** argv) {
  const int32_t n = argc > 1 ? atoi(argv[1]) : 170000;
  const int64_t pairs = argc > 2 ? atoll(argv[2]) : 600000;
  const int32_t width = argc > 3 ? atoi(argv[3]) : 32;
  if (gconv_abi_version() != GCONV_ABI_VERSION) {
    fprintf(stderr, "library ABI %d, header ABI %d\n", gconv_abi_version(), GCONV_ABI_VERSION);
    return 1;
  }
  if (gconv_abi_struct_bytes(0) != (int32_t)sizeof(gconv_csr_t) || gconv_abi_struct_bytes(1) != (int32_t)sizeof(gconv_graph_t)) {
    fprintf(stderr, "struct layout mismatch between this C compiler and the library\n");
    return 1;
  }

  /* u uniform, v = floor(n u^3): hubs at the low ids; both directions, no self-loops (SURVEY §8d, S-arxiv) */
  int64_t* ei = (int64_t*)malloc(sizeof(int64_t) * 4 * (size_t)pairs);
  int64_t e = 0;
  for (int64_t p = 0; p < pairs; ++p) {
    const int64_t u = (int64_t)(urand() * n);
    const double r = urand();
    const int64_t v = (int64_t)(r * r * r * n);
    if (u == v) continue;
    ei[2 * e] = u; ei[2 * e + 1] = v; ++e;   /* (pairs first, split into the two rows below) */
  }
  const int64_t E = 2 * e;
  int64_t* coo = (int64_t*)malloc(sizeof(int64_t) * 2 * (size_t)E);   /* [2, E]: row 0 sources, row 1 targets */
  for (int64_t k = 0; k < e; ++k) {
    coo[k] = ei[2 * k];         coo[E + k] = ei[2 * k + 1];
    coo[e + k] = ei[2 * k + 1]; coo[E + e + k] = ei[2 * k];
  }
  free(ei);

  hipStream_t stream;
  HIP(hipStreamCreate(&stream));
  int64_t* d_coo;
  HIP(hipMalloc((void**)&d_coo, sizeof(int64_t) * 2 * (size_t)E));
  HIP(hipMemcpy(d_coo, coo, sizeof(int64_t) * 2 * (size_t)E, hipMemcpyHostToDevice));
  const size_t arena_bytes = gconv_graph_build_bytes(E, n);
  void* arena;
  HIP(hipMalloc(&arena, arena_bytes));
  gconv_graph_t g;
  GCONV(gconv_graph_build(d_coo, E, n, /*normalize=*/1, arena, arena_bytes, &g, stream));
  HIP(hipStreamSynchronize(stream));
  printf("graph: %d nodes, %lld directed edges -> nnz %d, %d long rows in %d segments, symmetric %d, max degree %d\n", n,
         (long long)E, g.fwd.nnz, g.fwd.n_long, g.fwd.n_seg, g.symmetric, g.max_degree);

  const size_t hb = sizeof(float) * (size_t)n * width;
  float* h = (float*)malloc(hb);
  for (size_t i = 0; i < (size_t)n * width; ++i) h[i] = (float)(urand() * 2.0 - 1.0);
  float *d_h, *d_y;
  HIP(hipMalloc((void**)&d_h, hb));
  HIP(hipMalloc((void**)&d_y, hb));
  HIP(hipMemcpy(d_h, h, hb, hipMemcpyHostToDevice));
  const size_t ws_bytes = gconv_spmm_workspace_bytes(&g.fwd, width);
  void* ws;
  HIP(hipMalloc(&ws, ws_bytes > 16 ? ws_bytes : 16));

  /* Y = D^-1/2 (A + I) D^-1/2 H: plain call (no pre-scaled operands, no bias, no activation) */
  for (int it = 0; it < 5; ++it)
    GCONV(gconv_spmm_csr_f32(&g.fwd, g.dis, d_h, d_y, width, NULL, GCONV_ACT_NONE, 0u, ws, ws_bytes, stream));
  hipEvent_t t0, t1;
  HIP(hipEventCreate(&t0));
  HIP(hipEventCreate(&t1));
  const int iters = 50;
  HIP(hipEventRecord(t0, stream));
  for (int it = 0; it < iters; ++it)
    GCONV(gconv_spmm_csr_f32(&g.fwd, g.dis, d_h, d_y, width, NULL, GCONV_ACT_NONE, 0u, ws, ws_bytes, stream));
  HIP(hipEventRecord(t1, stream));
  HIP(hipEventSynchronize(t1));
  float ms = 0.f;
  HIP(hipEventElapsedTime(&ms, t0, t1));
  const double us = ms * 1e3 / iters;
  const double bytes = 4.0 * (n + 1) + 4.0 * g.fwd.nnz + 4.0 * n + 8.0 * (double)n * width;   /* SURVEY §8d */

  /* check: in-degree on the target incl. the unit loop, w = dis[j] dis[i]; 256 sampled rows on the host */
  float* y = (float*)malloc(hb);
  HIP(hipMemcpy(y, d_y, hb, hipMemcpyDeviceToHost));
  double* deg = (double*)calloc((size_t)n, sizeof(double));
  for (int64_t k = 0; k < E; ++k) deg[coo[E + k]] += 1.0;
  double worst = 0.0;
  for (int s = 0; s < 256; ++s) {
    const int64_t i = s < 8 ? s : (int64_t)(urand() * n);    /* the first rows are the hubs (long rows) */
    for (int c = 0; c < width; c += width - 1 > 0 ? width - 1 : 1) {
      double acc = h[(size_t)i * width + c] / (deg[i] + 1.0);                       /* the loop */
      for (int64_t k = 0; k < E; ++k)
        if (coo[E + k] == i) acc += h[(size_t)coo[k] * width + c] / sqrt((deg[i] + 1.0) * (deg[coo[k]] + 1.0));
      const double err = fabs(acc - y[(size_t)i * width + c]) / fmax(1.0, fabs(acc));
      if (err > worst) worst = err;
    }
  }
  printf("gconv_spmm_csr_f32, plain call: %.2f us per launch, %.1f GB/s at algorithmic bytes (%.1f MB), %.3f of the 8 TB/s "
         "spec; max relative error of 256 sampled rows vs a host loop %.2e\n",
         us, bytes / (us * 1e-6) / 1e9, bytes / 1e6, bytes / (us * 1e-6) / 1e9 / 8000.0, worst);

  /* the form the adapter chains (bench.py's roofline kernel): operand rows already carry dis[j], output rows are
   * scaled once more for the next hop — no second gather per edge:  Y[i] = dis[i]^2 (H'[i] + sum_j H'[j]) */
  const uint32_t flags = GCONV_SPMM_IN_PRESCALED | GCONV_SPMM_OUT_PRESCALE;
  for (int it = 0; it < 5; ++it)
    GCONV(gconv_spmm_csr_f32(&g.fwd, g.dis, d_h, d_y, width, NULL, GCONV_ACT_NONE, flags, ws, ws_bytes, stream));
  HIP(hipEventRecord(t0, stream));
  for (int it = 0; it < iters; ++it)
    GCONV(gconv_spmm_csr_f32(&g.fwd, g.dis, d_h, d_y, width, NULL, GCONV_ACT_NONE, flags, ws, ws_bytes, stream));
  HIP(hipEventRecord(t1, stream));
  HIP(hipEventSynchronize(t1));
  HIP(hipEventElapsedTime(&ms, t0, t1));
  const double us2 = ms * 1e3 / iters;
  HIP(hipMemcpy(y, d_y, hb, hipMemcpyDeviceToHost));
  double worst2 = 0.0;
  for (int s = 0; s < 64; ++s) {
    const int64_t i = s < 8 ? s : (int64_t)(urand() * n);
    double acc = h[(size_t)i * width];
    for (int64_t k = 0; k < E; ++k)
      if (coo[E + k] == i) acc += h[(size_t)coo[k] * width];
    acc /= deg[i] + 1.0;
    const double err = fabs(acc - y[(size_t)i * width]) / fmax(1.0, fabs(acc));
    if (err > worst2) worst2 = err;
  }
  printf("gconv_spmm_csr_f32, pre-scaled operands (the adapter's chain): %.2f us per launch, %.1f GB/s, %.3f of the spec; "
         "max relative error of 64 sampled rows %.2e\n",
         us2, bytes / (us2 * 1e-6) / 1e9, bytes / (us2 * 1e-6) / 1e9 / 8000.0, worst2);
  return worst < 1e-5 && worst2 < 1e-5 ? 0 : 4;
}
