/*
 * gconv_adapter.h — C ABI of the MI355X (gfx950) GConv-Adapter hot path.
 *
 * Drop-in boundary for the forward/backward of the reference's GConvAdapter
 * (`/root/reference/src/finetune/gconv_adapter.py`). The reference is pure
 * Python and has no FFI of its own; the arithmetic it calls lives in
 * torch-geometric's `GCNConv` (construction `gconv_adapter.py:40-41`, call
 * `gconv_adapter.py:92`). Every entry point below names the reference
 * interface it replaces. `INTEGRATION.md` shows the ctypes binding a
 * maintainer of the reference adds.
 *
 * Conventions
 *   - plain C: pointers, sizes, POD structs; no torch / C++ types.
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`.
 *   - the caller owns every buffer; nothing is allocated or freed here, and no
 *     pointer is retained after a call returns.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*). Only
 *     gconv_graph_build() synchronises that stream (it
 *     reads a few counters back to the host); everything else is asynchronous
 *     and hipGraph-capturable.
 *   - return value: 0 = success, > 0 = a hipError_t, < 0 = GCONV_E_*.
 *     No exceptions cross this boundary and no entry point aborts.
 *   - fp32 data, int32 CSR indices, int64 COO `edge_index` (PyG layout:
 *     row 0 = source j, row 1 = target i; messages flow j -> i).
 *
 * Rank-width buffers ("[N, Fp]"): the bottleneck rank r is padded to
 * Fp = gconv_padded_rank(r) in {4, 8, 16, 32, 64}; rows are Fp floats,
 * 16-byte aligned, columns >= r hold zeros.
 */
#ifndef GCONV_ADAPTER_H
#define GCONV_ADAPTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCONV_ABI_VERSION 11

typedef int32_t gconv_status_t;
typedef void*   gconv_stream_t; /* hipStream_t */

enum {
  GCONV_OK              = 0,
  GCONV_E_INVALID_ARG   = -1, /* null pointer, negative size, bad enum            */
  GCONV_E_UNSUPPORTED   = -2, /* shape outside the compiled kernel set            */
  GCONV_E_WORKSPACE     = -3, /* workspace_bytes smaller than *_workspace_bytes() */
  GCONV_E_BAD_INDEX     = -4, /* edge_index entry outside [0, num_nodes)          */
  GCONV_E_ALIGNMENT     = -5  /* a float buffer is not 16-byte aligned            */
};

/* non_linearity of the adapter (`gconv_adapter.py:44-51`) */
enum { GCONV_ACT_NONE = 0, GCONV_ACT_RELU = 1, GCONV_ACT_SILU = 2 };

/* rows with more than GCONV_LONG_ROW edges are split into segments of
 * GCONV_SEG_EDGES edges that separate waves reduce (deterministic, no atomics) */
#ifndef GCONV_LONG_ROW      /* (overridable at build time for kernel-variant experiments only) */
#define GCONV_LONG_ROW  64
#endif
#ifndef GCONV_SEG_EDGES
#define GCONV_SEG_EDGES 128
#endif

/* ------------------------------------------------------------------------- */
/* Introspection                                                             */
/* ------------------------------------------------------------------------- */
int32_t     gconv_abi_version(void);
/* sizeof of the ABI's POD structs as compiled into the library, for bindings to check their
 * mirrors: 0 gconv_csr_t, 1 gconv_graph_t, 2 gconv_adapter_params_t, 3 gconv_pending_finalize_t,
 * 4 gconv_tile_t, 5 gconv_tiling_t */
int32_t     gconv_abi_struct_bytes(int32_t which);
const char* gconv_status_string(gconv_status_t status);
/* Fp for a bottleneck rank r (0 if r is unsupported: r < 1 or r > 64). */
int32_t     gconv_padded_rank(int32_t rank);

/* ------------------------------------------------------------------------- */
/* K0 graph preparation                                                      */
/* replaces: PyG `gcn_norm` (add_remaining_self_loops + degree + deg^-1/2),  */
/* which GCNConv re-runs on every conv call (2x per adapter per forward) for */
/* the same `edge_index` — reached from `gconv_adapter.py:92`; and the       */
/* implicit COO gather/scatter indexing of `MessagePassing.propagate`.       */
/* ------------------------------------------------------------------------- */

/* CSR view of the propagation operator (or of its transpose). Row i lists the
 * source nodes j of the edges j -> i (duplicates kept, sorted by j). With
 * `normalize`, pre-existing self-loops are removed and the unit self-loop of
 * every node is IMPLICIT (not stored); the kernels add it. */
typedef struct gconv_csr {
  int32_t        num_nodes;
  int32_t        nnz;           /* stored edges (after self-loop removal)     */
  const int32_t* rowptr;        /* [num_nodes + 1]                            */
  const int32_t* col;           /* [nnz]                                      */
  int32_t        n_long;        /* rows with more than GCONV_LONG_ROW edges   */
  int32_t        n_seg;         /* total GCONV_SEG_EDGES-edge segments        */
  const int32_t* long_rows;     /* [n_long] row ids                           */
  const int32_t* long_seg_base; /* [n_long] first segment of each long row    */
  const int32_t* seg_long;      /* [n_seg]  segment -> index into long_rows   */
  uint32_t*      long_sync;     /* [n_long] arrival counters of a long row's segment waves:
                                 * zero between launches (the wave that completes a row adds
                                 * the row's segment sums and resets its counter). Launches
                                 * that share a CSR must therefore be stream-ordered.      */
  int32_t        near_edges;    /* stored edges whose column lies in the same EIGHTH of the node range as
                                 * their row (counted by the radix builder; 0 = not counted: the one-launch
                                 * builders of small graphs / molecule batches). At least half of nnz: the
                                 * graph's neighbourhoods are local, and the row kernel then gives every XCD
                                 * a CONTIGUOUS eighth of the rows instead of every eighth block (round 5:
                                 * with round-robin blocks each of the eight private L2s fetches every H row —
                                 * 7 x the table on the banded control, profiles/r05_spmm_banded_pmc.json).  */
} gconv_csr_t;

/* Tiles of the fused (one-launch) adapter kernels. The node range is cut into contiguous tiles
 * of about gconv_tile_nodes(N) nodes — N / 256, so that each of the 256 persistent workgroups
 * gets one tile of about the same size — that ONE workgroup each runs end to end (rank-width
 * intermediates never leave LDS). A cut sits on a boundary that no edge crosses when there is
 * one within a quarter tile below its nominal position (collated molecule batches: between two
 * molecules; with several tiles per workgroup — more than 256 x GCONV_TILE_NODES_MAX nodes — the
 * workgroups balance themselves by dealing tiles largest first, and a cut may sit up to a whole
 * tile, at most 64 nodes, below); otherwise it goes through a component and the tile carries a HALO: the nodes
 * outside its range that its rows gather from (halo1) and the nodes those gather from (halo2),
 * whose rank-width rows the tile recomputes for itself — tiles exchange nothing.
 * (Round 2 cut at edge-free boundaries only; a batch's largest molecule then was one tile.) */
#ifndef GCONV_TILE_NODES
#define GCONV_TILE_NODES 16       /* smallest nominal tile (one 16-row MFMA block)             */
#endif
#ifndef GCONV_TILE_NODES_MAX
#define GCONV_TILE_NODES_MAX 128  /* largest: beyond 256 x this many nodes a workgroup walks several tiles */
#endif
#define GCONV_TILE_HALO_NODES 64  /* halo1 + halo2 nodes a tile may carry                      */
#define GCONV_TILE_HALO_EDGES 256 /* stored edges of its halo1 rows                            */
#define GCONV_TILE_HALO_SLOT  456 /* 4-byte words per tile in `halo`: [0,64) halo node ids (halo1
                                   * first), [64,129) local rowptr of the halo1 rows, [132,388)
                                   * their local column ids, [388,452) dis of the halo nodes
                                   * (float; only when normalised)                              */
/* nominal nodes per tile of an N-node graph (env GCONV_TILE_NODES_OVERRIDE=k forces k) */
int32_t gconv_tile_nodes(int32_t num_nodes);

typedef struct gconv_tile {       /* 32 bytes                                                  */
  int32_t first;                  /* first own node                                            */
  int32_t count;                  /* own nodes                                                 */
  int32_t n1;                     /* count + |halo1|: rows of the first propagation            */
  int32_t n0;                     /* n1 + |halo2|: rows of the down projection                 */
  int32_t slot;                   /* index of the tile's slot in `halo` (= its position in tile_ptr) */
  int32_t e0;                     /* rowptr[first]                                             */
  int32_t ne_own;                 /* stored edges of the own rows                              */
  int32_t ne_halo;                /* stored edges of the halo1 rows                            */
} gconv_tile_t;

/* tile data of ONE direction (S or S^T; the cuts are shared) */
typedef struct gconv_tiling {
  const gconv_tile_t* tiles;      /* [n_tiles] largest (n0) first, ties in node order (stable):
                                   * the persistent workgroups deal themselves tiles from this
                                   * list boustrophedon, so no workgroup gets two big ones. With
                                   * at most 256 tiles (one per workgroup): node order, slot == index */
  const int32_t*      lcol;       /* [nnz] the CSR's `col` in tile-local row numbers: node - first
                                   * for an own node, count + k for the tile's k-th halo node   */
  const int32_t*      halo;       /* [n_tiles][GCONV_TILE_HALO_SLOT]                            */
} gconv_tiling_t;

/* Everything the kernels need to apply S and S^T for one edge list. All pointers
 * point into the caller's arena (see gconv_graph_build). */
typedef struct gconv_graph {
  gconv_csr_t    fwd;            /* rows = targets: the forward operator S              */
  gconv_csr_t    bwd;            /* rows = sources: S^T (a copy of `fwd` if symmetric)  */
  const float*   dis;            /* [num_nodes] (in-degree + 1)^-1/2; NULL if !normalize */
  const int32_t* tile_ptr;       /* [n_tiles + 1] first node of every tile, ascending   */
  gconv_tiling_t tiles_fwd;      /* tiles of `fwd`                                      */
  gconv_tiling_t tiles_bwd;      /* tiles of `bwd` (a copy of tiles_fwd if symmetric)   */
  int32_t        n_tiles;        /* 0: not tileable (a tile's halo outgrew its slot)    */
  int32_t        tile_nodes_max; /* largest n0 (rows incl. halo) of a tile, either direction */
  int32_t        tile_edges_max; /* most stored edges (own + halo1 rows) of a tile       */
  int32_t        symmetric;      /* edge multiset equals its transpose                  */
  int32_t        normalize;
  int32_t        max_degree;     /* of `fwd`                                            */
  int32_t        bad_index;      /* 1 if an edge endpoint was outside [0, num_nodes)    */
} gconv_graph_t;

/* Bytes of device memory gconv_graph_build needs (outputs + scratch, upper bound). */
size_t gconv_graph_build_bytes(int64_t num_edges, int32_t num_nodes);

/* Builds the whole graph plan from a COO edge list in ONE call. Three builders, same outputs
 * (bit-identical arrays and counters; the tests compare them):
 *   <= 4096 nodes and edges      ONE single-workgroup launch (sort, CSR, symmetry, tiles in LDS)
 *   4097 .. 262144 nodes,        ONE launch of 256 resident workgroups, no sort: every edge takes a
 *   <= 2^20 edges, rows <= 16    slot of its row by atomic counting, rows are ranked in registers,
 *   edges, symmetric             two grid barriers (graph_prep.hip: counting_build_kernel). Uses a
 *                                ~1 MB block of device memory the library allocates once per device
 *                                and serialises concurrent builds of this kind on a mutex.
 *   anything else                key build -> radix sort -> CSR (+ deg^-1/2, long-row segments) ->
 *                                exact symmetry test -> S^T only if not symmetric -> tiles
 * `arena` (device, >= gconv_graph_build_bytes, 256-B aligned) receives every output
 * array; it must stay alive as long as `graph_host_out` is used.
 * The call waits for the counters (nnz, symmetry, maxima) to reach the host — a stream
 * synchronisation, or for the one-launch builders a completion mark in pinned memory — so it is
 * not hipGraph-capturable. The results are ORDERED ON `stream`, not globally complete: the last
 * launch (the size sort that fills `tiles_fwd.tiles`, or the build kernel's final instructions)
 * may still be pending when the call returns. Consumers on `stream` are ordered behind it by the
 * stream itself, a consumer on ANOTHER stream must first wait for an event recorded on `stream`
 * after this call.
 * Returns GCONV_E_BAD_INDEX if an endpoint is out of range — never a device fault. */
gconv_status_t gconv_graph_build(const int64_t* edge_index, int64_t num_edges,
                                 int32_t num_nodes, int32_t normalize,
                                 void* arena, size_t arena_bytes,
                                 gconv_graph_t* graph_host_out, gconv_stream_t stream);

/* The same build with hints. GCONV_BUILD_BACKGROUND: the caller runs this build NEXT TO other
 * kernels (a plan prefetched on a side stream while the previous step computes). The one-launch
 * builder of mid-size graphs is 256 workgroups that must all be resident and wait for one another
 * at grid barriers: beside the fused tile kernels, which fill every CU's register file, its
 * workgroups are placed a few at a time, hold their CUs while they wait for the rest, and every
 * tile kernel that meets them runs its tiles in two rounds (measured: HIV-512 step 0.2025 ->
 * 0.2474 ms). With the flag the build uses the multi-launch pipeline, whose short, narrow kernels
 * slip in between. Same outputs either way. */
enum { GCONV_BUILD_BACKGROUND = 1u };
gconv_status_t gconv_graph_build_ex(const int64_t* edge_index, int64_t num_edges,
                                    int32_t num_nodes, int32_t normalize, uint32_t flags,
                                    void* arena, size_t arena_bytes,
                                    gconv_graph_t* graph_host_out, gconv_stream_t stream);

/* ------------------------------------------------------------------------- */
/* K2 SpMM                                                                   */
/* replaces: `MessagePassing.propagate` (index_select -> weight mul ->       */
/* scatter_add) + `out + bias` of GCNConv.forward, reached from              */
/* `gconv_adapter.py:92`.                                                    */
/* ------------------------------------------------------------------------- */
enum {
  GCONV_SPMM_IN_PRESCALED = 1u, /* H rows are already multiplied by dis[row]  */
  GCONV_SPMM_OUT_PRESCALE = 2u  /* multiply output rows by dis[row] (chaining) */
};

/* scratch floats needed by gconv_spmm_csr_f32 for long-row partial sums */
size_t gconv_spmm_workspace_bytes(const gconv_csr_t* graph, int32_t width);

/* Y = act(S * H + bias), S = A (dis == NULL) or D^-1/2 (A0 + I) D^-1/2.
 * H, Y: [num_nodes, width] contiguous, width in {4, 8, 16, 32, 64}.
 * bias: [width] or NULL. */
gconv_status_t gconv_spmm_csr_f32(const gconv_csr_t* graph, const float* dis,
                                  const float* H, float* Y, int32_t width,
                                  const float* bias, int32_t act, uint32_t flags,
                                  void* workspace, size_t workspace_bytes,
                                  gconv_stream_t stream);

/* ------------------------------------------------------------------------- */
/* K1 / K3 skinny projections                                                */
/* replaces: `GCNConv.lin` (bias-free nn.Linear) of conv_down / conv_up      */
/* (`gconv_adapter.py:40-41`) and the residual / scalar tail                 */
/* (`gconv_adapter.py:94-95,105-106`).                                       */
/* ------------------------------------------------------------------------- */

/* H0[N, Fp] = row_scale .* (X[N, d] * W_down[r, d]^T); row_scale nullable. */
gconv_status_t gconv_gemm_down_f32(const float* x, const float* w_down, const float* row_scale,
                                   float* h0, int32_t num_nodes, int32_t hidden, int32_t rank,
                                   gconv_stream_t stream);

/* out[N, d] = s * (P[N, Fp] * W_up[d, r]^T + b_up + (skip ? X : 0));
 * b_up, scalar (device, 1 float) nullable. */
gconv_status_t gconv_gemm_up_fused_f32(const float* p, const float* w_up, const float* b_up,
                                       const float* scalar, const float* x, int32_t skip,
                                       float* out, int32_t num_nodes, int32_t hidden,
                                       int32_t rank, gconv_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Whole adapter                                                             */
/* ------------------------------------------------------------------------- */
typedef struct gconv_adapter_params {
  const float* w_down;  /* conv_down.lin.weight [rank, hidden]   16-byte aligned */
  const float* b_down;  /* conv_down.bias       [rank]         (nullable)     */
  const float* w_up;    /* conv_up.lin.weight   [hidden, rank]                */
  const float* b_up;    /* conv_up.bias         [hidden]  (nullable) 16-byte aligned:
                         * forward AND backward read it 4 floats per lane; a view at an odd
                         * offset of a foreign flat buffer returns GCONV_E_ALIGNMENT — torch
                         * allocations, state_dict tensors and FusedAdam's / FlatGradBucket's
                         * flat buffers (every parameter padded to 16 bytes) satisfy it       */
  const float* scalar;  /* scalar [1], NULL when learnable_scalar = False     */
  int32_t hidden;       /* d, multiple of 4, <= 1024                          */
  int32_t rank;         /* r, 1..64                                           */
  int32_t act;          /* GCONV_ACT_*                                        */
  int32_t skip;         /* skip_connection                                    */
  int32_t walk;         /* 0 / 1: which half of the fused tile kernels' persistent workgroups walks its tiles
                         * smallest-first (batches with two or more tiles per workgroup; ignored otherwise).
                         * Pass the SAME value to the forward and the backward call of an adapter (the backward
                         * takes the other half itself) and ALTERNATE it between consecutive adapters of a model:
                         * every workgroup then starts a kernel with the tile it ended the previous one with —
                         * the rows written last are read first (round 5: whole 4096-molecule batch, 5 chained
                         * adapters, -2.5 % per step). Results: `out` and grad_x do not depend on it; a parameter
                         * gradient is a sum of per-workgroup partials whose tiles are added in walk order — same
                         * bits for the same `walk`, equal within fp32 summation error otherwise.  (ABI 10)  */
} gconv_adapter_params_t;

size_t gconv_adapter_fwd_workspace_bytes(const gconv_graph_t* graph, int32_t hidden, int32_t rank);
size_t gconv_adapter_bwd_workspace_bytes(const gconv_graph_t* graph, int32_t hidden, int32_t rank);

/* replaces: `GConvAdapter.forward` (`gconv_adapter.py:80-108`) for
 * normalization='none':  out = s * (S act(S X Wd^T + bd) Wu^T + bu + X).
 *   graph->dis == NULL selects normalize=False. Block-diagonal graphs whose tiles fit
 *   LDS run as ONE fused launch (tile-local), everything else as four launches.
 *   saved_h  [N, Fp]: act output (row-scaled by dis when normalised), kept for backward
 *   saved_h1 [N, Fp]: pre-activation, written only for GCONV_ACT_SILU (else may be NULL)
 *   saved_p  [N, Fp]: P = S act(...), kept for backward                       */
gconv_status_t gconv_adapter_fwd_f32(const gconv_graph_t* graph,
                                     const gconv_adapter_params_t* params, const float* x,
                                     float* out, float* saved_h, float* saved_h1, float* saved_p,
                                     void* workspace, size_t workspace_bytes,
                                     gconv_stream_t stream);

/* Forward ONLY — replaces `GConvAdapter.forward` where the reference runs it under `torch.no_grad()`:
 * its three evaluations per epoch (`scripts/finetune_inductive_learning.py:95-104,401-409`) and the
 * transductive script's `evaluate` (`scripts/finetune_transductive_learning.py:233-260`). The same launches
 * and the same bits in `out` as gconv_adapter_fwd_f32; nothing is kept for a backward pass: the fused tile
 * path stores `out` only (its rank-width intermediates never leave LDS), the general path keeps them in
 * `workspace` (>= gconv_adapter_infer_workspace_bytes, 16-byte aligned, non-NULL).  (ABI 10)              */
size_t gconv_adapter_infer_workspace_bytes(const gconv_graph_t* graph, int32_t hidden, int32_t rank);
gconv_status_t gconv_adapter_infer_f32(const gconv_graph_t* graph,
                                       const gconv_adapter_params_t* params, const float* x,
                                       float* out, void* workspace, size_t workspace_bytes,
                                       gconv_stream_t stream);

/* Arms `event` (a hipEvent_t; NULL disarms) for the calling thread: the NEXT gconv_adapter_fwd_f32 /
 * gconv_adapter_infer_f32 call of this thread records it at the completion of its last kernel and disarms it
 * (also when that call fails: the event is then not recorded). On the fused tile path the event rides on the
 * kernel's own completion signal (hipExtLaunchKernel's stop event): no packet is queued between this call and
 * the stream's next kernel, where a hipEventRecord behind the call costs the step of a whole 4096-molecule
 * batch 6.5 us of idle GPU in front of the backward pass. Use: the side stream that builds the NEXT batches'
 * graph plans (the reference's DataLoader workers, `scripts/finetune_inductive_learning.py:246-258,291-293`)
 * waits for the forward pass's last adapter this way and runs beside the backward pass.  (ABI 11)            */
gconv_status_t gconv_arm_completion_event(void* event);

/* replaces: the autograd backward of the same ops, entered from
 * `loss.backward()` (`scripts/finetune_inductive_learning.py:66`,
 * `scripts/finetune_transductive_learning.py:229`).
 *   uses graph->bwd (S^T; identical to graph->fwd for symmetric edge lists)
 *   grad_x nullable (skipped when x needs no gradient); grad_scalar nullable.
 *   All grad outputs are overwritten (not accumulated).                      */
gconv_status_t gconv_adapter_bwd_f32(const gconv_graph_t* graph,
                                     const gconv_adapter_params_t* params, const float* x,
                                     const float* grad_out, const float* saved_h,
                                     const float* saved_h1, const float* saved_p,
                                     float* grad_x, float* grad_w_down, float* grad_b_down,
                                     float* grad_w_up, float* grad_b_up, float* grad_scalar,
                                     void* workspace, size_t workspace_bytes,
                                     gconv_stream_t stream);

/* The same backward in two steps, for callers that run several adapters' backward passes
 * before anyone reads the parameter gradients (one `loss.backward()` over a model with 5-10
 * adapters): `..._bwd_partial_f32` does everything except the last kernel — grad_x is complete
 * when it returns, the parameter gradients exist as per-workgroup partials in `workspace` — and
 * describes that last reduction in `*pending` (a host-side POD: pointers into `workspace` and to
 * the grad_* buffers, which must all stay alive and untouched until the reduction has run).
 * `gconv_adapter_bwd_finalize_f32` then runs any number of pending reductions,
 * GCONV_FINALIZE_BATCH per launch, on `stream` (the stream of the partial calls).
 * bwd == bwd_partial + bwd_finalize(…, 1), bit for bit.                                      */
/* (5 descriptors = an 0.8 KB by-value kernel argument; with 8 — 1.26 KB — the launch faulted
 * under rocprofv3's kernel tracing, never outside it: kept below 1 KB) */
#ifndef GCONV_FINALIZE_BATCH
#define GCONV_FINALIZE_BATCH 5
#endif
typedef struct gconv_pending_finalize {
  uint64_t opaque[24];
} gconv_pending_finalize_t;

gconv_status_t gconv_adapter_bwd_partial_f32(const gconv_graph_t* graph,
                                             const gconv_adapter_params_t* params, const float* x,
                                             const float* grad_out, const float* saved_h,
                                             const float* saved_h1, const float* saved_p,
                                             float* grad_x, float* grad_w_down, float* grad_b_down,
                                             float* grad_w_up, float* grad_b_up, float* grad_scalar,
                                             void* workspace, size_t workspace_bytes,
                                             gconv_pending_finalize_t* pending,
                                             gconv_stream_t stream);

gconv_status_t gconv_adapter_bwd_finalize_f32(const gconv_pending_finalize_t* pending,
                                              int32_t count, gconv_stream_t stream);

/* ---- one big graph over several devices: row slabs (single-graph case) ----------
 * The reference has no multi-device path; this is the counterpart of running
 * `GConvAdapter.forward` (`gconv_adapter.py:80-108`) on a graph that is not a batch of
 * components (`scripts/finetune_transductive_learning.py:208-230`), 1-D row partitioned:
 * every device holds the whole graph plan (`gconv_graph_build` on the full edge list — index
 * data only) and the rows [row_lo, row_hi) of X / out / G / dX. The two propagations need the
 * rank-width operand of ALL rows, so the rank-width arrays are full-height [num_nodes, Fp]
 * buffers in GLOBAL row coordinates on every device; a stage writes this device's rows of one
 * of them and the CALLER exchanges it (all-gather: RCCL `ncclAllGather` in place, every device
 * contributing its rows) before the next stage. The library never calls a collective.
 *
 *   forward   stage 0: h0[rows]      = dis . (X Wd^T)                      -> all-gather h0
 *             stage 1: saved_h[rows] = dis . act(S h0 + bd) (saved_h1: pre-activation, SiLU)
 *                                                                          -> all-gather saved_h
 *             stage 2: saved_p[rows] = S saved_h ; out = s (P Wu^T + bu + X)
 *   backward  stage 0: dp[rows]      = dis . s (G Wu)                      -> all-gather dp
 *             stage 1: dh1[rows]     = dis . (S^T dp) . act'               -> all-gather dh1
 *             stage 2: T = S^T dh1 (into dp[rows]) ; grad_x ; this device's CONTRIBUTION to
 *                      the parameter gradients                 -> all-reduce (sum) of those
 *
 * x / out / grad_out / grad_x are the device's own rows ([row_hi - row_lo, hidden], row 0 =
 * global row row_lo). The three backward stages share one `workspace` (partials written by
 * stages 0 and 1 are reduced by stage 2): it must not be touched in between. Uses the same
 * kernels as the single-device path on a row range: results for a row are bitwise those of
 * gconv_adapter_fwd_f32 / _bwd_f32 whatever the partition.                                  */
size_t gconv_adapter_slab_fwd_workspace_bytes(const gconv_graph_t* graph, int32_t rank);
size_t gconv_adapter_slab_bwd_workspace_bytes(const gconv_graph_t* graph, int32_t row_lo,
                                              int32_t row_hi, int32_t hidden, int32_t rank);

gconv_status_t gconv_adapter_slab_fwd_f32(const gconv_graph_t* graph,
                                          const gconv_adapter_params_t* params, int32_t row_lo,
                                          int32_t row_hi, int32_t stage, const float* x,
                                          float* out, float* h0, float* saved_h, float* saved_h1,
                                          float* saved_p, void* workspace,
                                          size_t workspace_bytes, gconv_stream_t stream);

gconv_status_t gconv_adapter_slab_bwd_f32(const gconv_graph_t* graph,
                                          const gconv_adapter_params_t* params, int32_t row_lo,
                                          int32_t row_hi, int32_t stage, const float* x,
                                          const float* grad_out, const float* saved_h,
                                          const float* saved_h1, const float* saved_p, float* dp,
                                          float* dh1, float* grad_x, float* grad_w_down,
                                          float* grad_b_down, float* grad_w_up, float* grad_b_up,
                                          float* grad_scalar, void* workspace,
                                          size_t workspace_bytes, gconv_stream_t stream);

/* ---- the end of the fine-tune step's forward pass (ABI 9) ----------------------
 * replaces: `self.pool(h, batch)` with PyG's global_mean_pool / global_add_pool
 *   (`src/models/inductive/gnn.py:187-195`) and the loop's NaN-masked loss in fp64
 *   (`scripts/finetune_inductive_learning.py:55-63`: `criterion(valid_output.double(), valid_target)` summed
 *   and divided by the number of valid targets; BCEWithLogitsLoss or MSELoss with reduction='none') —
 *   ~25 small torch launches per step.
 * gconv_segment_pool_f32: out[g, :] = sum (mean != 0: mean) of x[i, :] over the nodes with batch[i] == g,
 *   g in [0, num_graphs); `batch` int64, SORTED ascending (PyG's collation order); rows are added in node
 *   order — no atomics; counts (nullable, [num_graphs + 1]): counts[g] = the graph's node count, and
 *   counts[num_graphs] = 1 if `batch` turned out NOT to be sorted or to hold ids outside [0, num_graphs) (ABI 10:
 *   the output rows of the graphs where that shows are NaN — the loss carries it to the host without a
 *   synchronisation; 0 otherwise); an empty graph gives zeros.
 * gconv_segment_pool_bwd_f32: grad_x[i, :] = grad_out[batch[i], :] (/ counts[batch[i]] for the mean).
 * gconv_masked_loss_f64: logits fp32 [count], target fp64 [count] with NaN = no label; kind 0 = BCE with
 *   logits, 1 = squared error, both in fp64; loss_and_count[0] = sum over the labelled / their number (0 if
 *   none), [1] = that number; grad_logits[i] = d loss / d logits[i] (fp32; 0 where there is no label).
 * gconv_masked_loss_ws_f64 (ABI 10): the same loss for many logits (multi-task heads: 617 tasks x 4096 graphs) as
 *   two launches of up to 256 / 4096 workgroups — block partials in a fixed order through `workspace`
 *   (>= gconv_masked_loss_workspace_bytes(count), zero-initialised ONCE by the caller, left zeroed where it must
 *   be by every call); same results as gconv_masked_loss_f64 up to the summation order, bitwise repeatable.
 * hidden a multiple of 4, <= 1024.                                                                     */
gconv_status_t gconv_segment_pool_f32(const float* x, const int64_t* batch, int32_t num_nodes, int32_t hidden,
                                      int32_t num_graphs, int32_t mean, float* out, int32_t* counts,
                                      gconv_stream_t stream);
gconv_status_t gconv_segment_pool_bwd_f32(const float* grad_out, const int64_t* batch, const int32_t* counts,
                                          int32_t num_nodes, int32_t hidden, int32_t num_graphs, int32_t mean,
                                          float* grad_x, gconv_stream_t stream);
gconv_status_t gconv_masked_loss_f64(const float* logits, const double* target, int64_t count, int32_t kind,
                                     double* loss_and_count, float* grad_logits, gconv_stream_t stream);
size_t gconv_masked_loss_workspace_bytes(int64_t count);
gconv_status_t gconv_masked_loss_ws_f64(const float* logits, const double* target, int64_t count, int32_t kind,
                                        double* loss_and_count, float* grad_logits, void* workspace,
                                        size_t workspace_bytes, gconv_stream_t stream);

/* ---- next to the path: the optimizer step over the trainable set -------------
 * replaces: `optim.Adam(...)` + `optimizer.step()`
 *   (`scripts/finetune_inductive_learning.py:379,67`, `scripts/finetune_transductive_learning.py:205,230`)
 * for parameters kept in ONE flat fp32 range (adapters + head + norm affine, ~29 k floats): one
 * launch instead of torch's ~10 multi-tensor launches. Non-amsgrad Adam with L2 weight decay,
 * torch's update rule and operation order; `step` is the 1-based step count of this call
 * (bias corrections are computed from it in double on the host, as torch does). All four
 * arrays have `count` floats; param / exp_avg / exp_avg_sq are updated in place.            */
gconv_status_t gconv_adam_step_f32(float* param, const float* grad, float* exp_avg,
                                   float* exp_avg_sq, int64_t count, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, int64_t step,
                                   gconv_stream_t stream);

/* ---- next to the path: collation of a batch from a device-resident dataset -----
 * replaces: PyG `DataLoader` -> `Batch.from_data_list` -> `batch.to(device)`
 *   (`scripts/finetune_inductive_learning.py:291-293,45`) for datasets kept in HBM: the chosen
 * graphs' feature rows are concatenated, their (graph-local) edge ids offset and widened to
 * int64 [2, E], the graph offsets and the node -> graph vector written; two launches, no
 * synchronisation. Dataset layout: graphs back to back, `node_ptr` / `edge_ptr` [G + 1] device
 * int32 offsets, `edge_src` / `edge_dst` graph-LOCAL node ids with each graph's edges contiguous.
 * `num_edges_out` (= sum of the chosen graphs' edge counts, known to the host from its copy of
 * edge_ptr) is the row stride of `edge_index_out`; `x_out` has sum-of-nodes rows of `hidden`
 * floats; `batch_out` (node -> position in the batch) is optional.                             */
size_t gconv_collate_workspace_bytes(int32_t num_graphs);
gconv_status_t gconv_collate_f32(const float* x_all, int32_t hidden, const int32_t* node_ptr,
                                 const int32_t* edge_ptr, const int32_t* edge_src,
                                 const int32_t* edge_dst, const int32_t* graph_ids,
                                 int32_t num_graphs, int64_t num_edges_out, float* x_out,
                                 int64_t* edge_index_out, int32_t* graph_ptr_out,
                                 int32_t* batch_out, void* workspace, size_t workspace_bytes,
                                 gconv_stream_t stream);

/* The graph plan of such a batch without building it: `dataset` is the plan of the WHOLE
 * dataset's edge list (one gconv_graph_build at load time); the chosen graphs' CSR rows are
 * re-based, their columns offset, `dis` copied, tiles cut at graph boundaries and sorted by size.
 * No sort of the edges, no host synchronisation: `*graph_host_out` is complete when the call
 * returns, its arrays when `stream` reaches this point. `nnz_ptr[g]` = dataset rowptr at graph
 * g's first node ([G + 1], device); `num_nodes`, `nnz`, `max_graph_nodes` (largest chosen graph)
 * come from the host's copy of the dataset offsets. Needs what molecule datasets have — a
 * symmetric edge list and no row above GCONV_LONG_ROW edges — else GCONV_E_UNSUPPORTED
 * (use gconv_graph_build on the collated edge_index).                                         */
size_t gconv_graph_from_store_bytes(int32_t num_nodes, int64_t nnz, int32_t num_graphs);
gconv_status_t gconv_graph_from_store(const gconv_graph_t* dataset, const int32_t* node_ptr,
                                      const int32_t* nnz_ptr, const int32_t* graph_ids,
                                      int32_t num_graphs, int32_t num_nodes, int64_t nnz,
                                      int32_t max_graph_nodes, void* arena, size_t arena_bytes,
                                      gconv_graph_t* graph_host_out, gconv_stream_t stream);

/* ---- next to the path: aggregation of the frozen GIN layer between two adapters --
 * replaces: `MolecularGINConv.forward` up to its MLP (`src/layers/inductive/gin_conv.py:38-73`):
 * `add_self_loops`, the bond-type / bond-direction embedding look-ups (self-loops: type 4,
 * direction 0), message `x_j + e_ij`, sum aggregation (`aggr="add"`):
 *   agg[i] = x[i] + sum_{j->i} x[j] + emb_type[4] + emb_dir[0] + sum_{e: j->i} (emb_type[t_e] + emb_dir[d_e])
 * `gconv_gin_code_count` builds, once per batch, the per-node counts of the 18 (type, direction)
 * codes of the incoming edges ([num_nodes, 18] int32, + one trailing int = number of edges skipped
 * because an endpoint or attribute was out of range; `edge_index` int64 [2, E], `edge_attr` int64
 * [E, 2] as PyG collates them); `gconv_gin_aggregate_f32` walks a CSR of the batch's UNNORMALISED
 * graph plan (`gconv_graph_build(..., normalize = 0, ...)`: `fwd` for the layer's forward; `bwd`
 * with code_count == NULL gives the gradient w.r.t. x, d agg -> d x). emb_type [6, hidden],
 * emb_dir [3, hidden]; hidden a multiple of 4, <= 1024. The MLP and BatchNorm that follow
 * (`gin_conv.py:24-28`, `gnn.py:108-110`) are plain dense layers and stay with the BLAS.        */
size_t gconv_gin_code_count_bytes(int32_t num_nodes);
gconv_status_t gconv_gin_code_count(const int64_t* edge_index, const int64_t* edge_attr,
                                    int64_t num_edges, int32_t num_nodes, int32_t* code_count,
                                    gconv_stream_t stream);
gconv_status_t gconv_gin_aggregate_f32(const gconv_csr_t* csr, const float* x,
                                       const int32_t* code_count, const float* emb_type,
                                       const float* emb_dir, float* out, int32_t hidden,
                                       gconv_stream_t stream);

/* ---- next to the path: the adapter's `layer_norm` tail ---------------------------
 * replaces: `out = self.normalization(out)` for nn.LayerNorm and `out = out * self.scalar`
 * (`gconv_adapter.py:101-102,105-106`) — and their autograd backward — with one pass each way:
 *   out = scalar * (gamma . (z - mean) * rstd + beta),  per-row mean / biased variance, eps inside
 * the square root (torch.nn.LayerNorm). gamma, beta, scalar (device, 1 float) nullable. `stats`
 * [num_nodes, 2] receives (mean, rstd) for the backward. The backward overwrites grad_z (nullable),
 * grad_gamma / grad_beta [hidden] and grad_scalar [1] (each nullable); hidden a multiple of 4,
 * <= 1024. Deterministic (fixed-order reductions, no atomics).                                  */
size_t gconv_layer_norm_scale_bwd_workspace_bytes(int32_t num_nodes, int32_t hidden);
gconv_status_t gconv_layer_norm_scale_fwd_f32(const float* z, const float* gamma, const float* beta,
                                              const float* scalar, float* out, float* stats,
                                              int32_t num_nodes, int32_t hidden, float eps,
                                              gconv_stream_t stream);
gconv_status_t gconv_layer_norm_scale_bwd_f32(const float* grad_out, const float* z,
                                              const float* stats, const float* gamma,
                                              const float* beta, const float* scalar, float* grad_z,
                                              float* grad_gamma, float* grad_beta,
                                              float* grad_scalar, int32_t num_nodes, int32_t hidden,
                                              void* workspace, size_t workspace_bytes,
                                              gconv_stream_t stream);

/* ---- next to the path: BatchNorm1d over node features, batch statistics --------
 * replaces: `h = self.batch_norm_layers[layer](h)` after every GIN layer
 * (`src/models/inductive/gnn.py:108-110`; BatchNorm parameters train under the freeze rule,
 * `gnn.py:236-239`) — optionally with the ReLU that follows it (`gnn.py:127-128`) folded in — and
 * the adapter's own `batch_norm` tail (`gconv_adapter.py:98-100`); and their autograd backward.
 *   forward (training): per-column mean / biased variance over the num_nodes rows (>= 2), y =
 *     [relu](gamma (x - mean) rstd + beta); save_mean / save_rstd [hidden] kept for backward;
 *     running_mean / running_var (nullable) updated in place with torch's rule (momentum, unbiased
 *     variance). `gconv_batch_norm_apply_f32` is the evaluation-mode forward with given mean / rstd.
 *   backward: grad_x (nullable), grad_gamma, grad_beta (nullable) overwritten; `y` = the forward's
 *     output, needed only when the forward folded the ReLU in (else NULL).
 * Three launches each way (column sums with persistent waves, fixed-order reduction, apply), no
 * atomics; hidden a multiple of 4, <= 1024; one workspace size for both directions.             */
size_t gconv_batch_norm_workspace_bytes(int32_t num_nodes, int32_t hidden);
gconv_status_t gconv_batch_norm_fwd_f32(const float* x, const float* gamma, const float* beta,
                                        float* running_mean, float* running_var, float momentum,
                                        float eps, int32_t relu, float* y, float* save_mean,
                                        float* save_rstd, int32_t num_nodes, int32_t hidden,
                                        void* workspace, size_t workspace_bytes,
                                        gconv_stream_t stream);
/* ABI 9: the same forward that also counts the batch (`num_batches_tracked += 1`, nn.BatchNorm1d's int64
 * buffer; NULL = do not count) — inside its reduction launch instead of a launch of torch's.          */
gconv_status_t gconv_batch_norm_fwd_tracked_f32(const float* x, const float* gamma, const float* beta,
                                                float* running_mean, float* running_var,
                                                int64_t* num_batches_tracked, float momentum, float eps,
                                                int32_t relu, float* y, float* save_mean, float* save_rstd,
                                                int32_t num_nodes, int32_t hidden, void* workspace,
                                                size_t workspace_bytes, gconv_stream_t stream);
gconv_status_t gconv_batch_norm_apply_f32(const float* x, const float* mean, const float* rstd,
                                          const float* gamma, const float* beta, int32_t relu,
                                          float* y, int32_t num_nodes, int32_t hidden,
                                          gconv_stream_t stream);
gconv_status_t gconv_batch_norm_bwd_f32(const float* grad_y, const float* x, const float* y,
                                        const float* save_mean, const float* save_rstd,
                                        const float* gamma, float* grad_x, float* grad_gamma,
                                        float* grad_beta, int32_t num_nodes, int32_t hidden,
                                        void* workspace, size_t workspace_bytes,
                                        gconv_stream_t stream);

/* ---- the same BatchNorm when the batch's rows are sharded BY GRAPH over ranks (ABI 8) ----------
 * The reference is single-process: `batch_norm_layers[layer](h)` (`src/models/inductive/gnn.py:56,
 * 109-110`) and the adapter's `batch_norm` tail (`src/finetune/gconv_adapter.py:98-100`) normalise with
 * the statistics of the WHOLE batch. With one process per GPU and a shard of the batch each (SURVEY
 * §8e: "optional all_reduce of BN stats") the same result needs one small exchange per direction.
 * The library never calls a collective: both passes are cut where the exchange happens and the
 * caller (gconv_adapter_amd/norm.py over torch.distributed / RCCL) runs it in between.
 *   forward : local_stats  -> `local` [gconv_batch_norm_sync_words(hidden)] = [mean d][M2 d][count, 3 pad]
 *                              (M2 = sum (x - mean)^2 over this rank's num_nodes >= 0 rows; the count
 *                              is stored as int32 BITS so that an all-gather carries it unchanged)
 *             ALL-GATHER   -> `gathered` [world][sync_words], rank order
 *             fwd_sync     -> merges the ranks' (count, mean, M2) in rank order (Chan's update: same
 *                              bits on every rank), writes save_mean / save_rstd / *total_count (device),
 *                              updates the running statistics with torch's rule over the global count,
 *                              y = [relu](gamma (x - mean) rstd + beta) for this rank's rows
 *   backward: bwd_local    -> `sums` [2 hidden] = [sum g' xhat][sum g'] over this rank's rows: the rank's
 *                              CONTRIBUTION to grad_gamma / grad_beta (summed later by the step's flat
 *                              gradient all-reduce like every other parameter gradient)
 *             ALL-REDUCE(sum) of a COPY of `sums` -> `sums_all`
 *             bwd_dx       -> grad_x = gamma rstd (g' - sum_all g' / n_all - xhat sum_all g' xhat / n_all)
 * Workspace size: gconv_batch_norm_workspace_bytes. An empty shard (num_nodes = 0) is valid.      */
size_t gconv_batch_norm_sync_words(int32_t hidden);
gconv_status_t gconv_batch_norm_local_stats_f32(const float* x, int32_t num_nodes, int32_t hidden,
                                                float* local, void* workspace, size_t workspace_bytes,
                                                gconv_stream_t stream);
gconv_status_t gconv_batch_norm_fwd_sync_f32(const float* x, const float* gamma, const float* beta,
                                             float* running_mean, float* running_var, float momentum,
                                             float eps, int32_t relu, float* y, float* save_mean,
                                             float* save_rstd, int32_t* total_count, int32_t num_nodes,
                                             int32_t hidden, const float* gathered, int32_t world,
                                             gconv_stream_t stream);
gconv_status_t gconv_batch_norm_bwd_local_f32(const float* grad_y, const float* x, const float* y,
                                              const float* save_mean, const float* save_rstd,
                                              float* sums, int32_t num_nodes, int32_t hidden,
                                              void* workspace, size_t workspace_bytes,
                                              gconv_stream_t stream);
gconv_status_t gconv_batch_norm_bwd_dx_f32(const float* grad_y, const float* x, const float* y,
                                           const float* save_mean, const float* save_rstd,
                                           const float* gamma, const float* sums_all,
                                           const int32_t* total_count, float* grad_x, int32_t num_nodes,
                                           int32_t hidden, gconv_stream_t stream);

/* ---- next to the path: the frozen GIN layer's MLP -------------------------------
 * replaces: `self.mlp` = Linear(d, 2d) -> ReLU -> Linear(2d, d) of `MolecularGINConv`
 * (`src/layers/inductive/gin_conv.py:24-28`, applied in `update`, `:85`), forward and INPUT gradient
 * (the layer is frozen under the adapter recipe, `src/models/inductive/gnn.py:230-241`).
 * fp32 operands as sums of three bf16 values, six bf16 MFMA products per fp32 product, fp32
 * accumulation: fp32-ACCURATE (relative error of a product <= 2^-23), not bit-equal to an fp32 GEMM.
 * Every value is split ONCE: the frozen weights at load time, the layer input / upstream gradient
 * by one pass, the hidden activation by the epilogue of the GEMM that produces it (it never exists
 * in fp32). "bf16 triple" of a [R, F] matrix = three bf16 images [3][R][Fp], Fp = F rounded up to a
 * multiple of 32, padding columns zero; all triples 16-byte aligned.
 *   gconv_mlp_split_weights_bf16x3: W [rows, cols] fp32 (or, `transpose`, its transpose) -> its triple
 *     with the feature rows padded as well ([3][Mp][Kp]; `out` holds gconv_mlp_split_bytes(M, K) bytes).
 *   gconv_mlp_split_rows_bf16x3: X [N, F] fp32 -> its triple (gconv_mlp_rows_split_bytes(N, F) bytes).
 *   gconv_mlp_gemm_bf16x3: C [N, M] = epi(A . W^T + bias), A a row triple, W a weight triple;
 *     exactly one of c_f32 ([N, M] fp32, M a multiple of 4) / c_split (the triple of C) is non-NULL;
 *     epilogue 0: none, 1: ReLU, 2: multiply by [mask > 0], `mask_split` = the triple of a ReLU
 *     output of C's shape (its first image is read). */
size_t gconv_mlp_split_bytes(int32_t out_features, int32_t in_features);
size_t gconv_mlp_rows_split_bytes(int32_t num_rows, int32_t features);
gconv_status_t gconv_mlp_split_weights_bf16x3(const float* w, int32_t rows, int32_t cols, int32_t transpose,
                                              void* out, gconv_stream_t stream);
gconv_status_t gconv_mlp_split_rows_bf16x3(const float* x, int32_t num_rows, int32_t features, void* out,
                                           gconv_stream_t stream);
gconv_status_t gconv_mlp_gemm_bf16x3(const void* a_split, const void* w_split, const float* bias,
                                     const void* mask_split, float* c_f32, void* c_split, int32_t num_rows,
                                     int32_t in_features, int32_t out_features, int32_t epilogue,
                                     gconv_stream_t stream);

/* ---- the same MLP on the fp16 MFMA (round 4, ABI 8): two images + power-of-two scaling ---------
 * replaces the same reference lines as the bf16 triple above (`gin_conv.py:24-28,85`); half the matrix
 * work, two thirds of the operand bytes, the same fp32 accuracy (`csrc/mlp_f16.hip`). An fp32 matrix
 * travels as a "scaled pair": x 2^e = hi + lo in fp16 (11 + 11 mantissa bits), with ONE exponent e per row
 * of an activation / gradient matrix (the row's largest magnitude, or a bound of it, lands in [2^13, 2^14))
 * and one per weight matrix; |x| < 2^100.
 *   row pair    [2][N][Fp] fp16 images (Fp = F rounded up to 32, padding zero) followed, 16-byte aligned,
 *               by int32 e[N]: gconv_mlp_rows_pair_bytes(N, F) bytes
 *   weight pair [2][Mp][Kp] fp16 images of W 2^e followed by 4 floats {2^-e, max row 1-norm, max |bias|,
 *               e as int bits}: gconv_mlp_weights_pair_bytes(M, K) bytes; `bias` (nullable) is the bias the
 *               GEMM using this pair will add — its maximum enters the bound of the hidden activation
 *   gconv_mlp_gemm_f16x2: C = epi(A W^T + bias) in true units; exactly one of `c_f32` ([N, M] fp32) and
 *     `c_pair` (row pair, exponents written) is non-NULL. epilogue 0 none, 1 ReLU, 2 multiply by
 *     [mask > 0], `mask_pair` = the row pair of a ReLU output ([N, M]: its hi image is read).
 *   gconv_gin_aggregate_f16x2: `gconv_gin_aggregate_f32` with the bond term, the result written as the row
 *     pair the first GEMM multiplies — the fp32 aggregate and the split pass over it never exist.      */
size_t gconv_mlp_weights_pair_bytes(int32_t out_features, int32_t in_features);
size_t gconv_mlp_rows_pair_bytes(int32_t num_rows, int32_t features);
gconv_status_t gconv_mlp_split_weights_f16x2(const float* w, const float* bias, int32_t rows, int32_t cols,
                                             int32_t transpose, void* out, gconv_stream_t stream);
gconv_status_t gconv_mlp_split_rows_f16x2(const float* x, int32_t num_rows, int32_t features, void* out,
                                          gconv_stream_t stream);
gconv_status_t gconv_mlp_gemm_f16x2(const void* a_pair, const void* w_pair, const float* bias,
                                    const void* mask_pair, float* c_f32, void* c_pair, int32_t num_rows,
                                    int32_t in_features, int32_t out_features, int32_t epilogue,
                                    gconv_stream_t stream);
gconv_status_t gconv_gin_aggregate_f16x2(const gconv_csr_t* csr, const float* x, const int32_t* code_count,
                                         const float* emb_type, const float* emb_dir, void* out_pair,
                                         int32_t hidden, gconv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GCONV_ADAPTER_H */
