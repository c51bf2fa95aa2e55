// Whole-adapter forward / backward: kernel sequencing and workspace carving.
//
// Replaces GConvAdapter.forward (/root/reference/src/finetune/gconv_adapter.py:80-108,
// normalization='none') and its autograd backward. Identical math, reassociated
// so that BOTH propagations run at rank r (PyG's conv_up propagates at width d):
//
//   forward   H0' = dis . (X Wd^T)                         K1  gemm_rowsA<DOWN>
//             H2' = dis . act(dis . (A0 H0' + H0') + bd)   K2  spmm<EPI_FWD1>
//             P   =       dis . (A0 H2' + H2')             K2' spmm<EPI_STORE>
//             out = s (P Wu^T + bu + X)                    K3  up_fwd
//   backward  dP' = dis . s (G Wu)          ; ds1          B1  gemm_rowsA<DP>
//             dH1'= dis . (S^T dP) . act'   ; dbd          B2  spmm<EPI_BWD1> on S^T
//             T   = S^T dH1                                B3  spmm<EPI_STORE> on S^T
//             dX  = T Wd + s G ; dWd^T, dWu^T, colsum(G)   B4  bwd_main
//             parameter gradients                          B5  finalize
// (dis . = row scaling by deg^-1/2; without normalisation dis == 1 and the
// implicit self-loop term disappears.)
#include <cstring>

#include "fused_gather.h"
#include "fused_tile.h"
#include "gemm.h"
#include "spmm.h"

namespace gconv {

struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t count) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += align_up(count * sizeof(T) + 16, 256);
    return p;
  }
};

struct FwdWs {
  float* h0;
  float* seg;
  size_t bytes;
};
static FwdWs carve_fwd(void* ws, const gconv_csr_t* g, int fp) {
  Carver c(ws);
  FwdWs w;
  w.h0 = c.take<float>(static_cast<size_t>(g->num_nodes) * fp);
  w.seg = c.take<float>(static_cast<size_t>(g->n_seg) * fp);
  w.bytes = c.off;
  return w;
}

struct BwdWs {
  float* dp;    // dP', later reused for T
  float* dh1;
  float* seg;
  float* part_ds1;
  float* part_bd;
  float* part_bd2;  // stage-1 reduction of part_bd when it is tall
  float* part_wd;
  float* part_wu;
  float* part_cs;
  float* part_dsx;
  float* cs_tmp;
  int n_ds1, n_bd;
  BwdMainGeom geom;
  size_t bytes;
};
// `rows` = rows this device produces (the whole graph, or its row slab: then the two
// full-height rank-width buffers are the caller's — they are exchanged between stages)
static BwdWs carve_bwd(void* ws, const gconv_csr_t* gt, int d, int fp, int rows, bool slab) {
  Carver c(ws);
  BwdWs w;
  const size_t n = slab ? 0 : gt->num_nodes;
  w.geom = bwd_main_geom(rows, d);
  w.n_ds1 = rowsA_grid(rows, d, fp);
  w.n_bd = spmm_bias_partial_rows(gt, fp, rows);
  w.dp = c.take<float>(n * fp);
  w.dh1 = c.take<float>(n * fp);
  w.seg = c.take<float>(static_cast<size_t>(gt->n_seg) * fp);
  w.part_ds1 = c.take<float>(w.n_ds1);
  w.part_bd = c.take<float>(static_cast<size_t>(w.n_bd) * fp);
  w.part_bd2 = c.take<float>(static_cast<size_t>(kReduceBlocks) * fp);
  w.part_wd = c.take<float>(static_cast<size_t>(w.geom.nparts) * fp * d);
  w.part_wu = c.take<float>(static_cast<size_t>(w.geom.nparts) * fp * d);
  w.part_cs = c.take<float>(static_cast<size_t>(w.geom.nparts) * d);
  w.part_dsx = c.take<float>(static_cast<size_t>(w.geom.nparts) * w.geom.nchunks * w.geom.nslots);
  w.cs_tmp = c.take<float>(d);
  w.bytes = c.off;
  return w;
}

// backward workspace of the fused tile path: one partial row per tile
struct TileBwdWs {
  float *part_wd, *part_wu, *part_cs, *part_bd, *part_ds, *cs_tmp;
  int nparts;
  size_t bytes;
};
static TileBwdWs carve_bwd_tile(void* ws, int n_tiles, int d, int fp) {
  Carver c(ws);
  TileBwdWs w;
  w.nparts = tile_grid(n_tiles);
  const size_t t = w.nparts;
  w.part_wd = c.take<float>(t * fp * d);
  w.part_wu = c.take<float>(t * fp * d);
  w.part_cs = c.take<float>(t * d);
  w.part_bd = c.take<float>(t * fp);
  w.part_ds = c.take<float>(t);
  w.cs_tmp = c.take<float>(d);
  w.bytes = c.off;
  return w;
}

static gconv_status_t check_common(const gconv_csr_t* g, const gconv_adapter_params_t* p) {
  if (!g || !p || !g->rowptr || (g->nnz > 0 && !g->col) || g->num_nodes < 0)
    return GCONV_E_INVALID_ARG;
  if (p->hidden <= 0 || p->rank <= 0 || !p->w_down || !p->w_up) return GCONV_E_INVALID_ARG;
  if (p->act < GCONV_ACT_NONE || p->act > GCONV_ACT_SILU) return GCONV_E_INVALID_ARG;
  if (p->hidden % 4 != 0 || p->hidden > 1024 || padded_rank(p->rank) == 0)
    return GCONV_E_UNSUPPORTED;
  if (rowsA_lds_bytes(p->hidden, padded_rank(p->rank)) > kRowsAMaxLds) return GCONV_E_UNSUPPORTED;
  if (g->n_long > 0 && (!g->long_rows || !g->long_seg_base || !g->seg_long))
    return GCONV_E_INVALID_ARG;
  return GCONV_OK;
}

}  // namespace gconv

using namespace gconv;

extern "C" int32_t gconv_abi_version(void) { return GCONV_ABI_VERSION; }
extern "C" int32_t gconv_abi_struct_bytes(int32_t which) {
  switch (which) {
    case 0: return sizeof(gconv_csr_t);
    case 1: return sizeof(gconv_graph_t);
    case 2: return sizeof(gconv_adapter_params_t);
    case 3: return sizeof(gconv_pending_finalize_t);
    case 4: return sizeof(gconv_tile_t);
    case 5: return sizeof(gconv_tiling_t);
    default: return -1;
  }
}

extern "C" int32_t gconv_padded_rank(int32_t rank) { return padded_rank(rank); }

extern "C" const char* gconv_status_string(gconv_status_t status) {
  switch (status) {
    case GCONV_OK: return "success";
    case GCONV_E_INVALID_ARG: return "invalid argument";
    case GCONV_E_UNSUPPORTED:
      return "unsupported shape (hidden must be a multiple of 4 and <= 1024, rank in 1..64)";
    case GCONV_E_WORKSPACE: return "workspace too small";
    case GCONV_E_BAD_INDEX: return "edge_index entry outside [0, num_nodes)";
    case GCONV_E_ALIGNMENT: return "float buffer is not 16-byte aligned";
  }
  if (status > 0) return hipGetErrorString(static_cast<hipError_t>(status));
  return "unknown gconv status";
}

extern "C" size_t gconv_adapter_fwd_workspace_bytes(const gconv_graph_t* graph, int32_t hidden,
                                                    int32_t rank) {
  (void)hidden;
  const int fp = padded_rank(rank);
  if (!graph || fp == 0) return 0;
#ifdef GCONV_TILE_STAMPS
  if (tile_path_ok(graph, hidden, fp)) return 1 << 20;
#endif
  if (tile_path_ok(graph, hidden, fp)) return 0;   // the fused tile kernel keeps its state in LDS
  return carve_fwd(nullptr, &graph->fwd, fp).bytes;
}

extern "C" size_t gconv_adapter_bwd_workspace_bytes(const gconv_graph_t* graph, int32_t hidden,
                                                    int32_t rank) {
  const int fp = padded_rank(rank);
  if (!graph || fp == 0 || hidden <= 0) return 0;
#ifdef GCONV_TILE_STAMPS
  if (tile_path_ok(graph, hidden, fp)) return carve_bwd_tile(nullptr, graph->n_tiles, hidden, fp).bytes + (1 << 20);
#endif
  if (tile_path_ok(graph, hidden, fp)) return carve_bwd_tile(nullptr, graph->n_tiles, hidden, fp).bytes;
  return carve_bwd(nullptr, &graph->bwd, hidden, fp, graph->bwd.num_nodes, false).bytes;
}

// The four-launch forward on the rows [lo, hi) of the graph, `stages` = bit set of
//   1: K1   2: K2   4: K2' + K3
// x / out hold the rows [lo, hi) only (row 0 = global row lo); h0, saved_* are full-height
// arrays in global row coordinates. The single-device path is lo = 0, hi = n, stages = 7.
static gconv_status_t fwd_stages(const gconv_graph_t* gr, const gconv_adapter_params_t* p, int lo,
                                 int hi, int stages, const float* x, float* out, float* h0,
                                 float* saved_h, float* saved_h1, float* saved_p, float* seg,
                                 hipStream_t stream) {
  const gconv_csr_t* graph = &gr->fwd;
  const float* dis = gr->dis;
  const int rows = hi - lo, d = p->hidden, r = p->rank, fp = padded_rank(r);
  const size_t shift = static_cast<size_t>(lo) * fp;
  gconv_status_t st = GCONV_OK;
  if (stages & 1) {
    RowsAArgs k1{};
    k1.X = x; k1.W = p->w_down; k1.row_scale = dis ? dis + lo : nullptr; k1.out = h0 + shift;
    k1.n = rows; k1.d = d; k1.r = r; k1.fp = fp; k1.mode = ROWSA_DOWN;
    st = launch_rowsA(k1, stream);
    if (st != GCONV_OK) return st;
  }
  if (stages & 2) {
    SpmmArgs k2{};
    k2.dis = dis; k2.H = h0; k2.Y = saved_h; k2.Y2 = saved_h1;
    k2.bias = p->b_down; k2.act = p->act; k2.rank = r; k2.seg_partial = seg;
    k2.row_lo = lo; k2.row_hi = hi;
    st = launch_spmm(graph, k2, fp, EPI_FWD1, stream);
    if (st != GCONV_OK) return st;
  }
  if (stages & 4) {
    SpmmArgs k2b{};
    k2b.dis = dis; k2b.H = saved_h; k2b.Y = saved_p; k2b.rank = r; k2b.seg_partial = seg;
    k2b.row_lo = lo; k2b.row_hi = hi;
    if (lo == 0 && hi == graph->num_nodes && fused_gather_enabled(graph, d)) {
      // K2' + K3 in one launch (fused_gather.hip); only the long rows' segments run in front
      if (graph->n_seg > 0) {
        k2b.seg_only = 1;
        st = launch_spmm(graph, k2b, fp, EPI_STORE, stream);
        if (st != GCONV_OK) return st;
      }
      UpGatherArgs f{};
      f.rowptr = graph->rowptr; f.col = graph->col; f.dis = dis; f.H = saved_h; f.P = saved_p;
      f.up.W_up = p->w_up; f.up.b_up = p->b_up; f.up.scalar = p->scalar; f.up.X = x; f.up.out = out;
      f.up.n = rows; f.up.d = d; f.up.r = r; f.up.fp = fp; f.up.skip = p->skip;
      return launch_up_gather(f, stream);
    }
    st = launch_spmm(graph, k2b, fp, EPI_STORE, stream);
    if (st != GCONV_OK) return st;

    UpArgs k3{};
    k3.P = saved_p + shift; k3.W_up = p->w_up; k3.b_up = p->b_up; k3.scalar = p->scalar;
    k3.X = x; k3.out = out; k3.n = rows; k3.d = d; k3.r = r; k3.fp = fp; k3.skip = p->skip;
    st = launch_up_fwd(k3, stream);
  }
  return st;
}

// `keep` = the caller's saved_* buffers are the outputs of a training forward; !keep = forward only
// (gconv_adapter_infer_f32): the tile path stores nothing but `out`, the general path carves its two or three
// rank-width intermediates out of the workspace behind the forward's own scratch.
// The armed completion event of the calling thread (gconv_arm_completion_event): taken — armed or not used — by the
// next forward call, whatever that call returns.
static thread_local void* t_armed_event = nullptr;
extern "C" gconv_status_t gconv_arm_completion_event(void* event) {
  t_armed_event = event;
  return GCONV_OK;
}

static gconv_status_t adapter_forward(const gconv_graph_t* gr, const gconv_adapter_params_t* p, const float* x,
                                      float* out, float* saved_h, float* saved_h1, float* saved_p, bool keep,
                                      void* workspace, size_t workspace_bytes, gconv_stream_t stream_) {
  void* const done_event = t_armed_event;
  t_armed_event = nullptr;
  if (!gr) return GCONV_E_INVALID_ARG;
  const gconv_csr_t* graph = &gr->fwd;
  const float* dis = gr->dis;
  gconv_status_t st = check_common(graph, p);
  if (st != GCONV_OK) return st;
  const int n = graph->num_nodes, d = p->hidden, r = p->rank, fp = padded_rank(r);
  if (n == 0) return GCONV_OK;
  if (!x || !out || !workspace) return GCONV_E_INVALID_ARG;
  if (keep && (!saved_h || !saved_p)) return GCONV_E_INVALID_ARG;
  if (keep && p->act == GCONV_ACT_SILU && !saved_h1) return GCONV_E_INVALID_ARG;
  if (!aligned16(x) || !aligned16(out) || (saved_h && !aligned16(saved_h)) || (saved_p && !aligned16(saved_p)) ||
      (saved_h1 && !aligned16(saved_h1)) || !aligned16(workspace) ||
      (p->b_up && !aligned16(p->b_up)) || !aligned16(p->w_down))
    return GCONV_E_ALIGNMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (tile_path_ok(gr, d, fp)) {  // block-diagonal batch: one fused launch
    TileFwdArgs t{};
    t.tiles = gr->tiles_fwd.tiles; t.lcol = gr->tiles_fwd.lcol; t.halo = gr->tiles_fwd.halo;
    t.rowptr = graph->rowptr; t.dis = dis; t.n = n;
    t.x = x; t.out = out; t.saved_h = saved_h; t.saved_h1 = saved_h1; t.saved_p = saved_p;
    t.w_down = p->w_down; t.b_down = p->b_down; t.w_up = p->w_up; t.b_up = p->b_up;
    t.scalar = p->scalar; t.d = d; t.r = r; t.fp = fp; t.act = p->act; t.skip = p->skip;
    t.stagger = p->walk & 1;
#ifdef GCONV_TILE_STAMPS
    t.stamps = static_cast<long long*>(workspace);   // diagnostic build: workspace = stamp buffer
#endif
    return launch_tile_fwd(t, gr->n_tiles, gr->tile_nodes_max, gr->tile_edges_max, stream, done_event);
  }
  const FwdWs ws = carve_fwd(workspace, graph, fp);
  size_t need = ws.bytes;
  if (!keep) {
    Carver c(workspace);
    c.off = ws.bytes;
    const size_t rows = static_cast<size_t>(n) * fp;
    saved_h = c.take<float>(rows);
    saved_p = c.take<float>(rows);
    saved_h1 = p->act == GCONV_ACT_SILU ? c.take<float>(rows) : nullptr;
    need = c.off;
  }
  if (workspace_bytes < need) return GCONV_E_WORKSPACE;
  st = fwd_stages(gr, p, 0, n, 7, x, out, ws.h0, saved_h, saved_h1, saved_p, ws.seg, stream);
  if (st == GCONV_OK && done_event && hipEventRecord(static_cast<hipEvent_t>(done_event), stream) != hipSuccess)
    return GCONV_E_INVALID_ARG;   // (several launches: an ordinary record behind the last one)
  return st;
}

extern "C" gconv_status_t gconv_adapter_fwd_f32(const gconv_graph_t* gr,
                                                const gconv_adapter_params_t* p, const float* x,
                                                float* out, float* saved_h, float* saved_h1,
                                                float* saved_p, void* workspace,
                                                size_t workspace_bytes, gconv_stream_t stream_) {
  return adapter_forward(gr, p, x, out, saved_h, saved_h1, saved_p, true, workspace, workspace_bytes, stream_);
}

extern "C" size_t gconv_adapter_infer_workspace_bytes(const gconv_graph_t* graph, int32_t hidden, int32_t rank) {
  const int fp = padded_rank(rank);
  if (!graph || fp == 0) return 0;
  const size_t base = gconv_adapter_fwd_workspace_bytes(graph, hidden, rank);
  if (tile_path_ok(graph, hidden, fp)) return base;
  Carver c(nullptr);
  c.off = base;
  const size_t rows = static_cast<size_t>(graph->fwd.num_nodes) * fp;
  for (int i = 0; i < 3; ++i) c.take<float>(rows);   // (the third one: SiLU's pre-activation)
  return c.off;
}

extern "C" gconv_status_t gconv_adapter_infer_f32(const gconv_graph_t* gr, const gconv_adapter_params_t* p,
                                                  const float* x, float* out, void* workspace,
                                                  size_t workspace_bytes, gconv_stream_t stream_) {
  return adapter_forward(gr, p, x, out, nullptr, nullptr, nullptr, false, workspace, workspace_bytes, stream_);
}

static_assert(sizeof(FinalizeArgs) <= sizeof(gconv_pending_finalize_t),
              "gconv_pending_finalize_t must be able to hold a FinalizeArgs");

// The general-path backward on the rows [lo, hi), `stages` = bit set of
//   1: B1   2: B2 (+ bias-partial pre-reduction)   4: B3, B4, B5
// x / grad_out / grad_x hold the rows [lo, hi) only; saved_*, dp, dh1 are full-height arrays in
// global row coordinates; `ws` was carved for hi - lo rows. Single device: lo = 0, hi = n, 7.
static gconv_status_t bwd_stages(const gconv_graph_t* gr, const gconv_adapter_params_t* p, int lo,
                                 int hi, int stages, const float* x, const float* grad_out,
                                 const float* saved_h, const float* saved_h1, const float* saved_p,
                                 float* dp, float* dh1, float* grad_x, float* grad_w_down,
                                 float* grad_b_down, float* grad_w_up, float* grad_b_up,
                                 float* grad_scalar, const BwdWs& ws, FinalizeArgs* pending,
                                 hipStream_t stream) {
  const gconv_csr_t* graph_t = &gr->bwd;
  const float* dis = gr->dis;
  const int n = hi - lo, d = p->hidden, r = p->rank, fp = padded_rank(r);
  const size_t shift = static_cast<size_t>(lo) * fp;
  gconv_status_t st = GCONV_OK;
  if (n > 0 && (stages & 1)) {
    RowsAArgs b1{};
    b1.X = grad_out; b1.W = p->w_up; b1.row_scale = dis ? dis + lo : nullptr; b1.scalar = p->scalar;
    b1.P = grad_scalar ? saved_p + shift : nullptr;
    b1.ds_partial = grad_scalar ? ws.part_ds1 : nullptr;
    b1.out = dp + shift; b1.n = n; b1.d = d; b1.r = r; b1.fp = fp; b1.mode = ROWSA_DP;
    st = launch_rowsA(b1, stream);
    if (st != GCONV_OK) return st;
  }
  if (n > 0 && (stages & 2)) {
    SpmmArgs b2{};
    b2.dis = dis; b2.H = dp; b2.Y = dh1;
    b2.saved = (p->act == GCONV_ACT_SILU) ? saved_h1 : saved_h;
    b2.act = p->act; b2.rank = r; b2.seg_partial = ws.seg; b2.bias_partial = ws.part_bd;
    b2.row_lo = lo; b2.row_hi = hi;
    st = launch_spmm(graph_t, b2, fp, EPI_BWD1, stream);
    if (st != GCONV_OK) return st;
    if (ws.n_bd > 2 * kReduceBlocks) {  // tall partial array: parallel stage-1 column sums
      st = launch_reduce_rows(ws.part_bd, ws.n_bd, fp, ws.part_bd2, stream);
      if (st != GCONV_OK) return st;
    }
  }
  if (!(stages & 4)) return GCONV_OK;
  if (n > 0) {
    SpmmArgs b3{};
    b3.dis = dis; b3.H = dh1; b3.Y = dp; b3.rank = r; b3.seg_partial = ws.seg;
    b3.row_lo = lo; b3.row_hi = hi;
    BwdMainArgs b4{};
    b4.G = grad_out; b4.X = x; b4.T = dp + shift; b4.P = saved_p + shift; b4.W_down = p->w_down;
    b4.scalar = p->scalar; b4.b_up = grad_scalar ? p->b_up : nullptr; b4.dX = grad_x;
    b4.part_wd = ws.part_wd; b4.part_wu = ws.part_wu; b4.part_cs = ws.part_cs;
    b4.part_dsx = ws.part_dsx;
    b4.n = n; b4.d = d; b4.r = r; b4.fp = fp; b4.skip = p->skip;
    if (lo == 0 && hi == graph_t->num_nodes && fused_gather_enabled(graph_t, d)) {
      // B3 + B4 in one launch (fused_gather.hip); only the long rows' segments run in front
      if (graph_t->n_seg > 0) {
        b3.seg_only = 1;
        st = launch_spmm(graph_t, b3, fp, EPI_STORE, stream);
        if (st != GCONV_OK) return st;
      }
      BwdGatherArgs f{};
      f.rowptr = graph_t->rowptr; f.col = graph_t->col; f.dis = dis; f.H = dh1; f.T = dp; f.main = b4;
      st = launch_bwd_gather(f, stream);
    } else {
      st = launch_spmm(graph_t, b3, fp, EPI_STORE, stream);
      if (st != GCONV_OK) return st;
      st = launch_bwd_main(b4, stream);
    }
    if (st != GCONV_OK) return st;
  }

  FinalizeArgs f{};
  f.part_wd = ws.part_wd; f.part_wu = ws.part_wu; f.part_cs = ws.part_cs;
  f.nparts = n > 0 ? ws.geom.nparts : 0;
  f.part_dsx = ws.part_dsx; f.n_dsx = n > 0 ? ws.geom.nparts * ws.geom.nchunks * ws.geom.nslots : 0;
  f.part_ds1 = ws.part_ds1; f.n_ds1 = n > 0 ? ws.n_ds1 : 0;
  const bool bd_staged = n > 0 && ws.n_bd > 2 * kReduceBlocks;
  f.part_bd = bd_staged ? ws.part_bd2 : ws.part_bd;
  f.n_bd = n > 0 ? (bd_staged ? kReduceBlocks : ws.n_bd) : 0;
  f.scalar = p->scalar; f.b_up = p->b_up; f.cs_tmp = ws.cs_tmp;
  f.g_wd = grad_w_down; f.g_bd = grad_b_down; f.g_wu = grad_w_up; f.g_bu = grad_b_up;
  f.g_s = grad_scalar;
  f.d = d; f.r = r; f.fp = fp; f.skip = p->skip;
  if (pending) { *pending = f; return GCONV_OK; }
  return launch_finalize(f, stream);
}

// The whole backward; the last reduction is either launched (pending == nullptr) or described
// in *pending for gconv_adapter_bwd_finalize_f32.
static gconv_status_t adapter_bwd(const gconv_graph_t* gr, const gconv_adapter_params_t* p,
                                  const float* x, const float* grad_out, const float* saved_h,
                                  const float* saved_h1, const float* saved_p, float* grad_x,
                                  float* grad_w_down, float* grad_b_down, float* grad_w_up,
                                  float* grad_b_up, float* grad_scalar, void* workspace,
                                  size_t workspace_bytes, FinalizeArgs* pending,
                                  gconv_stream_t stream_) {
  if (!gr) return GCONV_E_INVALID_ARG;
  const gconv_csr_t* graph_t = &gr->bwd;
  const float* dis = gr->dis;
  gconv_status_t st = check_common(graph_t, p);
  if (st != GCONV_OK) return st;
  const int n = graph_t->num_nodes, d = p->hidden, r = p->rank, fp = padded_rank(r);
  if (!grad_w_down || !grad_w_up || !workspace) return GCONV_E_INVALID_ARG;
  if (n > 0 && (!x || !grad_out || !saved_h || !saved_p)) return GCONV_E_INVALID_ARG;
  if (p->act == GCONV_ACT_SILU && n > 0 && !saved_h1) return GCONV_E_INVALID_ARG;
  if (!aligned16(x) || !aligned16(grad_out) || !aligned16(saved_h) || !aligned16(saved_p) ||
      (saved_h1 && !aligned16(saved_h1)) || (grad_x && !aligned16(grad_x)) ||
      !aligned16(workspace) || !aligned16(p->w_down) || (p->b_up && !aligned16(p->b_up)))
    return GCONV_E_ALIGNMENT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (n > 0 && tile_path_ok(gr, d, fp)) {  // block-diagonal batch: fused launch + finalize
    const TileBwdWs tw = carve_bwd_tile(workspace, gr->n_tiles, d, fp);
    if (workspace_bytes < tw.bytes) return GCONV_E_WORKSPACE;
    TileBwdArgs t{};
    t.tiles = gr->tiles_bwd.tiles; t.lcol = gr->tiles_bwd.lcol; t.halo = gr->tiles_bwd.halo;
    t.rowptr = graph_t->rowptr; t.dis = dis; t.n = n;
    t.x = x; t.g = grad_out; t.saved_h = saved_h; t.saved_h1 = saved_h1; t.saved_p = saved_p;
    t.w_down = p->w_down; t.w_up = p->w_up; t.b_up = p->b_up; t.scalar = p->scalar;
    t.dx = grad_x;
    t.part_wd = tw.part_wd; t.part_wu = tw.part_wu; t.part_cs = tw.part_cs;
    t.part_bd = tw.part_bd; t.part_ds = tw.part_ds;
    t.d = d; t.r = r; t.fp = fp; t.act = p->act; t.skip = p->skip;
    t.stagger = (p->walk & 1) ^ 1;   // the backward takes the half the forward did not (gconv_adapter_params_t::walk)
    t.want_ds = grad_scalar != nullptr;
#ifdef GCONV_TILE_STAMPS
    t.stamps = reinterpret_cast<long long*>(static_cast<char*>(workspace) + tw.bytes);
#endif
    st = launch_tile_bwd(t, gr->n_tiles, gr->tile_nodes_max, gr->tile_edges_max, stream);
    if (st != GCONV_OK) return st;
    FinalizeArgs f{};   // ONE launch: every partial (d(scalar) included) is complete per workgroup
    f.part_wd = tw.part_wd; f.part_wu = tw.part_wu; f.part_cs = tw.part_cs; f.nparts = tw.nparts;
    f.part_ds1 = tw.part_ds; f.n_ds1 = tw.nparts; f.ds_complete = 1;
    f.part_bd = tw.part_bd; f.n_bd = tw.nparts;
    f.scalar = p->scalar; f.b_up = p->b_up; f.cs_tmp = tw.cs_tmp;
    f.g_wd = grad_w_down; f.g_bd = grad_b_down; f.g_wu = grad_w_up; f.g_bu = grad_b_up;
    f.g_s = grad_scalar; f.d = d; f.r = r; f.fp = fp; f.skip = p->skip;
    if (pending) { *pending = f; return GCONV_OK; }
    return launch_finalize(f, stream);
  }
  const BwdWs ws = carve_bwd(workspace, graph_t, d, fp, n, false);
  if (workspace_bytes < ws.bytes) return GCONV_E_WORKSPACE;
  return bwd_stages(gr, p, 0, n, 7, x, grad_out, saved_h, saved_h1, saved_p, ws.dp, ws.dh1, grad_x,
                    grad_w_down, grad_b_down, grad_w_up, grad_b_up, grad_scalar, ws, pending, stream);
}

extern "C" gconv_status_t gconv_adapter_bwd_f32(const gconv_graph_t* gr,
                                                const gconv_adapter_params_t* p, const float* x,
                                                const float* grad_out, const float* saved_h,
                                                const float* saved_h1, const float* saved_p,
                                                float* grad_x, float* grad_w_down,
                                                float* grad_b_down, float* grad_w_up,
                                                float* grad_b_up, float* grad_scalar,
                                                void* workspace, size_t workspace_bytes,
                                                gconv_stream_t stream) {
  return adapter_bwd(gr, p, x, grad_out, saved_h, saved_h1, saved_p, grad_x, grad_w_down, grad_b_down,
                     grad_w_up, grad_b_up, grad_scalar, workspace, workspace_bytes, nullptr, stream);
}

extern "C" gconv_status_t gconv_adapter_bwd_partial_f32(
    const gconv_graph_t* gr, const gconv_adapter_params_t* p, const float* x, const float* grad_out,
    const float* saved_h, const float* saved_h1, const float* saved_p, float* grad_x,
    float* grad_w_down, float* grad_b_down, float* grad_w_up, float* grad_b_up, float* grad_scalar,
    void* workspace, size_t workspace_bytes, gconv_pending_finalize_t* pending,
    gconv_stream_t stream) {
  if (!pending) return GCONV_E_INVALID_ARG;
  FinalizeArgs f{};
  const gconv_status_t st =
      adapter_bwd(gr, p, x, grad_out, saved_h, saved_h1, saved_p, grad_x, grad_w_down, grad_b_down,
                  grad_w_up, grad_b_up, grad_scalar, workspace, workspace_bytes, &f, stream);
  if (st != GCONV_OK) return st;
  memset(pending, 0, sizeof(*pending));
  memcpy(pending, &f, sizeof(f));
  return GCONV_OK;
}

extern "C" gconv_status_t gconv_adapter_bwd_finalize_f32(const gconv_pending_finalize_t* pending,
                                                         int32_t count, gconv_stream_t stream) {
  if (count < 0 || (count > 0 && !pending)) return GCONV_E_INVALID_ARG;
  if (count == 0) return GCONV_OK;
  static_assert(sizeof(gconv_pending_finalize_t) % alignof(FinalizeArgs) == 0, "array stride");
  FinalizeArgs f[GCONV_FINALIZE_BATCH];
  for (int base = 0; base < count; base += GCONV_FINALIZE_BATCH) {
    const int m = count - base < GCONV_FINALIZE_BATCH ? count - base : GCONV_FINALIZE_BATCH;
    for (int j = 0; j < m; ++j) memcpy(&f[j], &pending[base + j], sizeof(FinalizeArgs));
    const gconv_status_t st = launch_finalize_batched(f, m, static_cast<hipStream_t>(stream));
    if (st != GCONV_OK) return st;
  }
  return GCONV_OK;
}

// ---- one big graph over several devices: row slabs (include/gconv_adapter.h) -------
static gconv_status_t check_slab(const gconv_graph_t* gr, const gconv_csr_t* g,
                                 const gconv_adapter_params_t* p, int lo, int hi, int stage) {
  if (!gr) return GCONV_E_INVALID_ARG;
  const gconv_status_t st = check_common(g, p);
  if (st != GCONV_OK) return st;
  if (lo < 0 || hi < lo || hi > g->num_nodes || stage < 0 || stage > 2) return GCONV_E_INVALID_ARG;
  return GCONV_OK;
}

extern "C" size_t gconv_adapter_slab_fwd_workspace_bytes(const gconv_graph_t* graph, int32_t rank) {
  const int fp = padded_rank(rank);
  if (!graph || fp == 0) return 0;
  return align_up(static_cast<size_t>(graph->fwd.n_seg) * fp * sizeof(float) + 16, 256);
}

extern "C" size_t gconv_adapter_slab_bwd_workspace_bytes(const gconv_graph_t* graph, int32_t row_lo,
                                                         int32_t row_hi, int32_t hidden,
                                                         int32_t rank) {
  const int fp = padded_rank(rank);
  if (!graph || fp == 0 || hidden <= 0 || row_lo < 0 || row_hi < row_lo) return 0;
  return carve_bwd(nullptr, &graph->bwd, hidden, fp, row_hi - row_lo, true).bytes;
}

extern "C" gconv_status_t gconv_adapter_slab_fwd_f32(
    const gconv_graph_t* gr, const gconv_adapter_params_t* p, int32_t row_lo, int32_t row_hi,
    int32_t stage, const float* x, float* out, float* h0, float* saved_h, float* saved_h1,
    float* saved_p, void* workspace, size_t workspace_bytes, gconv_stream_t stream) {
  gconv_status_t st = check_slab(gr, gr ? &gr->fwd : nullptr, p, row_lo, row_hi, stage);
  if (st != GCONV_OK) return st;
  if (row_hi == row_lo) return GCONV_OK;   // a device without rows only takes part in the exchanges
  if (!x || !out || !h0 || !saved_h || !saved_p) return GCONV_E_INVALID_ARG;
  if (p->act == GCONV_ACT_SILU && !saved_h1) return GCONV_E_INVALID_ARG;
  if (!aligned16(x) || !aligned16(out) || !aligned16(h0) || !aligned16(saved_h) ||
      !aligned16(saved_p) || (saved_h1 && !aligned16(saved_h1)) ||
      (workspace && !aligned16(workspace)) || (p->b_up && !aligned16(p->b_up)) ||
      !aligned16(p->w_down))
    return GCONV_E_ALIGNMENT;
  if (gr->fwd.n_seg > 0 &&
      (!workspace || workspace_bytes < gconv_adapter_slab_fwd_workspace_bytes(gr, p->rank)))
    return GCONV_E_WORKSPACE;
  return fwd_stages(gr, p, row_lo, row_hi, 1 << stage, x, out, h0, saved_h, saved_h1, saved_p,
                    static_cast<float*>(workspace), static_cast<hipStream_t>(stream));
}

extern "C" gconv_status_t gconv_adapter_slab_bwd_f32(
    const gconv_graph_t* gr, const gconv_adapter_params_t* p, int32_t row_lo, int32_t row_hi,
    int32_t stage, const float* x, const float* grad_out, const float* saved_h,
    const float* saved_h1, const float* saved_p, float* dp, float* dh1, float* grad_x,
    float* grad_w_down, float* grad_b_down, float* grad_w_up, float* grad_b_up, float* grad_scalar,
    void* workspace, size_t workspace_bytes, gconv_stream_t stream) {
  gconv_status_t st = check_slab(gr, gr ? &gr->bwd : nullptr, p, row_lo, row_hi, stage);
  if (st != GCONV_OK) return st;
  const int rows = row_hi - row_lo;
  if (!grad_w_down || !grad_w_up || !workspace) return GCONV_E_INVALID_ARG;
  if (rows > 0 && (!x || !grad_out || !saved_h || !saved_p || !dp || !dh1)) return GCONV_E_INVALID_ARG;
  if (p->act == GCONV_ACT_SILU && rows > 0 && !saved_h1) return GCONV_E_INVALID_ARG;
  if (!aligned16(x) || !aligned16(grad_out) || !aligned16(saved_h) || !aligned16(saved_p) ||
      (saved_h1 && !aligned16(saved_h1)) || (grad_x && !aligned16(grad_x)) || !aligned16(dp) ||
      !aligned16(dh1) || !aligned16(workspace) || !aligned16(p->w_down) || (p->b_up && !aligned16(p->b_up)))
    return GCONV_E_ALIGNMENT;
  const BwdWs ws = carve_bwd(workspace, &gr->bwd, p->hidden, padded_rank(p->rank), rows, true);
  if (workspace_bytes < ws.bytes) return GCONV_E_WORKSPACE;
  return bwd_stages(gr, p, row_lo, row_hi, 1 << stage, x, grad_out, saved_h, saved_h1, saved_p, dp,
                    dh1, grad_x, grad_w_down, grad_b_down, grad_w_up, grad_b_up, grad_scalar, ws,
                    nullptr, static_cast<hipStream_t>(stream));
}
