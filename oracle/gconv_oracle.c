/* Plain-C restatement of the GConv-Adapter hot path in PyG's op order, fp64 — TEST
 * INFRASTRUCTURE ONLY (see oracle/__init__.py): compiled by __graft_entry__.build() into
 * oracle/_build/libgconv_oracle.so, loaded by tests/ only. PARITY UNPINNED (the reference has
 * no golden vectors and torch-geometric is not installable here): this file restates the
 * published semantics of torch_geometric.nn.GCNConv (2.5.3 / 1.7.2) at the reference's call
 * sites and is cross-checked against the two torch derivations (oracle/pyg_restated.py,
 * oracle/dense_ref.py) and the committed golden vectors.
 *
 * Third derivation, independent of torch: scalar loops over the edge list, and a backward
 * written by hand as the reverse of each forward op (no autograd).
 *
 * Reference anchors (/root/reference):
 *   composition / flags / residual / scalar order   src/finetune/gconv_adapter.py:80-108
 *   the two convs and the `normalize` kwarg          src/finetune/gconv_adapter.py:40-41
 * Restated PyG semantics:
 *   edge_index[0] = source j, edge_index[1] = target i, sum on the target; duplicates count;
 *   normalize=False: no loops added, unit weights;
 *   normalize=True : existing self-loops dropped, ONE loop of weight 1 per node, in-degree on the
 *                    target of the looped graph, dis = deg^-1/2 (inf -> 0), w = dis[src] * dis[dst];
 *   bias-free linear FIRST (conv_up therefore propagates at width d, as PyG does), then
 *   propagate, then + bias; out += x before the scalar.
 * normalization='none' only (the LayerNorm / BatchNorm tails stay with the torch oracle).     */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };

typedef struct {
  int64_t m;        /* edges of the operator (after gcn_norm when normalised) */
  int64_t *src, *dst;
  double* w;        /* NULL: unit weights */
} op_t;

/* gcn_norm (PyG gcn_conv.py :: gcn_norm, add_remaining_self_loops + symmetric weights) */
static int build_op(int64_t n, int64_t e, const int64_t* src, const int64_t* dst, int normalize, op_t* op) {
  if (!normalize) {
    op->m = e;
    op->src = (int64_t*)malloc(sizeof(int64_t) * (size_t)(e > 0 ? e : 1));
    op->dst = (int64_t*)malloc(sizeof(int64_t) * (size_t)(e > 0 ? e : 1));
    op->w = NULL;
    if (!op->src || !op->dst) return -1;
    memcpy(op->src, src, sizeof(int64_t) * (size_t)e);
    memcpy(op->dst, dst, sizeof(int64_t) * (size_t)e);
    return 0;
  }
  int64_t m = 0;
  for (int64_t k = 0; k < e; ++k) m += src[k] != dst[k];
  m += n;
  op->m = m;
  op->src = (int64_t*)malloc(sizeof(int64_t) * (size_t)(m > 0 ? m : 1));
  op->dst = (int64_t*)malloc(sizeof(int64_t) * (size_t)(m > 0 ? m : 1));
  op->w = (double*)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
  double* deg = (double*)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
  if (!op->src || !op->dst || !op->w || !deg) return -1;
  int64_t q = 0;
  for (int64_t k = 0; k < e; ++k)
    if (src[k] != dst[k]) { op->src[q] = src[k]; op->dst[q] = dst[k]; ++q; }
  for (int64_t i = 0; i < n; ++i) { op->src[q] = i; op->dst[q] = i; ++q; }
  for (int64_t k = 0; k < m; ++k) deg[op->dst[k]] += 1.0;          /* in-degree on the target */
  for (int64_t i = 0; i < n; ++i) deg[i] = deg[i] > 0.0 ? 1.0 / sqrt(deg[i]) : 0.0;   /* inf -> 0 */
  for (int64_t k = 0; k < m; ++k) op->w[k] = deg[op->src[k]] * deg[op->dst[k]];
  free(deg);
  return 0;
}
static void free_op(op_t* op) { free(op->src); free(op->dst); free(op->w); }

/* out[n, c] = in[n, k] * W[c, k]^T   (bias-free nn.Linear) */
static void linear(int64_t n, int k, int c, const double* in, const double* W, double* out) {
  for (int64_t i = 0; i < n; ++i)
    for (int a = 0; a < c; ++a) {
      double s = 0.0;
      for (int b = 0; b < k; ++b) s += in[i * k + b] * W[(int64_t)a * k + b];
      out[i * c + a] = s;
    }
}
/* MessagePassing.propagate, aggr='add', flow source -> target: out[dst] += w * h[src] */
static void propagate(const op_t* op, int64_t n, int c, const double* h, double* out) {
  memset(out, 0, sizeof(double) * (size_t)(n * c));
  for (int64_t k = 0; k < op->m; ++k) {
    const double w = op->w ? op->w[k] : 1.0;
    const double* hs = h + op->src[k] * c;
    double* od = out + op->dst[k] * c;
    for (int a = 0; a < c; ++a) od[a] += w * hs[a];
  }
}
/* its adjoint: gh[src] += w * gout[dst] */
static void propagate_t(const op_t* op, int64_t n, int c, const double* gout, double* gh) {
  memset(gh, 0, sizeof(double) * (size_t)(n * c));
  for (int64_t k = 0; k < op->m; ++k) {
    const double w = op->w ? op->w[k] : 1.0;
    const double* gd = gout + op->dst[k] * c;
    double* gs = gh + op->src[k] * c;
    for (int a = 0; a < c; ++a) gs[a] += w * gd[a];
  }
}
static double act_f(int act, double h) {
  if (act == ACT_RELU) return h > 0.0 ? h : 0.0;
  if (act == ACT_SILU) return h / (1.0 + exp(-h));
  return h;
}
static double act_g(int act, double h) {
  if (act == ACT_RELU) return h > 0.0 ? 1.0 : 0.0;
  if (act == ACT_SILU) { const double s = 1.0 / (1.0 + exp(-h)); return s * (1.0 + h * (1.0 - s)); }
  return 1.0;
}

/* Forward and (when grad_out != NULL) backward of one adapter call.
 *   x [n, d]; w_down [r, d]; b_down [r] or NULL; w_up [d, r]; b_up [d] or NULL; scalar: pointer
 *   to one double or NULL (learnable_scalar=False); out [n, d].
 *   gradients (all nullable individually): g_x [n, d], g_wd [r, d], g_bd [r], g_wu [d, r],
 *   g_bu [d], g_s [1].  Returns 0, -1 on allocation failure, -2 on an edge endpoint out of range. */
int gconv_oracle_adapter_f64(int64_t n, int64_t e, const int64_t* src, const int64_t* dst, int d, int r,
                             const double* x, const double* w_down, const double* b_down,
                             const double* w_up, const double* b_up, const double* scalar, int act,
                             int skip, int normalize, const double* grad_out, double* out, double* g_x,
                             double* g_wd, double* g_bd, double* g_wu, double* g_bu, double* g_s) {
  for (int64_t k = 0; k < e; ++k)
    if (src[k] < 0 || src[k] >= n || dst[k] < 0 || dst[k] >= n) return -2;
  op_t op;
  if (build_op(n, e, src, dst, normalize, &op)) return -1;
  const size_t nr = (size_t)(n * r > 0 ? n * r : 1), nd = (size_t)(n * d > 0 ? n * d : 1);
  double* l1 = (double*)malloc(sizeof(double) * nr);   /* x Wd^T            */
  double* h1 = (double*)malloc(sizeof(double) * nr);   /* S l1 + bd         */
  double* h2 = (double*)malloc(sizeof(double) * nr);   /* act(h1)           */
  double* l2 = (double*)malloc(sizeof(double) * nd);   /* h2 Wu^T  [n, d]   */
  double* h3 = (double*)malloc(sizeof(double) * nd);   /* S l2 + bu (+ x)   */
  if (!l1 || !h1 || !h2 || !l2 || !h3) return -1;
  const double s = scalar ? scalar[0] : 1.0;

  linear(n, d, r, x, w_down, l1);                       /* conv_down: lin -> propagate -> + bias */
  propagate(&op, n, r, l1, h1);
  for (int64_t i = 0; i < n; ++i)
    for (int a = 0; a < r; ++a) {
      if (b_down) h1[i * r + a] += b_down[a];
      h2[i * r + a] = act_f(act, h1[i * r + a]);
    }
  linear(n, r, d, h2, w_up, l2);                        /* conv_up: propagates at width d */
  propagate(&op, n, d, l2, h3);
  for (int64_t i = 0; i < n; ++i)
    for (int j = 0; j < d; ++j) {
      double v = h3[i * d + j] + (b_up ? b_up[j] : 0.0);
      if (skip) v += x[i * d + j];                      /* gconv_adapter.py:94-95 */
      h3[i * d + j] = v;                                /* the un-scaled output */
      out[i * d + j] = v * s;                           /* :105-106 */
    }

  if (grad_out) {
    double* g3 = (double*)malloc(sizeof(double) * nd);  /* dL/d(un-scaled out) */
    double* gl2 = (double*)malloc(sizeof(double) * nd);
    double* gh = (double*)malloc(sizeof(double) * nr);
    double* gl1 = (double*)malloc(sizeof(double) * nr);
    if (!g3 || !gl2 || !gh || !gl1) return -1;
    double ds = 0.0;
    for (size_t q = 0; q < (size_t)(n * d); ++q) { ds += grad_out[q] * h3[q]; g3[q] = grad_out[q] * s; }
    if (g_s) g_s[0] = ds;
    if (g_bu) for (int j = 0; j < d; ++j) { double c = 0.0; for (int64_t i = 0; i < n; ++i) c += g3[i * d + j]; g_bu[j] = c; }
    propagate_t(&op, n, d, g3, gl2);                    /* through conv_up's propagate */
    if (g_wu)                                           /* l2 = h2 Wu^T  ->  dWu = gl2^T h2 */
      for (int j = 0; j < d; ++j)
        for (int a = 0; a < r; ++a) { double c = 0.0; for (int64_t i = 0; i < n; ++i) c += gl2[i * d + j] * h2[i * r + a]; g_wu[(int64_t)j * r + a] = c; }
    for (int64_t i = 0; i < n; ++i)                     /* dh2 = gl2 Wu ; dh1 = dh2 * act'(h1) */
      for (int a = 0; a < r; ++a) {
        double c = 0.0;
        for (int j = 0; j < d; ++j) c += gl2[i * d + j] * w_up[(int64_t)j * r + a];
        gh[i * r + a] = c * act_g(act, h1[i * r + a]);
      }
    if (g_bd) for (int a = 0; a < r; ++a) { double c = 0.0; for (int64_t i = 0; i < n; ++i) c += gh[i * r + a]; g_bd[a] = c; }
    propagate_t(&op, n, r, gh, gl1);                    /* through conv_down's propagate */
    if (g_wd)
      for (int a = 0; a < r; ++a)
        for (int j = 0; j < d; ++j) { double c = 0.0; for (int64_t i = 0; i < n; ++i) c += gl1[i * r + a] * x[i * d + j]; g_wd[(int64_t)a * d + j] = c; }
    if (g_x)
      for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < d; ++j) {
          double c = skip ? g3[i * d + j] : 0.0;
          for (int a = 0; a < r; ++a) c += gl1[i * r + a] * w_down[(int64_t)a * d + j];
          g_x[i * d + j] = c;
        }
    free(g3); free(gl2); free(gh); free(gl1);
  }
  free(l1); free(h1); free(h2); free(l2); free(h3);
  free_op(&op);
  return 0;
}
