// Autograd binding of the adapter in C++ and the native graph-plan prefetch worker (host code
// only, no device code; the HIP runtime API is used for one event per prefetched plan).
//
// The product boundary stays the plain C ABI of include/gconv_adapter.h; this file is the
// thinnest possible bridge from torch's autograd engine to gconv_adapter_fwd_f32 /
// gconv_adapter_bwd_f32, so that a GConvAdapter call costs a few microseconds of host time
// instead of ~30 (forward) + ~60 (backward) in Python — molecule batches are launch-bound.
// It mirrors functional.GConvAdapterFunction (the pure-Python binding, kept as the fallback
// when this module is not built) argument for argument.
//
// Replaces on the reference side: the torch autograd graph that PyG's GCNConv builds for
// `conv_up(act_fn(conv_down(x, ei)), ei)` (/root/reference/src/finetune/gconv_adapter.py:92).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/engine.h>
#include <torch/extension.h>

#include <algorithm>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>

#include "../../include/gconv_adapter.h"

namespace {

using fwd_fn_t = gconv_status_t (*)(const gconv_graph_t*, const gconv_adapter_params_t*,
                                    const float*, float*, float*, float*, float*, void*, size_t,
                                    gconv_stream_t);
using bwd_fn_t = gconv_status_t (*)(const gconv_graph_t*, const gconv_adapter_params_t*,
                                    const float*, const float*, const float*, const float*,
                                    const float*, float*, float*, float*, float*, float*, float*,
                                    void*, size_t, gconv_stream_t);
using str_fn_t = const char* (*)(gconv_status_t);
using bytes_fn_t = size_t (*)(int64_t, int32_t);
using build_fn_t = gconv_status_t (*)(const int64_t*, int64_t, int32_t, int32_t, void*, size_t,
                                      gconv_graph_t*, gconv_stream_t);
using build_ex_fn_t = gconv_status_t (*)(const int64_t*, int64_t, int32_t, int32_t, uint32_t, void*, size_t,
                                         gconv_graph_t*, gconv_stream_t);
using bwd_partial_fn_t = gconv_status_t (*)(const gconv_graph_t*, const gconv_adapter_params_t*,
                                            const float*, const float*, const float*, const float*,
                                            const float*, float*, float*, float*, float*, float*, float*,
                                            void*, size_t, gconv_pending_finalize_t*, gconv_stream_t);
using bwd_finalize_fn_t = gconv_status_t (*)(const gconv_pending_finalize_t*, int32_t, gconv_stream_t);
using ln_fwd_fn_t = gconv_status_t (*)(const float*, const float*, const float*, const float*, float*, float*,
                                       int32_t, int32_t, float, gconv_stream_t);
using ln_bwd_fn_t = gconv_status_t (*)(const float*, const float*, const float*, const float*, const float*,
                                       const float*, float*, float*, float*, float*, int32_t, int32_t, void*,
                                       size_t, gconv_stream_t);
using ln_bytes_fn_t = size_t (*)(int32_t, int32_t);
using rank_fn_t = int32_t (*)(int32_t);
using ws_fn_t = size_t (*)(const gconv_graph_t*, int32_t, int32_t);
using infer_fn_t = gconv_status_t (*)(const gconv_graph_t*, const gconv_adapter_params_t*, const float*, float*,
                                      void*, size_t, gconv_stream_t);
using spmm_ws_fn_t = size_t (*)(const gconv_csr_t*, int32_t);
using spmm_fn_t = gconv_status_t (*)(const gconv_csr_t*, const float*, const float*, float*, int32_t, const float*,
                                     int32_t, uint32_t, void*, size_t, gconv_stream_t);

using bn_bytes_fn_t = size_t (*)(int32_t, int32_t);
using bn_fwd_fn_t = gconv_status_t (*)(const float*, const float*, const float*, float*, float*, int64_t*, float, float,
                                       int32_t, float*, float*, float*, int32_t, int32_t, void*, size_t,
                                       gconv_stream_t);
using bn_bwd_fn_t = gconv_status_t (*)(const float*, const float*, const float*, const float*, const float*,
                                       const float*, float*, float*, float*, int32_t, int32_t, void*, size_t,
                                       gconv_stream_t);
using pair_bytes_fn_t = size_t (*)(int32_t, int32_t);
using split_rows_fn_t = gconv_status_t (*)(const float*, int32_t, int32_t, void*, gconv_stream_t);
using pair_gemm_fn_t = gconv_status_t (*)(const void*, const void*, const float*, const void*, float*, void*, int32_t,
                                          int32_t, int32_t, int32_t, gconv_stream_t);
using gin_agg_fn_t = gconv_status_t (*)(const gconv_csr_t*, const float*, const int32_t*, const float*, const float*,
                                        float*, int32_t, gconv_stream_t);
using gin_agg_pair_fn_t = gconv_status_t (*)(const gconv_csr_t*, const float*, const int32_t*, const float*,
                                             const float*, void*, int32_t, gconv_stream_t);

using arm_fn_t = gconv_status_t (*)(void*);

struct Lib {
  void* handle = nullptr;
  arm_fn_t arm = nullptr;
  bn_bytes_fn_t bn_bytes = nullptr;
  bn_fwd_fn_t bn_fwd = nullptr;
  bn_bwd_fn_t bn_bwd = nullptr;
  pair_bytes_fn_t pair_bytes = nullptr;
  split_rows_fn_t split_rows = nullptr;
  pair_gemm_fn_t pair_gemm = nullptr;
  gin_agg_fn_t gin_agg = nullptr;
  gin_agg_pair_fn_t gin_agg_pair = nullptr;
  fwd_fn_t fwd = nullptr;
  infer_fn_t infer = nullptr;
  bwd_fn_t bwd = nullptr;
  str_fn_t str = nullptr;
  bytes_fn_t build_bytes = nullptr;
  build_fn_t build = nullptr;
  build_ex_fn_t build_ex = nullptr;
  bwd_partial_fn_t bwd_partial = nullptr;
  bwd_finalize_fn_t bwd_finalize = nullptr;
  ln_fwd_fn_t ln_fwd = nullptr;
  ln_bwd_fn_t ln_bwd = nullptr;
  ln_bytes_fn_t ln_bytes = nullptr;
  rank_fn_t padded_rank = nullptr;
  ws_fn_t ws_fwd = nullptr, ws_bwd = nullptr;
  spmm_ws_fn_t spmm_ws = nullptr;
  spmm_fn_t spmm = nullptr;
};
Lib g_lib;

void load_library(const std::string& path) {
  if (g_lib.handle) return;
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
  TORCH_CHECK(h != nullptr, "gconv_adapter_amd: cannot dlopen ", path, ": ", dlerror());
  g_lib.fwd = reinterpret_cast<fwd_fn_t>(dlsym(h, "gconv_adapter_fwd_f32"));
  g_lib.bwd = reinterpret_cast<bwd_fn_t>(dlsym(h, "gconv_adapter_bwd_f32"));
  g_lib.infer = reinterpret_cast<infer_fn_t>(dlsym(h, "gconv_adapter_infer_f32"));
  g_lib.arm = reinterpret_cast<arm_fn_t>(dlsym(h, "gconv_arm_completion_event"));
  g_lib.str = reinterpret_cast<str_fn_t>(dlsym(h, "gconv_status_string"));
  g_lib.build_bytes = reinterpret_cast<bytes_fn_t>(dlsym(h, "gconv_graph_build_bytes"));
  g_lib.build = reinterpret_cast<build_fn_t>(dlsym(h, "gconv_graph_build"));
  g_lib.build_ex = reinterpret_cast<build_ex_fn_t>(dlsym(h, "gconv_graph_build_ex"));
  g_lib.bwd_partial = reinterpret_cast<bwd_partial_fn_t>(dlsym(h, "gconv_adapter_bwd_partial_f32"));
  g_lib.bwd_finalize = reinterpret_cast<bwd_finalize_fn_t>(dlsym(h, "gconv_adapter_bwd_finalize_f32"));
  g_lib.ln_fwd = reinterpret_cast<ln_fwd_fn_t>(dlsym(h, "gconv_layer_norm_scale_fwd_f32"));
  g_lib.ln_bwd = reinterpret_cast<ln_bwd_fn_t>(dlsym(h, "gconv_layer_norm_scale_bwd_f32"));
  g_lib.ln_bytes = reinterpret_cast<ln_bytes_fn_t>(dlsym(h, "gconv_layer_norm_scale_bwd_workspace_bytes"));
  g_lib.padded_rank = reinterpret_cast<rank_fn_t>(dlsym(h, "gconv_padded_rank"));
  g_lib.ws_fwd = reinterpret_cast<ws_fn_t>(dlsym(h, "gconv_adapter_fwd_workspace_bytes"));
  g_lib.ws_bwd = reinterpret_cast<ws_fn_t>(dlsym(h, "gconv_adapter_bwd_workspace_bytes"));
  g_lib.spmm_ws = reinterpret_cast<spmm_ws_fn_t>(dlsym(h, "gconv_spmm_workspace_bytes"));
  g_lib.spmm = reinterpret_cast<spmm_fn_t>(dlsym(h, "gconv_spmm_csr_f32"));
  g_lib.bn_bytes = reinterpret_cast<bn_bytes_fn_t>(dlsym(h, "gconv_batch_norm_workspace_bytes"));
  g_lib.bn_fwd = reinterpret_cast<bn_fwd_fn_t>(dlsym(h, "gconv_batch_norm_fwd_tracked_f32"));
  g_lib.bn_bwd = reinterpret_cast<bn_bwd_fn_t>(dlsym(h, "gconv_batch_norm_bwd_f32"));
  g_lib.pair_bytes = reinterpret_cast<pair_bytes_fn_t>(dlsym(h, "gconv_mlp_rows_pair_bytes"));
  g_lib.split_rows = reinterpret_cast<split_rows_fn_t>(dlsym(h, "gconv_mlp_split_rows_f16x2"));
  g_lib.pair_gemm = reinterpret_cast<pair_gemm_fn_t>(dlsym(h, "gconv_mlp_gemm_f16x2"));
  g_lib.gin_agg = reinterpret_cast<gin_agg_fn_t>(dlsym(h, "gconv_gin_aggregate_f32"));
  g_lib.gin_agg_pair = reinterpret_cast<gin_agg_pair_fn_t>(dlsym(h, "gconv_gin_aggregate_f16x2"));
  TORCH_CHECK(g_lib.bn_bytes && g_lib.bn_fwd && g_lib.bn_bwd && g_lib.pair_bytes && g_lib.split_rows &&
                  g_lib.pair_gemm && g_lib.gin_agg && g_lib.gin_agg_pair,
              "gconv_adapter_amd: C ABI symbols of the fine-tune surroundings missing in ", path);
  TORCH_CHECK(g_lib.fwd && g_lib.infer && g_lib.arm && g_lib.bwd && g_lib.str && g_lib.build_bytes && g_lib.build && g_lib.bwd_partial &&
                  g_lib.bwd_finalize && g_lib.ln_fwd && g_lib.ln_bwd && g_lib.ln_bytes && g_lib.padded_rank &&
                  g_lib.ws_fwd && g_lib.ws_bwd && g_lib.spmm_ws && g_lib.spmm && g_lib.build_ex,
              "gconv_adapter_amd: C ABI symbols missing in ", path);
  g_lib.handle = h;
}

const float* fptr(const at::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

// Forward workspace. The fused tile path needs none (gconv_adapter_fwd_workspace_bytes == 0; the
// Python side rounds that up to a 16-byte placeholder because the C ABI wants a non-null pointer):
// one placeholder per device is allocated once and shared by every call — nothing ever touches it
// (an allocation per adapter call is ~1 us of a ~10 us host path). Any workspace a kernel really
// uses is allocated per call: two forwards on different streams must not share scratch. The map is
// leaked on purpose (static destructors run after the caching allocator is gone).
at::Tensor forward_workspace(const at::Tensor& like, int64_t bytes) {
  if (bytes > 16) return at::empty({bytes}, like.options().dtype(at::kByte));
  static std::mutex mu;
  static auto* dummy = new std::unordered_map<int, at::Tensor>();
  std::lock_guard<std::mutex> lk(mu);
  at::Tensor& t = (*dummy)[like.get_device()];
  if (!t.defined()) t = at::empty({16}, like.options().dtype(at::kByte));
  return t;
}
float* fptr_mut(at::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

void check(gconv_status_t st, const char* what) {
  // RuntimeError in Python: what the reference's train / eval loops catch and skip on
  TORCH_CHECK(st == 0, "gconv_adapter_amd.", what, " failed: ", g_lib.str(st), " (status ", st, ")");
}

struct AdapterFn : public torch::autograd::Function<AdapterFn> {
  static at::Tensor forward(torch::autograd::AutogradContext* ctx, at::Tensor x, at::Tensor w_down,
                            c10::optional<at::Tensor> b_down_, at::Tensor w_up,
                            c10::optional<at::Tensor> b_up_, c10::optional<at::Tensor> scalar_,
                            at::Tensor arena, int64_t graph_addr, int64_t fp, int64_t ws_fwd,
                            int64_t ws_bwd, int64_t act, bool skip, int64_t stream, int64_t walk) {
    // optional parameters travel as c10::optional (an undefined Tensor is not a legal
    // autograd input); inside they are plain, possibly undefined, tensors
    at::Tensor b_down = b_down_.value_or(at::Tensor()), b_up = b_up_.value_or(at::Tensor()),
               scalar = scalar_.value_or(at::Tensor());
    TORCH_CHECK(g_lib.handle, "gconv_adapter_amd: library not loaded");
    TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat,
                "gconv_adapter_amd computes in fp32 on a HIP device (no CPU path)");
    if (!x.is_contiguous()) x = x.contiguous();
    if (!w_down.is_contiguous()) w_down = w_down.contiguous();
    if (!w_up.is_contiguous()) w_up = w_up.contiguous();
    const int64_t n = x.size(0), d = x.size(1), r = w_down.size(0);
    gconv_graph_t g;
    std::memcpy(&g, reinterpret_cast<const void*>(graph_addr), sizeof(g));
    TORCH_CHECK(n == g.fwd.num_nodes, "x has ", n, " nodes but the graph plan was built for ",
                g.fwd.num_nodes);
    at::Tensor out = at::empty_like(x);
    at::Tensor saved = at::empty({act == GCONV_ACT_SILU ? 3 : 2, n, fp}, x.options());
    at::Tensor ws = forward_workspace(x, ws_fwd);
    float* sv = saved.data_ptr<float>();
    gconv_adapter_params_t p{fptr(w_down), fptr(b_down), fptr(w_up), fptr(b_up), fptr(scalar),
                             static_cast<int32_t>(d), static_cast<int32_t>(r),
                             static_cast<int32_t>(act), skip ? 1 : 0, static_cast<int32_t>(walk)};
    check(g_lib.fwd(&g, &p, x.data_ptr<float>(), out.data_ptr<float>(), sv,
                    act == GCONV_ACT_SILU ? sv + 2 * n * fp : nullptr, sv + n * fp, ws.data_ptr(),
                    static_cast<size_t>(ws_fwd), reinterpret_cast<void*>(stream)),
          "adapter_fwd");
    ctx->saved_data["walk"] = walk;
    ctx->save_for_backward({x, w_down, b_down, w_up, b_up, scalar, saved, arena});
    ctx->saved_data["g"] = std::string(reinterpret_cast<const char*>(&g), sizeof(g));
    ctx->saved_data["fp"] = fp;
    ctx->saved_data["ws_bwd"] = ws_bwd;
    ctx->saved_data["act"] = act;
    ctx->saved_data["skip"] = skip;
    ctx->saved_data["stream"] = stream;  // the engine replays backward on the forward's stream
    return out;
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::variable_list grads) {
    auto sv = ctx->get_saved_variables();
    at::Tensor x = sv[0], w_down = sv[1], b_down = sv[2], w_up = sv[3], b_up = sv[4],
               scalar = sv[5], saved = sv[6];
    gconv_graph_t g;
    const std::string& gs = ctx->saved_data["g"].toStringRef();
    std::memcpy(&g, gs.data(), sizeof(g));
    const int64_t fp = ctx->saved_data["fp"].toInt(), ws_bwd = ctx->saved_data["ws_bwd"].toInt();
    const int64_t act = ctx->saved_data["act"].toInt(), stream = ctx->saved_data["stream"].toInt();
    const bool skip = ctx->saved_data["skip"].toBool();
    at::Tensor go = grads[0];
    TORCH_CHECK(go.defined() && go.is_cuda() && go.scalar_type() == at::kFloat,
                "gconv_adapter_amd: grad_output must be an fp32 HIP tensor");
    if (!go.is_contiguous()) go = go.contiguous();
    const int64_t n = x.size(0), d = x.size(1), r = w_down.size(0);
    at::Tensor gx = ctx->needs_input_grad(0) ? at::empty_like(x) : at::Tensor();
    at::Tensor gwd = at::empty_like(w_down), gwu = at::empty_like(w_up);
    at::Tensor gbd = b_down.defined() ? at::empty_like(b_down) : at::Tensor();
    at::Tensor gbu = b_up.defined() ? at::empty_like(b_up) : at::Tensor();
    at::Tensor gsc = scalar.defined() ? at::empty_like(scalar) : at::Tensor();
    at::Tensor ws = at::empty({ws_bwd}, x.options().dtype(at::kByte));
    const float* svp = saved.data_ptr<float>();
    gconv_adapter_params_t p{fptr(w_down), fptr(b_down), fptr(w_up), fptr(b_up), fptr(scalar),
                             static_cast<int32_t>(d), static_cast<int32_t>(r),
                             static_cast<int32_t>(act), skip ? 1 : 0,
                             static_cast<int32_t>(ctx->saved_data["walk"].toInt())};
    check(g_lib.bwd(&g, &p, x.data_ptr<float>(), go.data_ptr<float>(), svp,
                    act == GCONV_ACT_SILU ? svp + 2 * n * fp : nullptr, svp + n * fp,
                    fptr_mut(gx), gwd.data_ptr<float>(), fptr_mut(gbd), gwu.data_ptr<float>(),
                    fptr_mut(gbu), fptr_mut(gsc), ws.data_ptr(), static_cast<size_t>(ws_bwd),
                    reinterpret_cast<void*>(stream)),
          "adapter_bwd");
    at::Tensor none;
    return {gx, gwd, gbd, gwu, gbu, gsc, none, none, none, none, none, none, none, none, none};
  }
};

// ---------------------------------------------------------------------------------------
// Direct-deposit variant. The parameters are not autograd inputs of the node: backward writes
// their gradients into `p.grad` itself (assign if undefined, add in place otherwise — what
// AccumulateGrad does), so the engine runs ONE node per adapter instead of six and nothing but
// x / the saved rank-width tensors is packed for backward. Same kernels, same numbers; what
// changes is who stores the result. Not for callers that need `torch.autograd.grad(…, params)`
// or autograd hooks on the parameters (DDP): GConvAdapter.direct_param_grads = False selects
// AdapterFn above. A ParamPack holds the module's parameter tensors (the Parameter objects
// are stable across .to() / load_state_dict / optimizer steps).
// ---------------------------------------------------------------------------------------
struct ParamPack {
  at::Tensor w_down, b_down, w_up, b_up, scalar;
  // The parameters' grad accumulators, kept alive from the forward on: `backward(inputs=params)`
  // identifies a named leaf by its accumulator node (a leaf without one counts as unreachable and
  // cannot be told apart later), and the backward node below decides per parameter by looking that
  // node up in the task's exec_info.
  std::shared_ptr<torch::autograd::Node> acc[5];
  // Gradient sinks (set_grad_sink): a caller-owned buffer per parameter — a view of a flat
  // all-reduce bucket or of FusedAdam's flat gradient buffer. When a parameter's .grad is undefined at
  // backward time, the reduction writes its gradient THERE and .grad becomes that view: no pack / copy
  // before the collective or the optimizer. Undefined = none.
  at::Tensor sink[5];
  const at::Tensor& param(int i) const {
    return i == 0 ? w_down : i == 1 ? b_down : i == 2 ? w_up : i == 3 ? b_up : scalar;
  }
  void ensure_accumulators() {
    const at::Tensor* ps[5] = {&w_down, &b_down, &w_up, &b_up, &scalar};
    for (int i = 0; i < 5; ++i)
      if (!acc[i] && ps[i]->defined() && ps[i]->requires_grad() && !ps[i]->grad_fn())
        acc[i] = torch::autograd::impl::grad_accumulator(*ps[i]);
  }
};
std::mutex g_pack_mu;
std::unordered_map<int64_t, std::shared_ptr<ParamPack>> g_packs;
int64_t g_next_pack = 1;
// parameter -> its gradient sink (guarded by g_pack_mu). The key is the parameter's TensorImpl; the
// entry holds a WEAK reference to it: while the entry lives the address cannot be recycled for
// another tensor, and an entry whose parameter has died is dropped the next time the table is touched.
struct SinkEntry {
  c10::weak_intrusive_ptr<c10::TensorImpl> owner;
  at::Tensor dest;
};
std::unordered_map<c10::TensorImpl*, SinkEntry> g_sinks;
void prune_sinks_locked() {
  for (auto it = g_sinks.begin(); it != g_sinks.end();)
    it = it->second.owner.expired() ? g_sinks.erase(it) : std::next(it);
}

// Registers (or, with no destination, removes) the gradient sink of a parameter: a float tensor of
// the parameter's shape, 16-byte aligned, that the caller keeps alive and does not write during a
// backward pass. Used by dist.FlatGradBucket / FusedAdam.
void set_grad_sink(at::Tensor param, c10::optional<at::Tensor> dest, c10::optional<at::Tensor> only_if) {
  TORCH_CHECK(param.defined(), "set_grad_sink: undefined parameter");
  at::Tensor d = dest.value_or(at::Tensor());
  if (d.defined()) {
    TORCH_CHECK(d.is_cuda() && d.scalar_type() == at::kFloat && d.is_contiguous() && d.sizes() == param.sizes() &&
                    d.device() == param.device() && reinterpret_cast<uintptr_t>(d.data_ptr()) % 16 == 0,
                "set_grad_sink: the sink must be a contiguous, 16-byte aligned fp32 HIP tensor of the parameter's shape");
  }
  std::lock_guard<std::mutex> lk(g_pack_mu);
  prune_sinks_locked();
  if (!d.defined() && only_if.has_value()) {   // remove only the registration of this owner
    const auto it = g_sinks.find(param.unsafeGetTensorImpl());
    if (it == g_sinks.end() || it->second.dest.data_ptr() != only_if->data_ptr()) return;
  }
  g_sinks.erase(param.unsafeGetTensorImpl());
  if (d.defined())
    g_sinks.emplace(param.unsafeGetTensorImpl(),
                    SinkEntry{c10::weak_intrusive_ptr<c10::TensorImpl>(param.getIntrusivePtr()), d});
  for (auto& kv : g_packs)
    for (int i = 0; i < 5; ++i)
      if (kv.second->param(i).defined() && kv.second->param(i).unsafeGetTensorImpl() == param.unsafeGetTensorImpl())
        kv.second->sink[i] = d;
}

int64_t pack_create(at::Tensor w_down, c10::optional<at::Tensor> b_down, at::Tensor w_up,
                    c10::optional<at::Tensor> b_up, c10::optional<at::Tensor> scalar) {
  auto pk = std::make_shared<ParamPack>();
  pk->w_down = std::move(w_down);
  pk->b_down = b_down.value_or(at::Tensor());
  pk->w_up = std::move(w_up);
  pk->b_up = b_up.value_or(at::Tensor());
  pk->scalar = scalar.value_or(at::Tensor());
  std::lock_guard<std::mutex> lk(g_pack_mu);
  prune_sinks_locked();
  for (int i = 0; i < 5; ++i) {
    if (!pk->param(i).defined()) continue;
    const auto it = g_sinks.find(pk->param(i).unsafeGetTensorImpl());
    if (it != g_sinks.end()) pk->sink[i] = it->second.dest;
  }
  g_packs[g_next_pack] = pk;
  return g_next_pack++;
}
void pack_release(int64_t id) {
  std::lock_guard<std::mutex> lk(g_pack_mu);
  g_packs.erase(id);
}
std::shared_ptr<ParamPack> pack_get(int64_t id) {
  std::lock_guard<std::mutex> lk(g_pack_mu);
  auto it = g_packs.find(id);
  TORCH_CHECK(it != g_packs.end(), "gconv_adapter_amd: the adapter module of this graph node is gone");
  return it->second;
}

// The last kernel of every adapter's backward (the fixed-order reduction of the per-workgroup
// partials into dW / db / ds) is deferred to the end of the autograd pass and run for ALL
// adapters of that pass in one launch (gconv_adapter_bwd_finalize_f32): nobody reads a
// parameter gradient before backward() returns, and a model with five adapters otherwise pays
// five small launches and five kernel boundaries on the critical path of its backward.
//
// Lifetime: the deferred state belongs to ONE autograd graph task. It is owned by the final
// callback queued on that task (Engine::queue_callback), the process-wide table below only holds
// weak references keyed by the task id. A backward pass that raises never runs its final callbacks:
// the engine then destroys them with the graph task, the state dies with its callback, the pending
// reductions are dropped and no parameter's .grad has been touched — gradients are stored in
// p.grad only inside flush(), after the reduction that fills them has been launched. The next
// backward pass has a new task id, finds no live state and queues its own callback.
struct PendingGrad {
  at::Tensor param, grad;
};
struct TaskState {
  std::mutex mu;
  std::vector<gconv_pending_finalize_t> pending;
  std::vector<at::Tensor> keep;      // workspaces of the pending entries
  std::vector<PendingGrad> grads;    // reduced by `pending`, stored in p.grad by flush()
  std::vector<c10::TensorImpl*> sunk; // parameters whose sink this pass has already handed out
  int64_t stream = 0;
  int device = 0;                    // the stream's device (the callback may run elsewhere)
};

// Runs `fn` with `device` current (kernel launches need the stream's device), restores after.
template <class F>
void on_device(int device, F&& fn) {
  int cur = device;
  (void)hipGetDevice(&cur);
  if (cur != device) (void)hipSetDevice(device);
  fn();
  if (cur != device) (void)hipSetDevice(cur);
}
std::mutex g_def_mu;
std::unordered_map<int, std::weak_ptr<TaskState>> g_tasks;   // autograd graph task id -> its state

// launches the reductions queued so far and stores their results the way AccumulateGrad would:
// p.grad = g when undefined, p.grad += g otherwise. Caller holds st.mu.
void flush_locked(TaskState& st) {
  if (st.pending.empty()) return;
  std::vector<gconv_pending_finalize_t> pending;
  std::vector<PendingGrad> grads;
  std::vector<at::Tensor> keep;
  pending.swap(st.pending);
  grads.swap(st.grads);
  keep.swap(st.keep);
  st.sunk.clear();
  on_device(st.device, [&] {
    check(g_lib.bwd_finalize(pending.data(), static_cast<int32_t>(pending.size()),
                             reinterpret_cast<void*>(st.stream)),
          "adapter_bwd_finalize");
  });
  at::NoGradGuard no_grad;
  for (auto& a : grads) {
    at::Tensor& cur = a.param.mutable_grad();
    if (!cur.defined()) cur = std::move(a.grad);
    else cur.add_(a.grad);
  }
}

void queue_grad(const at::Tensor& p, at::Tensor g, bool wanted, TaskState& st) {
  if (!g.defined()) return;
  if (!wanted) {
    st.keep.push_back(std::move(g));   // still written by the reduction
    return;
  }
  st.grads.push_back({p, std::move(g)});
}

// state of the graph task that is executing the calling node (created, with its final callback, on
// first use)
std::shared_ptr<TaskState> current_task_state() {
  const int tid = torch::autograd::get_current_graph_task_id();
  std::shared_ptr<TaskState> st;
  {
    std::lock_guard<std::mutex> lk(g_def_mu);
    auto it = g_tasks.find(tid);
    if (it != g_tasks.end()) st = it->second.lock();
    if (st) return st;
    for (auto j = g_tasks.begin(); j != g_tasks.end();)   // tasks that ended (or failed) since
      j = j->second.expired() ? g_tasks.erase(j) : std::next(j);
    st = std::make_shared<TaskState>();
    g_tasks[tid] = st;
  }
  torch::autograd::Engine::get_default_engine().queue_callback([st, tid] {
    {
      std::lock_guard<std::mutex> lk(st->mu);
      flush_locked(*st);
    }
    std::lock_guard<std::mutex> lk(g_def_mu);
    auto it = g_tasks.find(tid);
    if (it != g_tasks.end() && it->second.lock() == st) g_tasks.erase(it);
  });
  return st;
}

// True when standard autograd accumulation has observers that a direct store into p.grad would
// bypass: tensor hooks, post-accumulate-grad hooks, or hooks on a live grad accumulator (what
// DistributedDataParallel's reducer installs). Checked on every call: hooks can be added any time.
bool has_grad_observers(const at::Tensor& p) {
  if (!p.defined()) return false;
  auto* meta = torch::autograd::impl::get_autograd_meta(p);
  if (!meta) return false;
  if (!meta->hooks_.empty() || meta->post_acc_grad_hooks_) return true;
  if (auto acc = meta->grad_accumulator_.lock())
    if (!acc->post_hooks().empty() || !acc->pre_hooks().empty() || !acc->tensor_pre_hooks().empty() ||
        acc->tensor_post_acc_grad_hooks())
      return true;
  return false;
}

struct AdapterFnDirect : public torch::autograd::Function<AdapterFnDirect> {
  static at::Tensor forward(torch::autograd::AutogradContext* ctx, at::Tensor x,
                            c10::optional<at::Tensor> anchor, int64_t pack_id, at::Tensor arena,
                            int64_t graph_addr, int64_t fp, int64_t ws_fwd, int64_t ws_bwd, int64_t act,
                            bool skip, int64_t stream, int64_t walk) {
    // `anchor` is the module's W_down: a real edge to a parameter's grad accumulator keeps the node
    // alive when x needs no gradient AND puts it on the path of backward(inputs=parameters); its
    // gradient slot stays undefined (the accumulator is a no-op), the deposit happens below.
    (void)anchor;
    TORCH_CHECK(g_lib.handle, "gconv_adapter_amd: library not loaded");
    TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat,
                "gconv_adapter_amd computes in fp32 on a HIP device (no CPU path)");
    const std::shared_ptr<ParamPack> pk = pack_get(pack_id);
    if (!x.is_contiguous()) x = x.contiguous();
    TORCH_CHECK(pk->w_down.is_contiguous() && pk->w_up.is_contiguous(),
                "gconv_adapter_amd: adapter weights must be contiguous");
    const int64_t n = x.size(0), d = x.size(1), r = pk->w_down.size(0);
    gconv_graph_t g;
    std::memcpy(&g, reinterpret_cast<const void*>(graph_addr), sizeof(g));
    TORCH_CHECK(n == g.fwd.num_nodes, "x has ", n, " nodes but the graph plan was built for ",
                g.fwd.num_nodes);
    at::Tensor out = at::empty_like(x);
    at::Tensor saved = at::empty({act == GCONV_ACT_SILU ? 3 : 2, n, fp}, x.options());
    at::Tensor ws = forward_workspace(x, ws_fwd);
    float* sv = saved.data_ptr<float>();
    gconv_adapter_params_t p{fptr(pk->w_down), fptr(pk->b_down), fptr(pk->w_up), fptr(pk->b_up),
                             fptr(pk->scalar), static_cast<int32_t>(d), static_cast<int32_t>(r),
                             static_cast<int32_t>(act), skip ? 1 : 0, static_cast<int32_t>(walk)};
    check(g_lib.fwd(&g, &p, x.data_ptr<float>(), out.data_ptr<float>(), sv,
                    act == GCONV_ACT_SILU ? sv + 2 * n * fp : nullptr, sv + n * fp, ws.data_ptr(),
                    static_cast<size_t>(ws_fwd), reinterpret_cast<void*>(stream)),
          "adapter_fwd");
    ctx->save_for_backward({x, saved, arena});
    ctx->saved_data["g"] = std::string(reinterpret_cast<const char*>(&g), sizeof(g));
    ctx->saved_data["i"] = std::vector<int64_t>{pack_id, fp, ws_bwd, act, skip ? 1 : 0, stream, walk};
    return out;
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::variable_list grads) {
    auto sv = ctx->get_saved_variables();
    at::Tensor x = sv[0], saved = sv[1];
    gconv_graph_t g;
    const std::string& gs = ctx->saved_data["g"].toStringRef();
    std::memcpy(&g, gs.data(), sizeof(g));
    const std::vector<int64_t> iv = ctx->saved_data["i"].toIntVector();
    const int64_t fp = iv[1], ws_bwd = iv[2], act = iv[3], stream = iv[5];
    const bool skip = iv[4] != 0;
    const std::shared_ptr<ParamPack> pk = pack_get(iv[0]);
    at::Tensor go = grads[0];
    TORCH_CHECK(go.defined() && go.is_cuda() && go.scalar_type() == at::kFloat,
                "gconv_adapter_amd: grad_output must be an fp32 HIP tensor");
    if (!go.is_contiguous()) go = go.contiguous();
    const int64_t n = x.size(0), d = x.size(1), r = pk->w_down.size(0);
    at::NoGradGuard no_grad;
    at::Tensor gx = ctx->needs_input_grad(0) ? at::empty_like(x) : at::Tensor();
    // A graph task that names its inputs (torch.autograd.grad(out, x), backward(inputs=[...])):
    // a parameter receives its gradient only if it is one of them — its grad accumulator is then
    // in the task's exec_info, marked for execution. AccumulateGrad would run for exactly those.
    // With none of them named the partials are dropped (nothing is stored, no reduction runs).
    const auto* exec_info = torch::autograd::get_current_graph_task_exec_info();
    const bool named = exec_info && !exec_info->empty();
    auto wanted = [&](const at::Tensor& prm) {
      if (!prm.defined() || !prm.requires_grad()) return false;
      if (!named) return true;
      const auto acc = torch::autograd::impl::try_get_grad_accumulator(prm);
      if (!acc) return false;
      const auto it = exec_info->find(acc.get());
      return it != exec_info->end() && it->second.should_execute();
    };
    const bool want[5] = {wanted(pk->w_down), wanted(pk->b_down), wanted(pk->w_up), wanted(pk->b_up),
                          wanted(pk->scalar)};
    const bool any_wanted = want[0] || want[1] || want[2] || want[3] || want[4];
    // Where the five parameter gradients go. A parameter with a registered sink (a view of a flat
    // all-reduce / optimizer buffer) and no gradient yet gets its gradient written there, once per
    // pass; everything else goes to views of ONE fresh allocation (one allocator round trip instead
    // of five per backward node; every view starts on a 16-byte boundary).
    const std::shared_ptr<TaskState> st = (!named || any_wanted) ? current_task_state() : nullptr;
    at::Tensor gout[5];
    bool to_sink[5] = {false, false, false, false, false};
    if (st) {
      std::lock_guard<std::mutex> lk(st->mu);
      for (int i = 0; i < 5; ++i) {
        const at::Tensor& prm = pk->param(i);
        if (!want[i] || !pk->sink[i].defined() || prm.grad().defined() || pk->sink[i].sizes() != prm.sizes()) continue;
        auto* impl = prm.unsafeGetTensorImpl();
        if (std::find(st->sunk.begin(), st->sunk.end(), impl) != st->sunk.end()) continue;   // second use this pass
        st->sunk.push_back(impl);
        gout[i] = pk->sink[i];
        to_sink[i] = true;
      }
    }
    auto pad4 = [](int64_t v) { return (v + 3) / 4 * 4; };
    int64_t need = 0, off5[5];
    for (int i = 0; i < 5; ++i) {
      off5[i] = need;
      if (pk->param(i).defined() && !to_sink[i]) need += pad4(pk->param(i).numel());
    }
    if (need > 0) {
      at::Tensor flat = at::empty({need}, x.options());
      for (int i = 0; i < 5; ++i)
        if (pk->param(i).defined() && !to_sink[i])
          gout[i] = flat.narrow(0, off5[i], pk->param(i).numel()).view(pk->param(i).sizes());
    }
    at::Tensor &gwd = gout[0], &gbd = gout[1], &gwu = gout[2], &gbu = gout[3], &gsc = gout[4];
    at::Tensor ws = at::empty({ws_bwd}, x.options().dtype(at::kByte));
    const float* svp = saved.data_ptr<float>();
    gconv_adapter_params_t p{fptr(pk->w_down), fptr(pk->b_down), fptr(pk->w_up), fptr(pk->b_up),
                             fptr(pk->scalar), static_cast<int32_t>(d), static_cast<int32_t>(r),
                             static_cast<int32_t>(act), skip ? 1 : 0, static_cast<int32_t>(iv[6])};
    gconv_pending_finalize_t pend;
    check(g_lib.bwd_partial(&g, &p, x.data_ptr<float>(), go.data_ptr<float>(), svp,
                            act == GCONV_ACT_SILU ? svp + 2 * n * fp : nullptr, svp + n * fp,
                            fptr_mut(gx), gwd.data_ptr<float>(), fptr_mut(gbd), gwu.data_ptr<float>(),
                            fptr_mut(gbu), fptr_mut(gsc), ws.data_ptr(), static_cast<size_t>(ws_bwd),
                            &pend, reinterpret_cast<void*>(stream)),
          "adapter_bwd");
    if (named && !any_wanted)
      return {gx, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
              at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    {
      std::lock_guard<std::mutex> lk(st->mu);
      if (!st->pending.empty() && st->stream != stream) flush_locked(*st);   // another stream: do not mix
      st->stream = stream;
      st->device = x.get_device();
      st->pending.push_back(pend);
      st->keep.push_back(ws);
      queue_grad(pk->w_down, std::move(gwd), want[0], *st);
      queue_grad(pk->b_down, std::move(gbd), want[1], *st);
      queue_grad(pk->w_up, std::move(gwu), want[2], *st);
      queue_grad(pk->b_up, std::move(gbu), want[3], *st);
      queue_grad(pk->scalar, std::move(gsc), want[4], *st);
    }
    at::Tensor none;
    return {gx, none, none, none, none, none, none, none, none, none, none, none};
  }
};

at::Tensor adapter_apply_direct(at::Tensor x, c10::optional<at::Tensor> anchor, int64_t pack_id,
                                at::Tensor arena, int64_t graph_addr, int64_t fp, int64_t ws_fwd,
                                int64_t ws_bwd, int64_t act, bool skip, int64_t stream, int64_t walk) {
  {
    const std::shared_ptr<ParamPack> pk = pack_get(pack_id);
    if (has_grad_observers(pk->w_down) || has_grad_observers(pk->b_down) || has_grad_observers(pk->w_up) ||
        has_grad_observers(pk->b_up) || has_grad_observers(pk->scalar)) {
      auto opt = [](const at::Tensor& t) { return t.defined() ? c10::optional<at::Tensor>(t) : c10::nullopt; };
      return AdapterFn::apply(x, pk->w_down, opt(pk->b_down), pk->w_up, opt(pk->b_up), opt(pk->scalar), arena,
                              graph_addr, fp, ws_fwd, ws_bwd, act, skip, stream, walk);
    }
  }
  pack_get(pack_id)->ensure_accumulators();
  return AdapterFnDirect::apply(x, anchor, pack_id, arena, graph_addr, fp, ws_fwd, ws_bwd, act, skip, stream, walk);
}

// Forward only (gconv_adapter_infer_f32): no autograd node, no saved tensors — what GConvAdapter.forward calls under
// torch.no_grad() / when nothing that reaches it requires a gradient (the reference's evaluation passes,
// /root/reference/scripts/finetune_inductive_learning.py:95-104). `arena` keeps the plan's memory referenced for
// the duration of the call only: the launch is ordered on `stream` like every other use of the plan.
at::Tensor adapter_infer(at::Tensor x, at::Tensor w_down, c10::optional<at::Tensor> b_down_, at::Tensor w_up,
                         c10::optional<at::Tensor> b_up_, c10::optional<at::Tensor> scalar_, at::Tensor arena,
                         int64_t graph_addr, int64_t ws_bytes, int64_t act, bool skip, int64_t stream,
                         int64_t walk) {
  at::Tensor b_down = b_down_.value_or(at::Tensor()), b_up = b_up_.value_or(at::Tensor()),
             scalar = scalar_.value_or(at::Tensor());
  TORCH_CHECK(g_lib.handle, "gconv_adapter_amd: library not loaded");
  TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat,
              "gconv_adapter_amd computes in fp32 on a HIP device (no CPU path)");
  (void)arena;
  if (!x.is_contiguous()) x = x.contiguous();
  if (!w_down.is_contiguous()) w_down = w_down.contiguous();
  if (!w_up.is_contiguous()) w_up = w_up.contiguous();
  const int64_t n = x.size(0), d = x.size(1), r = w_down.size(0);
  gconv_graph_t g;
  std::memcpy(&g, reinterpret_cast<const void*>(graph_addr), sizeof(g));
  TORCH_CHECK(n == g.fwd.num_nodes, "x has ", n, " nodes but the graph plan was built for ", g.fwd.num_nodes);
  at::Tensor out = at::empty_like(x);
  at::Tensor ws = forward_workspace(x, ws_bytes);
  gconv_adapter_params_t p{fptr(w_down), fptr(b_down), fptr(w_up), fptr(b_up), fptr(scalar),
                           static_cast<int32_t>(d), static_cast<int32_t>(r),
                           static_cast<int32_t>(act), skip ? 1 : 0, static_cast<int32_t>(walk)};
  check(g_lib.infer(&g, &p, x.data_ptr<float>(), out.data_ptr<float>(), ws.data_ptr(), static_cast<size_t>(ws_bytes),
                    reinterpret_cast<void*>(stream)),
        "adapter_infer");
  return out;
}

at::Tensor adapter_apply(at::Tensor x, at::Tensor w_down, c10::optional<at::Tensor> b_down,
                         at::Tensor w_up, c10::optional<at::Tensor> b_up,
                         c10::optional<at::Tensor> scalar, at::Tensor arena, int64_t graph_addr,
                         int64_t fp, int64_t ws_fwd, int64_t ws_bwd, int64_t act, bool skip,
                         int64_t stream, int64_t walk) {
  return AdapterFn::apply(x, w_down, b_down, w_up, b_up, scalar, arena, graph_addr, fp, ws_fwd,
                          ws_bwd, act, skip, stream, walk);
}

// ---------------------------------------------------------------------------------------
// Graph-plan prefetch worker: builds the plan of a FUTURE batch (gconv_graph_build on a side
// stream) from a native thread, so the build — a dozen small launches and one stream
// synchronisation — neither holds the interpreter lock nor sits in front of the next step.
// Plays the role of the DataLoader workers that collate the next batch in the reference
// (/root/reference/scripts/finetune_inductive_learning.py:246-258).
//
// Memory: the arena is allocated by the submitting thread on ITS current stream; the side
// stream first waits for an event recorded there at submit time (edge_index is complete and
// every earlier use of the recycled block has drained) — or for the caller's completion event
// (gconv_arm_completion_event). gconv_graph_build returns with its last launch (the size sort of the
// component tiles) still queued on the side stream: the worker synchronises the side stream once more, so
// a plan that plan_take() hands out has nothing pending anywhere and the consumer's stream waits for nothing.
// ---------------------------------------------------------------------------------------
// Events that only ORDER device work (the build's side stream behind the caller's stream): no timing, device-scope
// release (the scope makes no measurable difference here: 4.9 against 5.0 us in front of the stream's next kernel).
static const unsigned kOrderingEventFlags = hipEventDisableTiming | hipEventReleaseToDevice;

struct PlanJob {
  at::Tensor edge_index, arena;
  int64_t num_edges = 0;
  int32_t num_nodes = 0, normalize = 0, device = 0;
  hipEvent_t ready = nullptr;
  bool own_ready = true;   // false: the caller's completion event (plan_submit's ready_event)
  void* side_stream = nullptr;
  gconv_graph_t graph{};
  gconv_status_t status = 0;
  bool done = false;
};

class PlanWorker {
 public:
  ~PlanWorker() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    if (thread_.joinable()) thread_.join();
  }

  // ready_event: 0, or a hipEvent_t that is (or will be, by a launch already made) recorded on `main_stream` behind
  // everything the build has to wait for — the submitting thread then queues nothing on its stream here
  int64_t submit(at::Tensor edge_index, int64_t num_nodes, bool normalize, int64_t main_stream,
                 int64_t side_stream, int64_t ready_event) {
    TORCH_CHECK(g_lib.handle, "gconv_adapter_amd: library not loaded");
    TORCH_CHECK(edge_index.is_cuda() && edge_index.scalar_type() == at::kLong && edge_index.dim() == 2 &&
                    edge_index.size(0) == 2 && edge_index.is_contiguous(),
                "gconv_adapter_amd: edge_index must be a contiguous int64 [2, E] HIP tensor");
    auto job = std::make_shared<PlanJob>();
    job->edge_index = edge_index;
    job->num_edges = edge_index.size(1);
    job->num_nodes = static_cast<int32_t>(num_nodes);
    job->normalize = normalize ? 1 : 0;
    job->device = edge_index.get_device();
    job->side_stream = reinterpret_cast<void*>(side_stream);
    const size_t bytes = g_lib.build_bytes(job->num_edges, job->num_nodes);
    job->arena = at::empty({static_cast<int64_t>(bytes)}, edge_index.options().dtype(at::kByte));
    if (ready_event != 0) {
      job->ready = reinterpret_cast<hipEvent_t>(ready_event);
      job->own_ready = false;
    } else {
      TORCH_CHECK(hipEventCreateWithFlags(&job->ready, kOrderingEventFlags) == hipSuccess &&
                      hipEventRecord(job->ready, reinterpret_cast<hipStream_t>(main_stream)) == hipSuccess,
                  "gconv_adapter_amd: cannot record the prefetch event");
    }
    std::lock_guard<std::mutex> lk(mu_);
    const int64_t id = next_id_++;
    jobs_[id] = job;
    queue_.push_back(job);
    if (!thread_.joinable()) thread_ = std::thread([this] { run(); });
    cv_.notify_all();
    return id;
  }

  // (arena, gconv_graph_t bytes, status); blocks until the build call has returned and the side stream has drained
  std::tuple<at::Tensor, py::bytes, int64_t> take(int64_t id, int64_t consumer_stream) {
    std::shared_ptr<PlanJob> job;
    {
      py::gil_scoped_release nogil;
      std::unique_lock<std::mutex> lk(mu_);
      auto it = jobs_.find(id);
      TORCH_CHECK(it != jobs_.end(), "gconv_adapter_amd: unknown prefetch job ", id);
      job = it->second;
      done_cv_.wait(lk, [&] { return job->done; });
      jobs_.erase(it);
    }
    gconv_status_t st = job->status;
    // status > 0 is a hipError_t. The worker has synchronised the side stream behind the build's last launch:
    // nothing is pending there, the consumer's stream needs no wait (see run()).
    (void)consumer_stream;
    if (job->own_ready) (void)hipEventDestroy(job->ready);
    job->status = st;
    return {job->arena, py::bytes(reinterpret_cast<const char*>(&job->graph), sizeof(gconv_graph_t)),
            static_cast<int64_t>(job->status)};
  }

 private:
  void run() {
    int cur = -1;
    for (;;) {
      std::shared_ptr<PlanJob> job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
        if (stop_) return;
        job = queue_.front();
        queue_.pop_front();
      }
      if (job->device != cur) {
        (void)hipSetDevice(job->device);
        cur = job->device;
      }
      hipStream_t side = static_cast<hipStream_t>(job->side_stream);
      gconv_status_t st = static_cast<gconv_status_t>(hipStreamWaitEvent(side, job->ready, 0));
      if (st == 0)
        // (a prefetched plan is built next to the step's kernels: no grid-wide launch)
        st = g_lib.build_ex(job->edge_index.data_ptr<int64_t>(), job->num_edges, job->num_nodes, job->normalize,
                            GCONV_BUILD_BACKGROUND, job->arena.data_ptr(), static_cast<size_t>(job->arena.numel()),
                            &job->graph, side);
      // gconv_graph_build returns with its last launch (the size sort of > 256 tiles) still queued: wait for it here,
      // on this thread, so that the consumer needs no cross-stream wait. An event recorded behind the sort and
      // queried / waited for by plan_take cost the consumer stream's next kernel 6 us of idle GPU every step
      // (tools/lab/r05_timeline.sh: the gap in front of the step's first kernel; gone with the event). Only batches of
      // more than 256 tiles have that launch, and their steps leave the host the time.
      if (st == 0) st = static_cast<gconv_status_t>(hipStreamSynchronize(side));
      {
        std::lock_guard<std::mutex> lk(mu_);
        job->status = st;
        job->done = true;
      }
      done_cv_.notify_all();
    }
  }

  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  std::deque<std::shared_ptr<PlanJob>> queue_;
  std::unordered_map<int64_t, std::shared_ptr<PlanJob>> jobs_;
  std::thread thread_;
  int64_t next_id_ = 1;
  bool stop_ = false;
};

PlanWorker& worker() {
  static PlanWorker* w = new PlanWorker();  // leaked on purpose: no thread join at interpreter exit
  return *w;
}

int64_t plan_submit(at::Tensor edge_index, int64_t num_nodes, bool normalize, int64_t main_stream,
                    int64_t side_stream, int64_t ready_event) {
  return worker().submit(std::move(edge_index), num_nodes, normalize, main_stream, side_stream, ready_event);
}
// a HIP event for gconv_arm_completion_event (graph.CompletionEvent owns it)
int64_t event_create() {
  hipEvent_t ev = nullptr;
  TORCH_CHECK(hipEventCreateWithFlags(&ev, kOrderingEventFlags) == hipSuccess, "gconv_adapter_amd: hipEventCreate failed");
  return reinterpret_cast<int64_t>(ev);
}
void event_destroy(int64_t ev) { (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)); }
bool event_query(int64_t ev) { return hipEventQuery(reinterpret_cast<hipEvent_t>(ev)) == hipSuccess; }
void arm_completion_event(int64_t ev) {
  TORCH_CHECK(g_lib.handle, "gconv_adapter_amd: library not loaded");
  check(g_lib.arm(reinterpret_cast<void*>(ev)), "arm_completion_event");
}
std::tuple<at::Tensor, py::bytes, int64_t> plan_take(int64_t id, int64_t consumer_stream) {
  return worker().take(id, consumer_stream);
}


// out = scalar * LayerNorm(z): the adapter's layer_norm tail (gconv_adapter.py:101-102,105-106),
// mirroring functional.LayerNormScaleFunction (csrc/norm.hip does the work)
struct LayerNormScaleFn : public torch::autograd::Function<LayerNormScaleFn> {
  static at::Tensor forward(torch::autograd::AutogradContext* ctx, at::Tensor z,
                            c10::optional<at::Tensor> gamma_, c10::optional<at::Tensor> beta_,
                            c10::optional<at::Tensor> scalar_, double eps, int64_t stream) {
    at::Tensor gamma = gamma_.value_or(at::Tensor()), beta = beta_.value_or(at::Tensor()),
               scalar = scalar_.value_or(at::Tensor());
    TORCH_CHECK(g_lib.handle, "gconv_adapter_amd: library not loaded");
    TORCH_CHECK(z.is_cuda() && z.scalar_type() == at::kFloat && z.dim() >= 1,
                "gconv_adapter_amd computes in fp32 on a HIP device (no CPU path)");
    if (!z.is_contiguous()) z = z.contiguous();
    const int64_t d = z.size(-1), n = d > 0 ? z.numel() / d : 0;
    at::Tensor out = at::empty_like(z);
    at::Tensor stats = at::empty({n, 2}, z.options());
    check(g_lib.ln_fwd(z.data_ptr<float>(), fptr(gamma), fptr(beta), fptr(scalar), out.data_ptr<float>(),
                       stats.data_ptr<float>(), static_cast<int32_t>(n), static_cast<int32_t>(d),
                       static_cast<float>(eps), reinterpret_cast<void*>(stream)),
          "layer_norm_scale_fwd");
    ctx->save_for_backward({z, stats, gamma, beta, scalar});
    ctx->saved_data["stream"] = stream;
    return out;
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::variable_list grads) {
    auto sv = ctx->get_saved_variables();
    at::Tensor z = sv[0], stats = sv[1], gamma = sv[2], beta = sv[3], scalar = sv[4];
    at::Tensor go = grads[0];
    if (!go.is_contiguous()) go = go.contiguous();
    const int64_t d = z.size(-1), n = d > 0 ? z.numel() / d : 0;
    at::Tensor gz, gg, gb, gs;
    if (ctx->needs_input_grad(0)) gz = at::empty_like(z);
    if (gamma.defined()) gg = at::empty_like(gamma);
    if (beta.defined()) gb = at::empty_like(beta);
    if (scalar.defined()) gs = at::empty_like(scalar);
    const size_t bytes = g_lib.ln_bytes(static_cast<int32_t>(n), static_cast<int32_t>(d));
    at::Tensor ws = at::empty({static_cast<int64_t>(bytes > 16 ? bytes : 16)}, z.options().dtype(at::kByte));
    check(g_lib.ln_bwd(go.data_ptr<float>(), z.data_ptr<float>(), stats.data_ptr<float>(), fptr(gamma),
                       fptr(beta), fptr(scalar), fptr_mut(gz), fptr_mut(gg), fptr_mut(gb), fptr_mut(gs),
                       static_cast<int32_t>(n), static_cast<int32_t>(d), ws.data_ptr(),
                       static_cast<size_t>(ws.numel()), reinterpret_cast<void*>(ctx->saved_data["stream"].toInt())),
          "layer_norm_scale_bwd");
    return {gz, gg, gb, gs, at::Tensor(), at::Tensor()};
  }
};

at::Tensor layer_norm_scale(at::Tensor z, c10::optional<at::Tensor> gamma, c10::optional<at::Tensor> beta,
                            c10::optional<at::Tensor> scalar, double eps, int64_t stream) {
  return LayerNormScaleFn::apply(std::move(z), std::move(gamma), std::move(beta), std::move(scalar), eps, stream);
}

// ---- the adapters' surroundings in the fine-tune step as C++ autograd nodes ----------------------
// On a 512-molecule shard the step is host-bound once the kernels are fused (~80 launches per step): the Python
// bindings of BatchNorm (norm.BatchNormFunction) and of the frozen GIN layer (gin.GinFrozenLayerFunction) cost
// 31 + 43 us of Python per layer forward and as much again on the autograd thread. The same C ABI calls in the
// same order from here: one host call forward, none from Python backward.

// y = [relu](gamma (x - mean_batch) rstd_batch + beta), running statistics updated in place
// (/root/reference/src/models/inductive/gnn.py:108-110 with the ReLU of :127-128 folded in)
struct BatchNormFn : public torch::autograd::Function<BatchNormFn> {
  static at::Tensor forward(torch::autograd::AutogradContext* ctx, at::Tensor x, c10::optional<at::Tensor> gamma_,
                            c10::optional<at::Tensor> beta_, c10::optional<at::Tensor> rmean_,
                            c10::optional<at::Tensor> rvar_, c10::optional<at::Tensor> tracked_, double momentum,
                            double eps, bool relu, int64_t stream) {
    at::Tensor gamma = gamma_.value_or(at::Tensor()), beta = beta_.value_or(at::Tensor()),
               rmean = rmean_.value_or(at::Tensor()), rvar = rvar_.value_or(at::Tensor()),
               tracked = tracked_.value_or(at::Tensor());   // num_batches_tracked: counted by the kernel
    TORCH_CHECK(!tracked.defined() || (tracked.is_cuda() && tracked.scalar_type() == at::kLong),
                "gconv_adapter_amd.BatchNorm1d: num_batches_tracked must be an int64 tensor on the device");
    TORCH_CHECK(g_lib.handle, "gconv_adapter_amd: library not loaded");
    TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.dim() == 2,
                "gconv_adapter_amd.BatchNorm1d takes [N, d] fp32 node features on a HIP device");
    if (!x.is_contiguous()) x = x.contiguous();
    const int32_t n = static_cast<int32_t>(x.size(0)), d = static_cast<int32_t>(x.size(1));
    at::Tensor y = at::empty_like(x);
    at::Tensor stats = at::empty({2, d}, x.options());
    const size_t bytes = g_lib.bn_bytes(n, d);
    at::Tensor ws = at::empty({static_cast<int64_t>(bytes > 16 ? bytes : 16)}, x.options().dtype(at::kByte));
    check(g_lib.bn_fwd(x.data_ptr<float>(), fptr(gamma), fptr(beta), fptr_mut(rmean), fptr_mut(rvar),
                       tracked.defined() ? tracked.data_ptr<int64_t>() : nullptr, static_cast<float>(momentum), static_cast<float>(eps), relu ? 1 : 0, y.data_ptr<float>(),
                       stats.data_ptr<float>(), stats.data_ptr<float>() + d, n, d, ws.data_ptr(),
                       static_cast<size_t>(ws.numel()), reinterpret_cast<void*>(stream)),
          "batch_norm_fwd");
    ctx->save_for_backward({x, relu ? y : at::Tensor(), stats, gamma});
    ctx->saved_data["has_beta"] = beta.defined();
    ctx->saved_data["stream"] = stream;
    return y;
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::variable_list grads) {
    auto sv = ctx->get_saved_variables();
    at::Tensor x = sv[0], y = sv[1], stats = sv[2], gamma = sv[3];
    at::Tensor g = grads[0];
    if (!g.is_contiguous()) g = g.contiguous();
    const int32_t n = static_cast<int32_t>(x.size(0)), d = static_cast<int32_t>(x.size(1));
    at::Tensor gx, gg, gb;
    if (ctx->needs_input_grad(0)) gx = at::empty_like(x);
    if (gamma.defined()) gg = at::empty({d}, x.options());
    if (ctx->saved_data["has_beta"].toBool()) gb = at::empty({d}, x.options());
    const size_t bytes = g_lib.bn_bytes(n, d);
    at::Tensor ws = at::empty({static_cast<int64_t>(bytes > 16 ? bytes : 16)}, x.options().dtype(at::kByte));
    check(g_lib.bn_bwd(g.data_ptr<float>(), x.data_ptr<float>(), fptr(y), stats.data_ptr<float>(),
                       stats.data_ptr<float>() + d, fptr(gamma), fptr_mut(gx), fptr_mut(gg), fptr_mut(gb), n, d,
                       ws.data_ptr(), static_cast<size_t>(ws.numel()),
                       reinterpret_cast<void*>(ctx->saved_data["stream"].toInt())),
          "batch_norm_bwd");
    return {gx, gg, gb, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
            at::Tensor()};
  }
};

at::Tensor batch_norm(at::Tensor x, c10::optional<at::Tensor> gamma, c10::optional<at::Tensor> beta,
                      c10::optional<at::Tensor> running_mean, c10::optional<at::Tensor> running_var,
                      c10::optional<at::Tensor> num_batches_tracked, double momentum, double eps, bool relu,
                      int64_t stream) {
  return BatchNormFn::apply(std::move(x), std::move(gamma), std::move(beta), std::move(running_mean),
                            std::move(running_var), std::move(num_batches_tracked), momentum, eps, relu, stream);
}

// The whole frozen GIN layer (/root/reference/src/layers/inductive/gin_conv.py:38-85 with every parameter
// frozen, the adapter recipe of gnn.py:230-241) as one node, mirroring gin.GinFrozenLayerFunction: aggregation
// -> the MLP's operand pair -> two GEMMs; backward: split of dY, two GEMMs, the aggregation on the transposed
// CSR. The CSR structs are copied by value (their arrays live in `arena`, which the node keeps alive); the
// weights arrive as the scaled fp16 pairs gin._PairWeights caches.
struct GinFrozenLayerFn : public torch::autograd::Function<GinFrozenLayerFn> {
  static at::Tensor pair_buffer(const at::Tensor& like, int64_t rows, int64_t features) {
    const size_t bytes = g_lib.pair_bytes(static_cast<int32_t>(rows), static_cast<int32_t>(features));
    return at::empty({static_cast<int64_t>(bytes)}, like.options().dtype(at::kByte));
  }

  static at::Tensor forward(torch::autograd::AutogradContext* ctx, at::Tensor x, at::Tensor arena, int64_t fwd_csr,
                            int64_t bwd_csr, at::Tensor counts, at::Tensor emb_type, at::Tensor emb_dir,
                            at::Tensor w1_pair, c10::optional<at::Tensor> b1_, at::Tensor w2_pair,
                            c10::optional<at::Tensor> b2_, at::Tensor w2t_pair, at::Tensor w1t_pair, int64_t hidden,
                            int64_t out_features, int64_t stream) {
    at::Tensor b1 = b1_.value_or(at::Tensor()), b2 = b2_.value_or(at::Tensor());
    TORCH_CHECK(g_lib.handle, "gconv_adapter_amd: library not loaded");
    TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.dim() == 2,
                "gconv_adapter_amd.MolecularGINConv takes [N, d] fp32 node features on a HIP device");
    TORCH_CHECK(fwd_csr != 0 && bwd_csr != 0 && arena.defined(), "gin_frozen_layer: no graph plan");
    if (!x.is_contiguous()) x = x.contiguous();
    const int32_t n = static_cast<int32_t>(x.size(0)), d = static_cast<int32_t>(x.size(1));
    const gconv_csr_t fwd = *reinterpret_cast<const gconv_csr_t*>(fwd_csr);
    void* st = reinterpret_cast<void*>(stream);
    at::Tensor a2 = pair_buffer(x, n, d);
    check(g_lib.gin_agg_pair(&fwd, x.data_ptr<float>(), static_cast<const int32_t*>(counts.data_ptr()), emb_type.data_ptr<float>(),
                             emb_dir.data_ptr<float>(), a2.data_ptr(), d, st),
          "gin_aggregate_f16x2");
    at::Tensor h2 = pair_buffer(x, n, hidden);
    check(g_lib.pair_gemm(a2.data_ptr(), w1_pair.data_ptr(), fptr(b1), nullptr, nullptr, h2.data_ptr(), n, d,
                          static_cast<int32_t>(hidden), 1, st),
          "mlp_gemm_f16x2");
    at::Tensor y = at::empty({n, out_features}, x.options());
    check(g_lib.pair_gemm(h2.data_ptr(), w2_pair.data_ptr(), fptr(b2), nullptr, y.data_ptr<float>(), nullptr, n,
                          static_cast<int32_t>(hidden), static_cast<int32_t>(out_features), 0, st),
          "mlp_gemm_f16x2");
    ctx->save_for_backward({h2, arena, w2t_pair, w1t_pair});
    ctx->saved_data["bwd_csr"] = std::string(reinterpret_cast<const char*>(bwd_csr), sizeof(gconv_csr_t));
    ctx->saved_data["dims"] = std::vector<int64_t>{n, d, hidden, out_features};
    ctx->saved_data["stream"] = stream;
    return y;
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::variable_list grads) {
    torch::autograd::variable_list out(16);
    if (!ctx->needs_input_grad(0)) return out;
    auto sv = ctx->get_saved_variables();
    at::Tensor h2 = sv[0], w2t = sv[2], w1t = sv[3];
    at::Tensor gy = grads[0];
    if (!gy.is_contiguous()) gy = gy.contiguous();
    const auto dims = ctx->saved_data["dims"].toIntVector();
    const int32_t n = static_cast<int32_t>(dims[0]), d = static_cast<int32_t>(dims[1]),
                  hidden = static_cast<int32_t>(dims[2]), outf = static_cast<int32_t>(dims[3]);
    gconv_csr_t bwd;
    const std::string& raw = ctx->saved_data["bwd_csr"].toStringRef();
    std::memcpy(&bwd, raw.data(), sizeof(bwd));
    void* st = reinterpret_cast<void*>(ctx->saved_data["stream"].toInt());
    at::Tensor g2 = pair_buffer(gy, n, outf);
    check(g_lib.split_rows(gy.data_ptr<float>(), n, outf, g2.data_ptr(), st), "mlp_split_rows_f16x2");
    at::Tensor gh2 = pair_buffer(gy, n, hidden);
    check(g_lib.pair_gemm(g2.data_ptr(), w2t.data_ptr(), nullptr, h2.data_ptr(), nullptr, gh2.data_ptr(), n, outf,
                          hidden, 2, st),
          "mlp_gemm_f16x2");
    at::Tensor ga = at::empty({n, d}, gy.options());
    check(g_lib.pair_gemm(gh2.data_ptr(), w1t.data_ptr(), nullptr, nullptr, ga.data_ptr<float>(), nullptr, n, hidden,
                          d, 0, st),
          "mlp_gemm_f16x2");
    at::Tensor gx = at::empty({n, d}, gy.options());
    check(g_lib.gin_agg(&bwd, ga.data_ptr<float>(), nullptr, nullptr, nullptr, gx.data_ptr<float>(), d, st),
          "gin_aggregate");
    out[0] = gx;
    return out;
  }
};

at::Tensor gin_frozen_layer(at::Tensor x, at::Tensor arena, int64_t fwd_csr, int64_t bwd_csr, at::Tensor counts,
                            at::Tensor emb_type, at::Tensor emb_dir, at::Tensor w1_pair, c10::optional<at::Tensor> b1,
                            at::Tensor w2_pair, c10::optional<at::Tensor> b2, at::Tensor w2t_pair,
                            at::Tensor w1t_pair, int64_t hidden, int64_t out_features, int64_t stream) {
  return GinFrozenLayerFn::apply(std::move(x), std::move(arena), fwd_csr, bwd_csr, std::move(counts),
                                 std::move(emb_type), std::move(emb_dir), std::move(w1_pair), std::move(b1),
                                 std::move(w2_pair), std::move(b2), std::move(w2t_pair), std::move(w1t_pair), hidden,
                                 out_features, stream);
}

}  // namespace

// ---------------------------------------------------------------------------------------
// torch custom ops: `torch.ops.gconv.*`. The dispatcher-visible surface of the
// same C ABI calls, for callers that want the pieces rather than the nn.Module — isolated
// benchmarking (`spmm`), a maintainer wiring the adapter into their own autograd code
// (`adapter_fwd` / `adapter_bwd`), torch.library tooling. The module's fast path above stays on
// pybind (a dispatcher round trip is ~2 us of a ~10 us host path) and computes the same numbers:
// both end in gconv_adapter_fwd_f32 / gconv_adapter_bwd_f32.
//
// A graph travels as two tensors: `arena` (uint8, device: every array of the plan) and `plan`
// (uint8, CPU: the gconv_graph_t whose pointers point into the arena). Keep them together.
// ---------------------------------------------------------------------------------------
namespace ops {

int64_t act_code(const std::string& act) {
  if (act == "none") return GCONV_ACT_NONE;
  if (act == "relu") return GCONV_ACT_RELU;
  if (act == "silu" || act == "swish") return GCONV_ACT_SILU;
  TORCH_CHECK(false, "gconv: non_linearity must be one of none / relu / silu (swish), got '", act, "'");
}

// torch's current stream on the tensor's device, as the hipStream_t the C ABI takes
void* current_stream(const at::Tensor& t) {
  return c10::hip::getCurrentHIPStream(static_cast<c10::DeviceIndex>(t.get_device())).stream();
}

gconv_graph_t plan_struct(const at::Tensor& arena, const at::Tensor& plan) {
  TORCH_CHECK(plan.device().is_cpu() && plan.scalar_type() == at::kByte && plan.is_contiguous() &&
                  plan.numel() == static_cast<int64_t>(sizeof(gconv_graph_t)),
              "gconv: `plan` must be the CPU uint8 tensor returned by gconv::graph_plan");
  TORCH_CHECK(arena.is_cuda() && arena.scalar_type() == at::kByte, "gconv: `arena` must be the device tensor "
              "returned by gconv::graph_plan");
  gconv_graph_t g;
  std::memcpy(&g, plan.data_ptr(), sizeof(g));
  const char* lo = static_cast<const char*>(arena.data_ptr());
  const char* rp = reinterpret_cast<const char*>(g.fwd.rowptr);
  TORCH_CHECK(rp >= lo && rp < lo + arena.numel(), "gconv: `plan` does not belong to this `arena`");
  return g;
}

// graph_plan(edge_index, num_nodes, normalize) -> (arena, plan)
std::tuple<at::Tensor, at::Tensor> graph_plan(const at::Tensor& edge_index, int64_t num_nodes, bool normalize) {
  TORCH_CHECK(g_lib.handle, "gconv_adapter_amd: library not loaded");
  TORCH_CHECK(edge_index.dim() == 2 && edge_index.size(0) == 2, "edge_index must have shape [2, num_edges]");
  TORCH_CHECK(edge_index.scalar_type() == at::kLong, "edge_index must be int64 (PyG convention)");
  TORCH_CHECK(edge_index.is_cuda(), "gconv_adapter_amd has no CPU path: edge_index must live on a HIP device");
  const at::Tensor ei = edge_index.contiguous();
  const int64_t e = ei.size(1);
  const size_t bytes = g_lib.build_bytes(e, static_cast<int32_t>(num_nodes));
  at::Tensor arena = at::empty({static_cast<int64_t>(bytes)}, ei.options().dtype(at::kByte));
  at::Tensor plan = at::empty({static_cast<int64_t>(sizeof(gconv_graph_t))}, at::TensorOptions().dtype(at::kByte));
  gconv_graph_t g;
  std::memset(&g, 0, sizeof(g));
  c10::OptionalDeviceGuard guard(ei.device());
  check(g_lib.build(ei.data_ptr<int64_t>(), e, static_cast<int32_t>(num_nodes), normalize ? 1 : 0, arena.data_ptr(),
                    bytes, &g, current_stream(ei)),
        "graph_build");
  std::memcpy(plan.data_ptr(), &g, sizeof(g));
  return {arena, plan};
}

at::Tensor arena_view(const at::Tensor& arena, const void* ptr, int64_t count, at::ScalarType dt) {
  if (ptr == nullptr || count <= 0) return at::empty({0}, arena.options().dtype(dt));
  const int64_t off = static_cast<const char*>(ptr) - static_cast<const char*>(arena.data_ptr());
  const int64_t isz = static_cast<int64_t>(c10::elementSize(dt));
  TORCH_CHECK(off >= 0 && off + count * isz <= arena.numel(), "gconv: plan array outside its arena");
  return arena.narrow(0, off, count * isz).view(dt);
}

// graph_prep(edge_index, num_nodes, normalize) -> (rowptr, col, dis, symmetric, rowptr_t, col_t):
// the CSR of S (in-edges per target row, columns ascending, self-loops dropped when normalising),
// deg^-1/2 (empty when not normalising), the exact symmetry flag and — for directed edge lists —
// the CSR of S^T (empty when symmetric: S^T == S). Views into one arena.
std::tuple<at::Tensor, at::Tensor, at::Tensor, bool, at::Tensor, at::Tensor> graph_prep(const at::Tensor& edge_index,
                                                                                      int64_t num_nodes,
                                                                                      bool normalize) {
  auto [arena, plan] = graph_plan(edge_index, num_nodes, normalize);
  gconv_graph_t g;
  std::memcpy(&g, plan.data_ptr(), sizeof(g));
  const int64_t n = g.fwd.num_nodes;
  at::Tensor rowptr = arena_view(arena, g.fwd.rowptr, n + 1, at::kInt), col = arena_view(arena, g.fwd.col, g.fwd.nnz, at::kInt);
  at::Tensor dis = arena_view(arena, g.dis, g.dis ? n : 0, at::kFloat);
  at::Tensor rowptr_t = arena_view(arena, g.symmetric ? nullptr : g.bwd.rowptr, n + 1, at::kInt);
  at::Tensor col_t = arena_view(arena, g.symmetric ? nullptr : g.bwd.col, g.bwd.nnz, at::kInt);
  return {rowptr, col, dis, g.symmetric != 0, rowptr_t, col_t};
}

struct Shapes { int64_t n, d, r, fp; };
// every parameter tensor a kernel will read through a raw pointer: an fp32 HIP tensor on x's device with
// exactly the elements the kernel indexes (a CPU weight or a short bias would otherwise be a device
// fault / an out-of-bounds read instead of an error)
void check_param(const at::Tensor& t, const at::Tensor& x, int64_t numel, const char* name) {
  if (!t.defined()) return;
  TORCH_CHECK(t.is_cuda() && t.device() == x.device() && t.scalar_type() == at::kFloat && t.is_contiguous() &&
                  t.numel() == numel,
              "gconv: ", name, " must be a contiguous fp32 tensor of ", numel, " elements on ", x.device(), ", got ",
              t.scalar_type(), " ", t.sizes(), " on ", t.device());
}
Shapes check_adapter_args(const gconv_graph_t& g, const at::Tensor& x, const at::Tensor& w_down, const at::Tensor& w_up,
                          const at::Tensor& b_down = at::Tensor(), const at::Tensor& b_up = at::Tensor(),
                          const at::Tensor& scalar = at::Tensor()) {
  TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.dim() == 2,
              "gconv computes in fp32 on a HIP device (no CPU path); x must be [N, d]");
  TORCH_CHECK(w_down.dim() == 2 && w_up.dim() == 2 && w_down.size(1) == x.size(1) && w_up.size(0) == x.size(1) &&
                  w_up.size(1) == w_down.size(0),
              "gconv: conv_down.lin.weight is [r, d], conv_up.lin.weight is [d, r]");
  check_param(w_down, x, w_down.size(0) * x.size(1), "conv_down.lin.weight");
  check_param(w_up, x, w_down.size(0) * x.size(1), "conv_up.lin.weight");
  check_param(b_down, x, w_down.size(0), "conv_down.bias");
  check_param(b_up, x, x.size(1), "conv_up.bias");
  check_param(scalar, x, 1, "scalar");
  TORCH_CHECK(x.size(0) == g.fwd.num_nodes, "x has ", x.size(0), " nodes but the graph plan was built for ", g.fwd.num_nodes);
  const int64_t fp = g_lib.padded_rank(static_cast<int32_t>(w_down.size(0)));
  TORCH_CHECK(fp > 0, "bottleneck_size=", w_down.size(0), " is unsupported (1..64)");
  return {x.size(0), x.size(1), w_down.size(0), fp};
}

// adapter_fwd(x, w_down, b_down?, w_up, b_up?, scalar?, arena, plan, act, skip) -> (out, saved):
// saved = [2 or 3, N, Fp]: H (post-activation, pre-scaled), P, and with SiLU the pre-activation H1
std::tuple<at::Tensor, at::Tensor> adapter_fwd(const at::Tensor& x_, const at::Tensor& w_down_,
                                               const c10::optional<at::Tensor>& b_down_, const at::Tensor& w_up_,
                                               const c10::optional<at::Tensor>& b_up_,
                                               const c10::optional<at::Tensor>& scalar_, const at::Tensor& arena,
                                               const at::Tensor& plan, std::string act_, bool skip) {
  TORCH_CHECK(g_lib.handle, "gconv_adapter_amd: library not loaded");
  const gconv_graph_t g = plan_struct(arena, plan);
  const at::Tensor x = x_.contiguous(), w_down = w_down_.contiguous(), w_up = w_up_.contiguous();
  auto cont = [](const c10::optional<at::Tensor>& t) { return t.has_value() && t->defined() ? t->contiguous() : at::Tensor(); };
  const at::Tensor b_down = cont(b_down_), b_up = cont(b_up_), scalar = cont(scalar_);
  const Shapes sh = check_adapter_args(g, x, w_down, w_up, b_down, b_up, scalar);
  const int64_t act = act_code(act_);
  at::Tensor out = at::empty_like(x);
  at::Tensor saved = at::empty({act == GCONV_ACT_SILU ? 3 : 2, sh.n, sh.fp}, x.options());
  const size_t wsb = std::max<size_t>(g_lib.ws_fwd(&g, static_cast<int32_t>(sh.d), static_cast<int32_t>(sh.r)), 16);
  at::Tensor ws = at::empty({static_cast<int64_t>(wsb)}, x.options().dtype(at::kByte));
  float* sv = saved.data_ptr<float>();
  gconv_adapter_params_t p{fptr(w_down), fptr(b_down), fptr(w_up), fptr(b_up), fptr(scalar), static_cast<int32_t>(sh.d),
                           static_cast<int32_t>(sh.r), static_cast<int32_t>(act), skip ? 1 : 0};
  c10::OptionalDeviceGuard guard(x.device());
  check(g_lib.fwd(&g, &p, x.data_ptr<float>(), out.data_ptr<float>(), sv,
                  act == GCONV_ACT_SILU ? sv + 2 * sh.n * sh.fp : nullptr, sv + sh.n * sh.fp, ws.data_ptr(), wsb,
                  current_stream(x)),
        "adapter_fwd");
  return {out, saved};
}

// adapter_bwd(grad_out, x, saved, w_down, w_up, b_up?, scalar?, arena, plan, act, skip, needs_dx)
//   -> (dx, dw_down, db_down, dw_up, db_up, dscalar); dx is empty when !needs_dx, dscalar when scalar is None
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> adapter_bwd(
    const at::Tensor& grad_out, const at::Tensor& x_, const at::Tensor& saved, const at::Tensor& w_down_,
    const at::Tensor& w_up_, const c10::optional<at::Tensor>& b_up_, const c10::optional<at::Tensor>& scalar_,
    const at::Tensor& arena, const at::Tensor& plan, std::string act_, bool skip, bool needs_dx) {
  TORCH_CHECK(g_lib.handle, "gconv_adapter_amd: library not loaded");
  const gconv_graph_t g = plan_struct(arena, plan);
  const at::Tensor x = x_.contiguous(), w_down = w_down_.contiguous(), w_up = w_up_.contiguous(),
                   go = grad_out.contiguous();
  auto cont = [](const c10::optional<at::Tensor>& t) { return t.has_value() && t->defined() ? t->contiguous() : at::Tensor(); };
  const at::Tensor b_up = cont(b_up_), scalar = cont(scalar_);
  const Shapes sh = check_adapter_args(g, x, w_down, w_up, at::Tensor(), b_up, scalar);
  const int64_t act = act_code(act_);
  TORCH_CHECK(go.is_cuda() && go.scalar_type() == at::kFloat && go.sizes() == x.sizes(),
              "gconv: grad_out must be an fp32 HIP tensor of x's shape");
  TORCH_CHECK(saved.is_cuda() && saved.is_contiguous() && saved.dim() == 3 && saved.size(0) == (act == GCONV_ACT_SILU ? 3 : 2) &&
                  saved.size(1) == sh.n && saved.size(2) == sh.fp,
              "gconv: `saved` is not what gconv::adapter_fwd returned for these arguments");
  at::Tensor dx = needs_dx ? at::empty_like(x) : at::empty({0}, x.options());
  at::Tensor dwd = at::empty_like(w_down), dwu = at::empty_like(w_up);
  at::Tensor dbd = at::empty({sh.r}, x.options()), dbu = at::empty({sh.d}, x.options());
  at::Tensor ds = scalar.defined() ? at::empty_like(scalar) : at::empty({0}, x.options());
  const size_t wsb = std::max<size_t>(g_lib.ws_bwd(&g, static_cast<int32_t>(sh.d), static_cast<int32_t>(sh.r)), 16);
  at::Tensor ws = at::empty({static_cast<int64_t>(wsb)}, x.options().dtype(at::kByte));
  const float* svp = saved.data_ptr<float>();
  gconv_adapter_params_t p{fptr(w_down), nullptr, fptr(w_up), fptr(b_up), fptr(scalar), static_cast<int32_t>(sh.d),
                           static_cast<int32_t>(sh.r), static_cast<int32_t>(act), skip ? 1 : 0};
  c10::OptionalDeviceGuard guard(x.device());
  check(g_lib.bwd(&g, &p, x.data_ptr<float>(), go.data_ptr<float>(), svp,
                  act == GCONV_ACT_SILU ? svp + 2 * sh.n * sh.fp : nullptr, svp + sh.n * sh.fp,
                  needs_dx ? dx.data_ptr<float>() : nullptr, dwd.data_ptr<float>(), dbd.data_ptr<float>(),
                  dwu.data_ptr<float>(), dbu.data_ptr<float>(), scalar.defined() ? ds.data_ptr<float>() : nullptr,
                  ws.data_ptr(), wsb, current_stream(x)),
        "adapter_bwd");
  return {dx, dwd, dbd, dwu, dbu, ds};
}

// spmm(arena, plan, h, transpose) -> S h (or S^T h): h is [N, width], width in {4, 8, 16, 32, 64}
at::Tensor spmm(const at::Tensor& arena, const at::Tensor& plan, const at::Tensor& h_, bool transpose) {
  TORCH_CHECK(g_lib.handle, "gconv_adapter_amd: library not loaded");
  const gconv_graph_t g = plan_struct(arena, plan);
  const at::Tensor h = h_.contiguous();
  TORCH_CHECK(h.is_cuda() && h.scalar_type() == at::kFloat && h.dim() == 2 && h.size(0) == g.fwd.num_nodes,
              "gconv::spmm: h must be an fp32 HIP tensor [num_nodes, width]");
  const gconv_csr_t* csr = transpose ? &g.bwd : &g.fwd;
  const int32_t width = static_cast<int32_t>(h.size(1));
  at::Tensor y = at::empty_like(h);
  const size_t wsb = std::max<size_t>(g_lib.spmm_ws(csr, width), 16);
  at::Tensor ws = at::empty({static_cast<int64_t>(wsb)}, h.options().dtype(at::kByte));
  c10::OptionalDeviceGuard guard(h.device());
  check(g_lib.spmm(csr, g.dis, h.data_ptr<float>(), y.data_ptr<float>(), width, nullptr, GCONV_ACT_NONE, 0u,
                   ws.data_ptr(), wsb, current_stream(h)),
        "spmm");
  return y;
}

// adapter(...) -> out, differentiable: the autograd node is AdapterFn (the same one the module's
// non-direct path uses)
at::Tensor adapter(const at::Tensor& x, const at::Tensor& w_down, const c10::optional<at::Tensor>& b_down,
                   const at::Tensor& w_up, const c10::optional<at::Tensor>& b_up,
                   const c10::optional<at::Tensor>& scalar, const at::Tensor& arena, const at::Tensor& plan,
                   std::string act_, bool skip) {
  TORCH_CHECK(g_lib.handle, "gconv_adapter_amd: library not loaded");
  const gconv_graph_t g = plan_struct(arena, plan);
  auto cont = [](const c10::optional<at::Tensor>& t) { return t.has_value() && t->defined() ? t->contiguous() : at::Tensor(); };
  const Shapes sh = check_adapter_args(g, x, w_down.contiguous(), w_up.contiguous(), cont(b_down), cont(b_up), cont(scalar));
  const int64_t wsf = std::max<int64_t>(g_lib.ws_fwd(&g, static_cast<int32_t>(sh.d), static_cast<int32_t>(sh.r)), 16);
  const int64_t wsb = std::max<int64_t>(g_lib.ws_bwd(&g, static_cast<int32_t>(sh.d), static_cast<int32_t>(sh.r)), 16);
  c10::OptionalDeviceGuard guard(x.device());
  return AdapterFn::apply(x, w_down, b_down, w_up, b_up, scalar, arena, reinterpret_cast<int64_t>(plan.data_ptr()),
                          sh.fp, wsf, wsb, act_code(act_), skip, reinterpret_cast<int64_t>(current_stream(x)), 0);
}

}  // namespace ops

TORCH_LIBRARY(gconv, m) {
  m.def("graph_plan(Tensor edge_index, int num_nodes, bool normalize) -> (Tensor arena, Tensor plan)");
  m.def("graph_prep(Tensor edge_index, int num_nodes, bool normalize) -> (Tensor rowptr, Tensor col, Tensor dis, "
        "bool symmetric, Tensor rowptr_t, Tensor col_t)");
  m.def("adapter_fwd(Tensor x, Tensor w_down, Tensor? b_down, Tensor w_up, Tensor? b_up, Tensor? scalar, Tensor arena, "
        "Tensor plan, str act, bool skip) -> (Tensor out, Tensor saved)");
  m.def("adapter_bwd(Tensor grad_out, Tensor x, Tensor saved, Tensor w_down, Tensor w_up, Tensor? b_up, Tensor? scalar, "
        "Tensor arena, Tensor plan, str act, bool skip, bool needs_dx) -> (Tensor dx, Tensor dw_down, Tensor db_down, "
        "Tensor dw_up, Tensor db_up, Tensor dscalar)");
  m.def("spmm(Tensor arena, Tensor plan, Tensor h, bool transpose) -> Tensor");
  m.def("adapter(Tensor x, Tensor w_down, Tensor? b_down, Tensor w_up, Tensor? b_up, Tensor? scalar, Tensor arena, "
        "Tensor plan, str act, bool skip) -> Tensor");
}
// (catch-all registrations: the kernels check the devices themselves and `plan` is a CPU tensor among
// device tensors; `adapter` builds its own autograd node)
TORCH_LIBRARY_IMPL(gconv, CatchAll, m) {
  m.impl("graph_plan", &ops::graph_plan);
  m.impl("graph_prep", &ops::graph_prep);
  m.impl("adapter_fwd", &ops::adapter_fwd);
  m.impl("adapter_bwd", &ops::adapter_bwd);
  m.impl("spmm", &ops::spmm);
  m.impl("adapter", &ops::adapter);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "C++ autograd bridge to the gconv_adapter C ABI + native graph-plan prefetch worker";
  m.def("load_library", &load_library);
  m.def("adapter_apply", &adapter_apply);
  m.def("adapter_infer", &adapter_infer);
  m.def("adapter_apply_direct", &adapter_apply_direct);
  m.def("pack_create", &pack_create);
  m.def("pack_release", &pack_release);
  // `optimizer.zero_grad(set_to_none=True)` for a list of parameters as ONE host call (the loop in Python costs
  // ~0.28 us per parameter: 7 us of a 155 us ESOL step)
  m.def("clear_grads", [](const std::vector<at::Tensor>& params) {
    for (const at::Tensor& p : params)
      if (p.defined() && p.grad().defined()) p.mutable_grad() = at::Tensor();
  });
  m.def("set_grad_sink", &set_grad_sink, pybind11::arg("param"), pybind11::arg("dest"),
        pybind11::arg("only_if") = c10::optional<at::Tensor>());
  m.def("plan_submit", &plan_submit);
  m.def("event_create", &event_create);
  m.def("event_destroy", &event_destroy);
  m.def("event_query", &event_query);
  m.def("arm_completion_event", &arm_completion_event);
  m.def("plan_take", &plan_take);
  m.def("layer_norm_scale", &layer_norm_scale);
  m.def("batch_norm", &batch_norm);
  m.def("gin_frozen_layer", &gin_frozen_layer);
}
