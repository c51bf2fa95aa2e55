// Autograd binding of the adapter in C++ (host code only, no device code, no HIP headers).
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
#include <torch/extension.h>

#include <cstring>
#include <string>

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

struct Lib {
  void* handle = nullptr;
  fwd_fn_t fwd = nullptr;
  bwd_fn_t bwd = nullptr;
  str_fn_t str = nullptr;
};
Lib g_lib;

void load_library(const std::string& path) {
  if (g_lib.handle) return;
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
  TORCH_CHECK(h != nullptr, "gconv_adapter_amd: cannot dlopen ", path, ": ", dlerror());
  g_lib.fwd = reinterpret_cast<fwd_fn_t>(dlsym(h, "gconv_adapter_fwd_f32"));
  g_lib.bwd = reinterpret_cast<bwd_fn_t>(dlsym(h, "gconv_adapter_bwd_f32"));
  g_lib.str = reinterpret_cast<str_fn_t>(dlsym(h, "gconv_status_string"));
  TORCH_CHECK(g_lib.fwd && g_lib.bwd && g_lib.str, "gconv_adapter_amd: C ABI symbols missing in ", path);
  g_lib.handle = h;
}

const float* fptr(const at::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
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
                            int64_t ws_bwd, int64_t act, bool skip, int64_t stream) {
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
    at::Tensor ws = at::empty({ws_fwd}, x.options().dtype(at::kByte));
    float* sv = saved.data_ptr<float>();
    gconv_adapter_params_t p{fptr(w_down), fptr(b_down), fptr(w_up), fptr(b_up), fptr(scalar),
                             static_cast<int32_t>(d), static_cast<int32_t>(r),
                             static_cast<int32_t>(act), skip ? 1 : 0};
    check(g_lib.fwd(&g, &p, x.data_ptr<float>(), out.data_ptr<float>(), sv,
                    act == GCONV_ACT_SILU ? sv + 2 * n * fp : nullptr, sv + n * fp, ws.data_ptr(),
                    static_cast<size_t>(ws_fwd), reinterpret_cast<void*>(stream)),
          "adapter_fwd");
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
                             static_cast<int32_t>(act), skip ? 1 : 0};
    check(g_lib.bwd(&g, &p, x.data_ptr<float>(), go.data_ptr<float>(), svp,
                    act == GCONV_ACT_SILU ? svp + 2 * n * fp : nullptr, svp + n * fp,
                    fptr_mut(gx), gwd.data_ptr<float>(), fptr_mut(gbd), gwu.data_ptr<float>(),
                    fptr_mut(gbu), fptr_mut(gsc), ws.data_ptr(), static_cast<size_t>(ws_bwd),
                    reinterpret_cast<void*>(stream)),
          "adapter_bwd");
    at::Tensor none;
    return {gx, gwd, gbd, gwu, gbu, gsc, none, none, none, none, none, none, none, none};
  }
};

at::Tensor adapter_apply(at::Tensor x, at::Tensor w_down, c10::optional<at::Tensor> b_down,
                         at::Tensor w_up, c10::optional<at::Tensor> b_up,
                         c10::optional<at::Tensor> scalar, at::Tensor arena, int64_t graph_addr,
                         int64_t fp, int64_t ws_fwd, int64_t ws_bwd, int64_t act, bool skip,
                         int64_t stream) {
  return AdapterFn::apply(x, w_down, b_down, w_up, b_up, scalar, arena, graph_addr, fp, ws_fwd,
                          ws_bwd, act, skip, stream);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "C++ autograd bridge to the gconv_adapter C ABI";
  m.def("load_library", &load_library);
  m.def("adapter_apply", &adapter_apply);
}
