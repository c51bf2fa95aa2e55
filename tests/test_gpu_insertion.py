"""SURVEY.md §8 a7 — every insertion mode of the reference, run on the GPU: adapters in the 'pre'
and / or 'post' position, inserted sequentially (replace the hidden state) or in parallel (added
to the layer's output), around a stand-in backbone with the reference's control flow:

  * molecule shape, 2-D `[N, d]` batch: `MolecularGNN.forward`
    (/root/reference/src/models/inductive/gnn.py:97-130) with this package's frozen GIN layers
    and BatchNorm; adapters called with three positional arguments (`:103,115`);
  * NodeFormer shape, `[1, N, d]`: `NodeFormer.forward`
    (/root/reference/src/models/transductive/nodeformer.py:403-438) with a frozen linear "conv", the
    in-place residual `z += layer_[i]` (`:418`), LayerNorm and ELU; adapters called with two arguments
    (`:407,427`);
  * DIFFormer shape, 2-D `[N, d]`: `DIFFormer.forward`
    (/root/reference/src/models/transductive/difformer.py:216-267). NOT NodeFormer's loop:
    the conv consumes the running `x`, not `layer_[i]` (`:241`), so a sequential pre-adapter reaches the
    output only through the alpha-mix residual `alpha x + (1 - alpha) layer_[i]` (`:244`); LayerNorm
    AFTER the mix (`:246`); a trainable input MLP + ReLU in front (`:221-225`, the freeze rule keeps
    every `fc` trainable, `:207-208`); adapters called with two arguments (`:236,251`).

Each model is compared with the SAME model assembled from the CPU oracles in fp64: loss and the
gradient of every trainable parameter (adapters, norm layers, head). BASELINE configs[4] is the
pre+post case."""
from __future__ import annotations

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import gconv_adapter_amd as ga
from oracle import gin_restated, pyg_restated

from helpers import KinkWatch, assert_close, grad_atol

pytestmark = pytest.mark.gpu

CFG = dict(conv_type="gcn", learnable_scalar=True, non_linearity="relu", normalization="none",
           normalize=True, skip_connection=True)
MODES = [(("pre", "post"), "sequential"), (("pre", "post"), "parallel"), (("pre",), "parallel"),
         (("post",), "sequential"), (("post",), "parallel")]


def _add_adapters(model, positions, kind, d, r, layers, head_names=("head",)):
    """`add_adapter` (gnn.py:63-82, nodeformer.py:354-388, difformer.py:180-214) called once per position."""
    model.adapter_type = kind
    for position in positions:
        bank = model.pre_adapters if position == "pre" else model.post_adapters
        for layer in range(layers):
            bank[str(layer)] = ga.hydra_compat.instantiate(
                dict(_target_="gconv_adapter_amd.GConvAdapter", bottleneck_size=r, **CFG), hidden_size=d)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "adapter" in k and "scalar" not in k:
                p.normal_(0, 0.1)          # trained-scale adapters (the reference's init is ~identity)
            elif k.endswith("scalar"):
                p.fill_(0.7)
    for name, module in model.named_modules():        # freeze rule (gnn.py:230-241)
        for _, p in module.named_parameters(recurse=False):
            p.requires_grad = ("adapter" in name or any(h in name for h in head_names)
                               or isinstance(module, (nn.LayerNorm, nn.BatchNorm1d)))


def _adapter_params(state, bank, layer):
    prefix = f"{bank}.{layer}."
    return {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}


# ------------------------------------------------------------------------------------------
# molecule shape (gnn.py:97-130)
# ------------------------------------------------------------------------------------------
class MiniGNN(nn.Module):
    D, R, LAYERS = 64, 8, 2

    def __init__(self):
        super().__init__()
        self.gnn_layers = nn.ModuleList([ga.MolecularGINConv(self.D) for _ in range(self.LAYERS)])
        self.batch_norm_layers = nn.ModuleList([ga.BatchNorm1d(self.D) for _ in range(self.LAYERS)])
        self.pre_adapters, self.post_adapters, self.adapter_type = nn.ModuleDict(), nn.ModuleDict(), None
        self.head = nn.Linear(self.D, 1)

    def forward(self, x, edge_index, edge_attr):
        hidden_states = [x]
        for layer in range(self.LAYERS):
            ls = str(layer)
            pre = None
            if ls in self.pre_adapters:
                pre = self.pre_adapters[ls](hidden_states[layer], edge_index, edge_attr)
                if self.adapter_type == "sequential":
                    hidden_states[layer] = pre
            h = self.gnn_layers[layer](hidden_states[layer], edge_index, edge_attr)
            if h.size(0) > 1:
                h = self.batch_norm_layers[layer](h)
            post = None
            if ls in self.post_adapters:
                post = self.post_adapters[ls](h, edge_index, edge_attr)
                if self.adapter_type == "sequential":
                    h = post
            if self.adapter_type == "parallel":
                if pre is not None:
                    h = h + pre
                if post is not None:
                    h = h + post
            if layer != self.LAYERS - 1:
                h = F.relu(h)
            hidden_states.append(h)
        return self.head(hidden_states[-1])


def _oracle_gnn(state, kind, x, ei, ea):
    flags = pyg_restated.AdapterFlags(learnable_scalar=True, normalize=True)
    hs = [x]
    for layer in range(MiniGNN.LAYERS):
        pre = post = None
        ap = _adapter_params(state, "pre_adapters", layer)
        if ap:
            pre = pyg_restated.adapter_forward(hs[layer], ei, ap, flags)
            if kind == "sequential":
                hs[layer] = pre
        gp = {k[len(f"gnn_layers.{layer}."):]: v for k, v in state.items() if k.startswith(f"gnn_layers.{layer}.")}
        h = gin_restated.gin_forward(hs[layer], ei, ea, gp)
        h = F.batch_norm(h, None, None, state[f"batch_norm_layers.{layer}.weight"],
                         state[f"batch_norm_layers.{layer}.bias"], training=True, eps=1e-5)
        ap = _adapter_params(state, "post_adapters", layer)
        if ap:
            post = pyg_restated.adapter_forward(h, ei, ap, flags)
            if kind == "sequential":
                h = post
        if kind == "parallel":
            if pre is not None:
                h = h + pre
            if post is not None:
                h = h + post
        if layer != MiniGNN.LAYERS - 1:
            h = F.relu(h)
        hs.append(h)
    return F.linear(hs[-1], state["head.weight"], state["head.bias"])


def _compare(model, state, want_loss, loss):
    """loss and EVERY trainable gradient element-wise (rtol 1e-5; absolute term relative to the tensor's
    scale for these sums over all nodes, `helpers.grad_atol`)."""
    assert abs(loss.item() - want_loss.item()) <= 1e-5 * max(1.0, abs(want_loss.item()))
    seen = 0
    for k, p in model.named_parameters():
        if not p.requires_grad:
            assert p.grad is None, k
            continue
        seen += 1
        got, want = p.grad.cpu().double(), state[k].grad
        assert_close(got, want, f"grad {k}", atol=grad_atol(want))
    return seen


@pytest.mark.parametrize("positions,kind", MODES)
def test_molecule_backbone_insertion_modes(dev, positions, kind):
    """BatchNorm + ReLU sit between the adapters and the loss: a unit within fp32 rounding of a ReLU kink
    could take the other side on the GPU and move every upstream gradient. Rounds 2-3 therefore compared
    these gradients as vectors (cosine); now the seeded inputs are drawn until the fp64 oracle model has
    NO unit near a kink (`helpers.KinkWatch`: ~10^5 ReLU inputs per model, a few tries) and every
    gradient is compared element-wise."""
    torch.manual_seed(0)
    model = MiniGNN()
    _add_adapters(model, positions, kind, MiniGNN.D, MiniGNN.R, MiniGNN.LAYERS)
    assert len(model.pre_adapters) == (MiniGNN.LAYERS if "pre" in positions else 0)
    assert len(model.post_adapters) == (MiniGNN.LAYERS if "post" in positions else 0)
    model = model.to(dev).train()
    for seed in range(2, 66):
        b = ga.synth.molecule_batch("esol", 24, hidden=MiniGNN.D, seed=seed)
        g = torch.Generator().manual_seed(seed + 1)
        e = b.edge_index.shape[1]
        ea = torch.stack([torch.randint(0, 4, (e,), generator=g), torch.randint(0, 3, (e,), generator=g)], 1)
        target = torch.randn(b.num_nodes, 1, generator=g).double()
        state = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()
                 if v.dtype.is_floating_point and "running" not in k}
        with KinkWatch() as watch:
            want_loss = (_oracle_gnn(state, kind, b.x.double(), b.edge_index, ea) - target).square().mean()
        if watch.risky == 0:
            break
    assert watch.risky == 0 and watch.units > 10_000, "no seeded batch without a unit at a ReLU kink"
    want_loss.backward()
    out = model(b.x.to(dev), b.edge_index.to(dev), ea.to(dev))
    loss = (out.double() - target.to(dev)).square().mean()
    loss.backward()
    n_adapters = MiniGNN.LAYERS * len(positions)
    assert _compare(model, state, want_loss, loss) == 5 * n_adapters + 2 * MiniGNN.LAYERS + 2


# ------------------------------------------------------------------------------------------
# NodeFormer shape (nodeformer.py:403-438): [1, N, d], two-argument adapter calls
# ------------------------------------------------------------------------------------------
class MiniFormer(nn.Module):
    D, R, LAYERS = 32, 16, 2

    def __init__(self):
        super().__init__()
        self.convs = nn.ModuleList([nn.Linear(self.D, self.D) for _ in range(self.LAYERS)])   # frozen stand-in
        self.bns = nn.ModuleList([nn.LayerNorm(self.D) for _ in range(self.LAYERS + 1)])
        self.pre_adapters, self.post_adapters, self.adapter_type = nn.ModuleDict(), nn.ModuleDict(), None
        self.head = nn.Linear(self.D, 3)

    def forward(self, x, adjs):
        z = F.elu(self.bns[0](x))
        layer_ = [z]
        for i, conv in enumerate(self.convs):
            ls = str(i)
            pre = None
            if ls in self.pre_adapters:
                pre = self.pre_adapters[ls](layer_[i], adjs[0])
                if self.adapter_type == "sequential":
                    layer_[i] = pre
            z = conv(layer_[i])
            z += layer_[i]                         # in place, as the reference does (:418)
            z = F.elu(self.bns[i + 1](z))
            post = None
            if ls in self.post_adapters:
                post = self.post_adapters[ls](z, adjs[0])
                if self.adapter_type == "sequential":
                    z = post
            if self.adapter_type == "parallel":
                if pre is not None:
                    z = z + pre
                if post is not None:
                    z = z + post
            layer_.append(z)
        return self.head(layer_[-1]).squeeze(0)


def _oracle_former(state, kind, x, ei):
    flags = pyg_restated.AdapterFlags(learnable_scalar=True, normalize=True)
    D = MiniFormer.D

    def ln(v, i):
        return F.layer_norm(v, (D,), state[f"bns.{i}.weight"], state[f"bns.{i}.bias"], 1e-5)
    z = F.elu(ln(x, 0))
    layer_ = [z]
    for i in range(MiniFormer.LAYERS):
        pre = post = None
        ap = _adapter_params(state, "pre_adapters", i)
        if ap:
            pre = pyg_restated.adapter_forward(layer_[i], ei, ap, flags)
            if kind == "sequential":
                layer_[i] = pre
        z = F.linear(layer_[i], state[f"convs.{i}.weight"], state[f"convs.{i}.bias"]) + layer_[i]
        z = F.elu(ln(z, i + 1))
        ap = _adapter_params(state, "post_adapters", i)
        if ap:
            post = pyg_restated.adapter_forward(z, ei, ap, flags)
            if kind == "sequential":
                z = post
        if kind == "parallel":
            if pre is not None:
                z = z + pre
            if post is not None:
                z = z + post
        layer_.append(z)
    return F.linear(layer_[-1], state["head.weight"], state["head.bias"]).squeeze(0)


@pytest.mark.parametrize("positions,kind", MODES)
def test_nodeformer_shape_insertion_modes(dev, positions, kind):
    torch.manual_seed(1)
    b = ga.synth.planetoid_like(700, 1500, hidden=MiniFormer.D, seed=4)      # [1, N, d], loops pre-added
    assert b.x.dim() == 3
    model = MiniFormer()
    _add_adapters(model, positions, kind, MiniFormer.D, MiniFormer.R, MiniFormer.LAYERS)
    model = model.to(dev).train()
    y = torch.randint(0, 3, (b.num_nodes,), generator=torch.Generator().manual_seed(5))
    state = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()}
    want_loss = F.cross_entropy(_oracle_former(state, kind, b.x.double(), b.edge_index), y)
    want_loss.backward()
    out = model(b.x.to(dev), [b.edge_index.to(dev)])
    loss = F.cross_entropy(out.double(), y.to(dev))
    loss.backward()
    n_adapters = MiniFormer.LAYERS * len(positions)
    # no ReLU / BatchNorm between the adapters and the loss besides the adapters' own (inside the
    # oracle too, on values bounded away from the kink by the 0.1-scale weights): element-wise
    assert _compare(model, state, want_loss, loss) == 5 * n_adapters + 2 * (MiniFormer.LAYERS + 1) + 2


# ------------------------------------------------------------------------------------------
# DIFFormer shape (difformer.py:216-267): [N, d], the conv consumes x, alpha-mix residual, LayerNorm after
# ------------------------------------------------------------------------------------------
class MiniDIFFormer(nn.Module):
    IN, D, R, LAYERS, ALPHA = 24, 32, 16, 2, 0.5

    def __init__(self):
        super().__init__()
        self.fcs = nn.ModuleList([nn.Linear(self.IN, self.D), nn.Linear(self.D, 3)])           # both trainable (:207-208)
        self.convs = nn.ModuleList([nn.Linear(self.D, self.D) for _ in range(self.LAYERS)])    # frozen stand-in for
        self.bns = nn.ModuleList([nn.LayerNorm(self.D) for _ in range(self.LAYERS + 1)])       # DIFFormerConv(x, x, ei, w)
        self.pre_adapters, self.post_adapters, self.adapter_type = nn.ModuleDict(), nn.ModuleDict(), None

    def forward(self, x, adjs):
        layer_ = []
        x = F.relu(self.bns[0](self.fcs[0](x)))             # :221-225 (dropout p = 0)
        layer_.append(x)
        for i, conv in enumerate(self.convs):
            ls = str(i)
            pre = None
            if ls in self.pre_adapters:
                pre = self.pre_adapters[ls](layer_[i], adjs[0])
                if self.adapter_type == "sequential":
                    layer_[i] = pre
            x = conv(x)                                     # x, NOT layer_[i] (:241)
            x = self.ALPHA * x + (1 - self.ALPHA) * layer_[i]   # :244
            x = self.bns[i + 1](x)                          # :246
            post = None
            if ls in self.post_adapters:
                post = self.post_adapters[ls](x, adjs[0])
                if self.adapter_type == "sequential":
                    x = post
            if self.adapter_type == "parallel":
                if pre is not None:
                    x = x + pre
                if post is not None:
                    x = x + post
            layer_.append(x)
        return self.fcs[-1](x)


def _oracle_difformer(state, kind, x, ei):
    flags = pyg_restated.AdapterFlags(learnable_scalar=True, normalize=True)
    D, a = MiniDIFFormer.D, MiniDIFFormer.ALPHA

    def ln(v, i):
        return F.layer_norm(v, (D,), state[f"bns.{i}.weight"], state[f"bns.{i}.bias"], 1e-5)
    x = F.relu(ln(F.linear(x, state["fcs.0.weight"], state["fcs.0.bias"]), 0))
    layer_ = [x]
    for i in range(MiniDIFFormer.LAYERS):
        pre = post = None
        ap = _adapter_params(state, "pre_adapters", i)
        if ap:
            pre = pyg_restated.adapter_forward(layer_[i], ei, ap, flags)
            if kind == "sequential":
                layer_[i] = pre
        x = F.linear(x, state[f"convs.{i}.weight"], state[f"convs.{i}.bias"])
        x = ln(a * x + (1 - a) * layer_[i], i + 1)
        ap = _adapter_params(state, "post_adapters", i)
        if ap:
            post = pyg_restated.adapter_forward(x, ei, ap, flags)
            if kind == "sequential":
                x = post
        if kind == "parallel":
            if pre is not None:
                x = x + pre
            if post is not None:
                x = x + post
        layer_.append(x)
    return F.linear(x, state["fcs.1.weight"], state["fcs.1.bias"])


@pytest.mark.parametrize("positions,kind", MODES)
def test_difformer_shape_insertion_modes(dev, positions, kind):
    """The five insertion modes in DIFFormer's control flow, 2-D input, loops pre-added (`adjs[0]` of
    `scripts/finetune_transductive_learning.py:111-114`). A ReLU sits between the trainable input MLP and the
    adapters: seeds are drawn until the fp64 oracle model has no unit near a kink (`helpers.KinkWatch`)."""
    torch.manual_seed(2)
    model = MiniDIFFormer()
    _add_adapters(model, positions, kind, MiniDIFFormer.D, MiniDIFFormer.R, MiniDIFFormer.LAYERS, head_names=("fc",))
    assert all(p.requires_grad for p in model.fcs.parameters()) and not any(p.requires_grad for p in model.convs.parameters())
    model = model.to(dev).train()
    state = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()}
    for seed in range(3, 67):
        b = ga.synth.planetoid_like(700, 1500, hidden=MiniDIFFormer.IN, seed=seed)
        x = b.x[0]                                           # DIFFormer takes [N, in] (`:216-221`)
        y = torch.randint(0, 3, (b.num_nodes,), generator=torch.Generator().manual_seed(seed + 1))
        for v in state.values():
            v.grad = None
        with KinkWatch() as watch:
            want_loss = F.cross_entropy(_oracle_difformer(state, kind, x.double(), b.edge_index), y)
        if watch.risky == 0:
            break
    assert watch.risky == 0 and watch.units > 10_000, "no seeded graph without a unit at a ReLU kink"
    want_loss.backward()
    out = model(x.to(dev), [b.edge_index.to(dev)])
    assert out.shape == (b.num_nodes, 3)
    loss = F.cross_entropy(out.double(), y.to(dev))
    loss.backward()
    n_adapters = MiniDIFFormer.LAYERS * len(positions)
    assert _compare(model, state, want_loss, loss) == 5 * n_adapters + 2 * (MiniDIFFormer.LAYERS + 1) + 4
