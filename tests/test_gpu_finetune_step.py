"""The caller of the hot path in miniature, on the GPU: a MolecularGNN-shaped
model assembled from this package's parts — pre-adapters inserted the reference's way
(`gnn.py:63-82,102-105`), frozen GIN layers, BatchNorm, mean pool, prediction head — trained with
the reference's loop body (`scripts/finetune_inductive_learning.py:43-67`: NaN-masked BCE in fp64,
backward, Adam) and its freeze rule (`gnn.py:230-241`). Checked against the same model built from
the CPU oracles in fp64."""
from __future__ import annotations

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import gconv_adapter_amd as ga
from oracle import gin_restated, pyg_restated

from helpers import KinkWatch, assert_close, grad_atol

pytestmark = pytest.mark.gpu

D, R, LAYERS, TASKS = 64, 8, 3, 2


class MiniMolecularGNN(nn.Module):
    """`MolecularGNN.forward` (`gnn.py:85-130`, sequential pre-adapters) + mean pool + head."""

    def __init__(self):
        super().__init__()
        self.gnn_layers = nn.ModuleList([ga.MolecularGINConv(D) for _ in range(LAYERS)])
        self.batch_norm_layers = nn.ModuleList([ga.BatchNorm1d(D, fused_relu=i != LAYERS - 1) for i in range(LAYERS)])
        self.pre_adapters = nn.ModuleDict()
        self.graph_pred_linear = nn.Linear(D, TASKS)

    def add_adapter(self):   # gnn.py:63-82 with the YAML of configs/finetune/gconv_adapter.yaml
        cfg = dict(bottleneck_size=R, conv_type="gcn", learnable_scalar=True, non_linearity="relu",
                   normalization="none", normalize=True, skip_connection=True)
        for layer in range(LAYERS):
            self.pre_adapters[str(layer)] = ga.GConvAdapter(hidden_size=D, **cfg)
        for name, module in self.named_modules():   # freeze rule, gnn.py:230-241
            for _, p in module.named_parameters(recurse=False):
                p.requires_grad = ("adapter" in name or "graph_pred_linear" in name
                                   or isinstance(module, (nn.LayerNorm, nn.BatchNorm1d)))

    def forward(self, x, edge_index, edge_attr, batch, num_graphs):
        h = x
        for layer in range(LAYERS):
            h = self.pre_adapters[str(layer)](h, edge_index, edge_attr)
            h = self.gnn_layers[layer](h, edge_index, edge_attr)
            h = self.batch_norm_layers[layer](h)      # ReLU of the inner layers folded in (gnn.py:127-128)
        pooled = torch.zeros(num_graphs, D, device=h.device, dtype=h.dtype).index_add_(0, batch, h)
        pooled = pooled / torch.bincount(batch, minlength=num_graphs).clamp(min=1).unsqueeze(1)
        return self.graph_pred_linear(pooled)


def _masked_loss(output, target):   # finetune_inductive_learning.py:50-63
    valid = ~torch.isnan(target)
    loss_matrix = F.binary_cross_entropy_with_logits(output[valid].double(), target[valid], reduction="none")
    return loss_matrix.sum() / valid.sum()


def _oracle_forward(state, x, ei, ea, batch, num_graphs):
    flags = pyg_restated.AdapterFlags(learnable_scalar=True, normalize=True)
    h = x
    for layer in range(LAYERS):
        ap = {k[len(f"pre_adapters.{layer}."):]: v for k, v in state.items() if k.startswith(f"pre_adapters.{layer}.")}
        h = pyg_restated.adapter_forward(h, ei, ap, flags)
        gp = {k[len(f"gnn_layers.{layer}."):]: v for k, v in state.items() if k.startswith(f"gnn_layers.{layer}.")}
        h = gin_restated.gin_forward(h, ei, ea, gp)
        w, b = state[f"batch_norm_layers.{layer}.weight"], state[f"batch_norm_layers.{layer}.bias"]
        h = F.batch_norm(h, None, None, w, b, training=True, eps=1e-5)
        if layer != LAYERS - 1:
            h = F.relu(h)
    pooled = torch.zeros(num_graphs, D, dtype=h.dtype).index_add_(0, batch, h)
    pooled = pooled / torch.bincount(batch, minlength=num_graphs).clamp(min=1).unsqueeze(1)
    return F.linear(pooled, state["graph_pred_linear.weight"], state["graph_pred_linear.bias"])


def _setup(dev, seed=2):
    torch.manual_seed(0)
    b = ga.synth.molecule_batch("esol", 24, hidden=D, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    e = b.edge_index.shape[1]
    ea = torch.stack([torch.randint(0, 4, (e,), generator=g), torch.randint(0, 3, (e,), generator=g)], 1)
    sizes = b.graph_ptr[1:] - b.graph_ptr[:-1]
    batch = torch.repeat_interleave(torch.arange(b.num_graphs), sizes)
    y = torch.randint(0, 2, (b.num_graphs, TASKS), generator=g).double()
    y[torch.rand(y.shape, generator=g) < 0.2] = float("nan")          # missing labels, as in MoleculeNet
    model = MiniMolecularGNN()
    model.add_adapter()
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "adapter" in k and "scalar" not in k:
                p.normal_(0, 0.1)                                       # trained-scale adapters
    return b, ea, batch, y, model.to(dev)


def test_trainable_set_follows_the_reference_freeze_rule(dev):
    *_, model = _setup(dev)
    names = {k for k, p in model.named_parameters() if p.requires_grad}
    assert all(("adapter" in k) or k.startswith("graph_pred_linear") or k.startswith("batch_norm_layers")
               for k in names)
    assert not any(k.startswith("gnn_layers") for k in names)
    per_adapter = 2 * R * D + R + D + 1
    assert sum(p.numel() for k, p in model.named_parameters() if "adapter" in k) == LAYERS * per_adapter


def test_first_step_matches_the_fp64_oracle_model(dev):
    """Loss and every trainable gradient ELEMENT-WISE at 1e-5 (rounds 2-3: as vectors, by cosine —
    BatchNorm + ReLU sit between the adapters and the loss and a unit at a ReLU kink may take the other
    side in fp32). The seeded batch is drawn until the fp64 oracle model has no unit near a kink
    (`helpers.KinkWatch`), so a difference beyond tolerance is a real one."""
    for seed in range(2, 66):
        b, ea, batch, y, model = _setup(dev, seed)
        state = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()
                 if v.dtype.is_floating_point and "running" not in k}
        with KinkWatch() as watch:
            want_loss = _masked_loss(_oracle_forward(state, b.x.double(), b.edge_index, ea, batch, b.num_graphs), y)
        if watch.risky == 0:
            break
    assert watch.risky == 0 and watch.units > 10_000, "no seeded batch without a unit at a ReLU kink"
    want_loss.backward()
    model.train()
    out = model(b.x.to(dev), b.edge_index.to(dev), ea.to(dev), batch.to(dev), b.num_graphs)
    loss = _masked_loss(out, y.to(dev))
    loss.backward()
    assert abs(loss.item() - want_loss.item()) <= 1e-5 * max(1.0, abs(want_loss.item()))
    for k, p in model.named_parameters():
        if not p.requires_grad:
            assert p.grad is None, k
            continue
        want = state[k].grad
        assert_close(p.grad.cpu().double(), want, f"grad {k}", atol=grad_atol(want))


def test_loop_body_trains_and_leaves_the_backbone_untouched(dev):
    b, ea, batch, y, model = _setup(dev)
    x, ei, ea, batch, y = b.x.to(dev), b.edge_index.to(dev), ea.to(dev), batch.to(dev), y.to(dev)
    frozen_before = {k: v.clone() for k, v in model.state_dict().items() if k.startswith("gnn_layers")}
    opt = ga.FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-2)
    model.train()
    losses = []
    for _ in range(40):
        opt.zero_grad()
        try:
            output = model(x, ei, ea, batch, b.num_graphs)
        except RuntimeError:      # the reference's loop skips such a batch (:51-53)
            continue
        loss = _masked_loss(output, y)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert len(losses) == 40 and losses[-1] < 0.6 * losses[0], losses[::8]
    for k, v in model.state_dict().items():
        if k.startswith("gnn_layers"):
            assert torch.equal(v, frozen_before[k]), k
    for k, p in model.named_parameters():   # FusedAdam moved them into its flat buffer, 16-byte aligned
        if p.requires_grad:
            assert p.data_ptr() % 16 == 0, k
    with pytest.raises(RuntimeError):   # what the loop's `except RuntimeError` relies on
        model(x[:, :D - 4].contiguous(), ei, ea, batch, b.num_graphs)
