"""Restatement of the fine-tune step's readout and loss — TEST INFRASTRUCTURE ONLY
(see ``oracle/__init__.py``).

* ``global_pool``: ``self.pool(h, batch)`` of the reference's graph-level model
  (``/root/reference/src/models/inductive/gnn.py:187-195``: ``global_add_pool`` / ``global_mean_pool``).
  The functions live in torch-geometric (PARITY UNPINNED as for the adapter); their published semantics:
  ``scatter(x, batch, dim=0, dim_size=size, reduce='sum' | 'mean')`` — row ``g`` of the result is the sum
  (mean) of the rows ``i`` with ``batch[i] == g``; a graph without nodes gives zeros (the mean divides by
  ``max(count, 1)``).
* ``masked_loss``: the training loop's loss, restated line by line from
  ``/root/reference/scripts/finetune_inductive_learning.py``:
    ``:50``     ``target = batch.y.view(output.shape).to(torch.float64)``
    ``:56-58``  ``valid_mask = ~torch.isnan(target)``; ``output[valid_mask]``, ``target[valid_mask]``
    ``:61-63``  ``loss_matrix = criterion(valid_output.double(), valid_target)``;
                ``loss = torch.sum(loss_matrix) / len(valid_target)``
  with ``criterion = BCEWithLogitsLoss(reduction='none')`` (classification) or ``MSELoss(reduction='none')``
  (regression). The reference skips the batch when no target is valid (``:60``); here that case is loss 0.

Plain torch on CPU tensors, autograd for the gradients.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def global_pool(x: torch.Tensor, batch: torch.Tensor, size: int, mean: bool) -> torch.Tensor:
    out = torch.zeros(size, x.shape[1], dtype=x.dtype).index_add_(0, batch, x)
    if mean:
        count = torch.zeros(size, dtype=x.dtype).index_add_(0, batch, torch.ones(batch.shape[0], dtype=x.dtype))
        out = out / count.clamp(min=1).unsqueeze(1)
    return out


def masked_loss(output: torch.Tensor, target: torch.Tensor, kind: str) -> torch.Tensor:
    target = target.view(output.shape).to(torch.float64)
    valid = ~torch.isnan(target)
    vo, vt = output[valid], target[valid]
    if len(vt) == 0:
        return output.sum().double() * 0.0
    if kind == "bce":
        m = F.binary_cross_entropy_with_logits(vo.double(), vt, reduction="none")
    elif kind == "mse":
        m = F.mse_loss(vo.double(), vt, reduction="none")
    else:
        raise ValueError(kind)
    return torch.sum(m) / len(vt)
