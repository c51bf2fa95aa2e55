"""Generates the golden fixtures under tests/golden/ from the fp64 oracle.

    python tests/golden/make_golden.py

The reference holds no golden vectors and its hot path (PyG GCNConv) cannot be
imported here, so these fixtures come from the oracle's restatement (PARITY
UNPINNED, see oracle/__init__.py). Inputs are stored in float32 (exactly what
the HIP path receives), expected outputs in float64. Weights are at trained
scale (std 0.1, non-zero biases): at the reference's init scale the adapter is
a near-identity and the comparison would be vacuous.

Fixtures:
  g1_path5_{norm,raw}.npz   N=5 path, one isolated node, one duplicate edge, one
                            pre-existing self-loop, one directed-only edge
  g2_molecules.npz          3 collated molecules, d=300, r=4 (inductive shape)
  g3_transductive.npz       x [1, N, 32], r=16, self-loops pre-added
  g4_flags.npz              36 flag combinations (skip x scalar x normalization x act)
"""
from __future__ import annotations

import itertools
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyg_restated as oracle  # noqa: E402

PARAM_KEYS = ["conv_down.lin.weight", "conv_down.bias", "conv_up.lin.weight", "conv_up.bias",
              "scalar", "normalization.weight", "normalization.bias"]


def f32(t: torch.Tensor) -> torch.Tensor:
    """Round to float32 and come back to float64: the oracle sees exactly the
    values the fp32 HIP path will see."""
    return t.float().double()


def run_case(x32: torch.Tensor, ei: torch.Tensor, flags: oracle.AdapterFlags, d: int, r: int,
             seed: int, prefix: str = "") -> dict:
    params = {k: f32(v) for k, v in oracle.init_params(d, r, flags, seed=seed).items()}
    g = torch.Generator().manual_seed(seed + 1000)
    grad_out = torch.randn(x32.shape, generator=g).float()
    out, grads = oracle.adapter_forward_backward(x32.double(), ei, params, flags, grad_out.double())
    rec = {prefix + "x": x32.numpy(), prefix + "edge_index": ei.numpy(),
           prefix + "grad_out": grad_out.numpy(), prefix + "out": out.numpy(),
           prefix + "grad.x": grads["x"].numpy()}
    for k, v in params.items():
        rec[prefix + "param." + k] = v.float().numpy()
        rec[prefix + "grad." + k] = grads[k].numpy()
    rec[prefix + "flags"] = np.array([flags.non_linearity, flags.normalization,
                                      str(flags.learnable_scalar), str(flags.skip_connection),
                                      str(flags.normalize)])
    return rec


def g1_edges() -> torch.Tensor:
    # path 0-1-2-3 (both directions), node 4 isolated, duplicate 1->2, self-loop on 2,
    # directed-only 3->0
    pairs = [(0, 1), (1, 0), (1, 2), (2, 1), (2, 3), (3, 2), (1, 2), (2, 2), (3, 0)]
    return torch.tensor(pairs, dtype=torch.long).t().contiguous()


def main() -> None:
    gen = torch.Generator().manual_seed(1234)

    # G1
    for normalize, tag in ((True, "norm"), (False, "raw")):
        x = torch.randn(5, 8, generator=gen).float()
        flags = oracle.AdapterFlags(learnable_scalar=True, normalize=normalize)
        np.savez(os.path.join(HERE, f"g1_path5_{tag}.npz"),
                 **run_case(x, g1_edges(), flags, d=8, r=3, seed=11))

    # G2: three small molecules collated into one block-diagonal batch
    sys.path.insert(0, ROOT)
    import gconv_adapter_amd as ga
    sizes = torch.tensor([4, 3, 5])
    ei, _ = ga.synth.molecule_edges(sizes, gen)
    x = torch.randn(int(sizes.sum()), 300, generator=gen).float()
    flags = oracle.AdapterFlags(learnable_scalar=True, normalize=False)  # YAML default
    np.savez(os.path.join(HERE, "g2_molecules.npz"), **run_case(x, ei, flags, d=300, r=4, seed=22))

    # G3: transductive shape [1, N, 32], loops already in the edge list
    n = 40
    u = torch.randint(n, (70,), generator=gen)
    v = torch.randint(n, (70,), generator=gen)
    keep = u != v
    u, v = u[keep], v[keep]
    key = torch.unique(torch.cat([u * n + v, v * n + u]))
    loops = torch.arange(n)
    ei = torch.stack([torch.cat([key // n, loops]), torch.cat([key % n, loops])])
    x = torch.randn(1, n, 32, generator=gen).float()
    flags = oracle.AdapterFlags(learnable_scalar=True, normalize=True)
    np.savez(os.path.join(HERE, "g3_transductive.npz"), **run_case(x, ei, flags, d=32, r=16, seed=33))

    # G4: the flag grid of ablations 1-5
    n = 12
    sizes = torch.tensor([4, 5, 3])
    ei, _ = ga.synth.molecule_edges(sizes, gen)
    cases = []
    for i, (skip, scalar, norm, act) in enumerate(itertools.product(
            (True, False), (True, False), ("none", "layer_norm", "batch_norm"),
            ("relu", "silu", "none"))):
        x = torch.randn(n, 16, generator=gen).float()
        flags = oracle.AdapterFlags(non_linearity=act, normalization=norm, learnable_scalar=scalar,
                                    skip_connection=skip, normalize=bool(i % 2))
        cases.append(run_case(x, ei, flags, d=16, r=4, seed=100 + i))
    # same shapes in every case: stack along a leading case axis (absent params -> zeros)
    keys = sorted({k for c in cases for k in c})
    rec = {}
    for k in keys:
        proto = next(c[k] for c in cases if k in c)
        rec[k] = np.stack([c.get(k, np.zeros_like(proto)) for c in cases])
    np.savez(os.path.join(HERE, "g4_flags.npz"), **rec)

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
