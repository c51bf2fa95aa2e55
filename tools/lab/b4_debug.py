"""Which rows / outputs of the S-arxiv backward differ from the oracle (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gconv_adapter_amd as ga
from oracle import pyg_restated as oracle
from tests.helpers import build_module, run_module
dev = torch.device("cuda:0")
batch, cfg = ga.synth.make_workload("arxiv")
d, r = cfg["d"], cfg["r"]
flags = oracle.AdapterFlags(non_linearity="relu", normalization="none", learnable_scalar=True, skip_connection=True, normalize=True)
params = {k: v.float().double() for k, v in oracle.init_params(d, r, flags, seed=0).items()}
g = torch.randn(batch.x.shape, generator=torch.Generator().manual_seed(1))
want_out, want = oracle.adapter_forward_backward(batch.x.double(), batch.edge_index, params, flags, g.double(), need_dx=True)
mod = build_module(ga, d, r, flags, params, dev)
got = run_module(mod, batch.x, batch.edge_index, g, dev, need_dx=True)
print("out err", (got["out"] - want_out).abs().max().item())
for k, w in want.items():
    e = (got[k] - w.double()).abs()
    print(k, "max err", e.max().item(), "max |want|", w.abs().max().item())
e = (got["x"] - want["x"].double()).abs().amax(dim=1)
bad = (e > 1e-3).nonzero().flatten()
print("bad rows", bad.numel(), "of", e.numel(), "first", bad[:20].tolist(), "last", bad[-5:].tolist())
ok = (e <= 1e-3).nonzero().flatten()
print("ok rows", ok[:40].tolist())
ec = (got["x"] - want["x"].double()).abs().amax(dim=0)
print("bad cols", (ec > 1e-3).nonzero().flatten()[:20].tolist(), "n", int((ec > 1e-3).sum()))
row = int(bad[0]) if bad.numel() else 0
print("row", row, "got", got["x"][row, :8].tolist(), "want", want["x"][row, :8].tolist())
