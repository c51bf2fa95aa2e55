"""Host time per phase of the fine-tune step of bench.py's extras (no synchronisation inside the
step: what the Python thread spends issuing work):  python tools/finetunehost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import gconv_adapter_amd as ga

dev = torch.device("cuda:0")
bench.pin_host_threads()
batch, cfg = ga.synth.make_workload("hiv")
d, r, e = cfg["d"], cfg["r"], batch.edge_index.shape[1]
g = torch.Generator().manual_seed(0)
ea = torch.stack([torch.randint(0, 4, (e,), generator=g), torch.randint(0, 3, (e,), generator=g)], 1).to(dev)
sizes = batch.graph_ptr[1:] - batch.graph_ptr[:-1]
node_graph = torch.repeat_interleave(torch.arange(batch.num_graphs), sizes).to(dev)
inv_count = (1.0 / sizes.clamp(min=1).float()).unsqueeze(1).to(dev)
model = bench._FinetuneModel(d, r, cfg["adapters"], 1, True).to(dev)
opt = ga.FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
x, ei = batch.x.to(dev), batch.edge_index.to(dev)
target = torch.randint(0, 2, (batch.num_graphs,), generator=g).double().to(dev)
sub = {}
def _timed(mod, name):
    orig = mod.forward
    def f(*a, **k):
        t = time.perf_counter(); out = orig(*a, **k); sub[name] = sub.get(name, 0.0) + time.perf_counter() - t
        return out
    mod.forward = f
for i in range(cfg["adapters"]):
    _timed(model.pre_adapters[str(i)], "adapters"); _timed(model.gnn_layers[i], "gin"); _timed(model.batch_norm_layers[i], "bn")
acc = {"zero_grad": 0.0, "forward": 0.0, "loss": 0.0, "backward": 0.0, "optimizer": 0.0}
steps = 100
for it in range(steps + 20):
    if it == 20:
        torch.cuda.synchronize()
        acc = {k: 0.0 for k in acc}; sub.clear()
        t_all = time.perf_counter()
    t0 = time.perf_counter(); opt.zero_grad(set_to_none=True)
    t1 = time.perf_counter(); out = model(x, ei, ea, node_graph, batch.num_graphs, inv_count)
    t2 = time.perf_counter()
    loss = ga.masked_loss(out, target.view(out.shape), "bce")
    t3 = time.perf_counter(); loss.backward()
    t4 = time.perf_counter(); opt.step()
    t5 = time.perf_counter()
    for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        acc[k] += v
    if it % 10 == 9:
        torch.cuda.synchronize()     # keep the queue shallow: host time, not back-pressure
torch.cuda.synchronize()
print({k: f"{v / steps * 1e6:.0f} us" for k, v in acc.items()}, f"sum {sum(acc.values()) / steps * 1e6:.0f} us")
print("inside forward:", {k: f"{v / steps * 1e6:.0f} us" for k, v in sub.items()}, f"wall per step {(time.perf_counter() - t_all) / steps * 1e6:.0f} us")
