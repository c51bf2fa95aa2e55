"""cProfile of the fine-tune step's host side (what the Python thread does to issue one step; no synchronisation):
    python tools/lab/host_prof.py [steps]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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

def step():
    opt.zero_grad(set_to_none=True)
    out = model(x, ei, ea, node_graph, batch.num_graphs, inv_count)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out.view(-1).double(), target)
    loss.backward()
    opt.step()

for _ in range(50):
    step()
torch.cuda.synchronize()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
