"""Device-side collation (MoleculeStore.collate) vs CPU collation + H2D copy of the same batch:
512 random molecules out of an HIV-sized resident dataset, d = 300."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 41127
t0 = time.time(); whole = ga.synth.molecule_batch("hiv", G, hidden=300, seed=0); t1 = time.time()
store = ga.MoleculeStore.from_batch(whole, dev); torch.cuda.synchronize(); t2 = time.time()
print(f"dataset: {G} molecules, {whole.num_nodes} nodes, {whole.edge_index.shape[1]} edges, "
      f"{whole.x.numel() * 4 / 1e9:.2f} GB of features; generated in {t1 - t0:.1f} s, stored in {t2 - t1:.1f} s")
ga.dist.pin_to_local_cores()
rng = np.random.default_rng(0)
def dev_collate():
    return store.collate(rng.choice(G, 512, replace=False))
for _ in range(20): dev_collate()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); a.record()
for _ in range(200): dev_collate()
b.record(); host = (time.perf_counter() - t0) / 200 * 1e6; b.synchronize()
print(f"device collation of 512 molecules: host {host:.0f} us, stream time {a.elapsed_time(b) / 200 * 1e3:.0f} us per batch")
# CPU: what Batch.from_data_list + .to(device) amounts to (per-graph tensors prepared beforehand)
ptr = whole.graph_ptr; src, dst = whole.edge_index
gid = torch.bucketize(dst, ptr[1:], right=True); order = torch.sort(gid, stable=True).indices
src, dst, gid = src[order], dst[order], gid[order]
eptr = torch.zeros(G + 1, dtype=torch.long); eptr[1:] = torch.bincount(gid, minlength=G).cumsum(0)
xs = [whole.x[int(ptr[k]):int(ptr[k + 1])] for k in range(G)]
es = [torch.stack([src[int(eptr[k]):int(eptr[k + 1])], dst[int(eptr[k]):int(eptr[k + 1])]]) - int(ptr[k]) for k in range(G)]
def cpu_collate():
    ids = rng.choice(G, 512, replace=False)
    off, xl, el = 0, [], []
    for k in ids:
        xl.append(xs[k]); el.append(es[k] + off); off += xs[k].shape[0]
    return torch.cat(xl).to(dev, non_blocking=True), torch.cat(el, dim=1).to(dev, non_blocking=True)
for _ in range(5): cpu_collate()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): cpu_collate()
torch.cuda.synchronize()
print(f"CPU collation + H2D of the same batch: {(time.perf_counter() - t0) / 50 * 1e6:.0f} us per batch (1 thread, pageable memory)")
