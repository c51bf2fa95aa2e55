"""configs[3] adapter step, eager against ONE replayed HIP graph (bench.measure_graph_replay; the default bench run starts it as `bench.py --graph-replay` in a child process)."""
import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import gconv_adapter_amd as ga
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
ga._lib.load()
for n in ("cora", "pubmed"):
    print(bench.measure_graph_replay(dev, n), flush=True)
