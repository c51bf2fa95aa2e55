"""The fine-tune step of bench.py's extras alone (HIP events, best of 3 averages):
    python tools/finetunestep.py [steps] [workload=hiv|hiv4096|esol]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
bench.pin_host_threads()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
name = sys.argv[2] if len(sys.argv) > 2 else "hiv"
print(bench.measure_finetune_step(dev, True, name=name, steps=steps))
