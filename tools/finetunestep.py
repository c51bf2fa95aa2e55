"""The whole fine-tune step of bench.py's extras on its own (for rocprofv3 --kernel-trace --stats):
    python tools/finetunestep.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
bench.pin_host_threads()
print(bench.measure_finetune_step(dev, True, steps=int(sys.argv[1]) if len(sys.argv) > 1 else 100))
