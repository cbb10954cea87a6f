"""Probe: duration of hopmi.log_melspec at the bench's batch size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, hopmi
dev = torch.device("cuda:0")
a = torch.randn(128, 36267, device=dev)
out = hopmi.log_melspec(a); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): hopmi.log_melspec(a)
e1.record(); torch.cuda.synchronize()
print(f"log_melspec(128 clips): {e0.elapsed_time(e1) / 20:.3f} ms per call")
