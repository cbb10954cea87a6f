import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from torch.profiler import ProfilerActivity, profile
import trace_aten as ta
dev = torch.device("cuda:0")
step = ta.make_step(dev)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
seen = set()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::cat", "aten::copy_") and e.input_shapes and e.input_shapes[0] and len(e.input_shapes[0]) and (e.name == "aten::cat" or (torch.tensor(e.input_shapes[0]).prod().item() if e.input_shapes[0] else 0) >= 1000000):
        st = [s for s in (e.stack or []) if "hop-hetero" in s or "hopmi" in s or "steps.py" in s][:3]
        key = (e.name, str(e.input_shapes)[:60], tuple(st))
        if key in seen: continue
        seen.add(key)
        print(e.name, str(e.input_shapes)[:70], "|", " <- ".join(s.split("/")[-1][:60] for s in st))
