"""Which tensor operations (by operator and input shapes) make up the small-launch tail of one eager train_llm step?"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import trace_aten as ta
dev = torch.device("cuda:0")
step = ta.make_step(dev)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key.startswith("aten::") and e.self_device_time_total > 0:
        rows.append((e.self_device_time_total / 3.0, e.count / 3.0, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"aten self device time per step: {tot:.0f} us")
for t, c, k, sh in rows[:45]:
    print(f"{t:8.1f} us {c:5.1f}x  {k:28s} {sh}")
