#!/usr/bin/env python3
"""Diagnostic: device time of one eager train_llm step (BASELINE configs[1] / [2] per GPU) grouped by aten operator and input
shapes (torch.profiler, CPU + device activities).  OPS=sum,copy_ selects operators; DTYPE=bf16 the mixed-precision step."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hopmi
from oracle.golden_util import Accel, step_args
import test_gpu_parity as T
from torch.profiler import profile, ProfilerActivity

epoch = int(os.environ.get("EPOCH", "0"))
want = set(os.environ.get("OPS", "aten::sum").split(","))
dev = torch.device("cuda:0")
if os.environ.get("DTYPE") == "bf16":
    hopmi.mixed_precision("bf16")
torch.manual_seed(0)
m, d, bcfg, inp = T._full_size_setup(9, 128)
m.to(dev).train(); d.to(dev).train()
g_opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=5e-4, betas=(0.5, 0.999))
d_opt = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999))
x = {k: v.to(dev) for k, v in inp.items()}


def step():
    return hopmi.train_llm(step_args(9), epoch, x["in_audio"], x["log_melspec"], x["text"], x["target_dir_vec"], x["vid_indices"],
                           m, d, g_opt, d_opt, Accel())


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()

by = collections.defaultdict(lambda: [0, 0.0])
tot = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    t = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
    self_t = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if ev.name.startswith("aten::") and self_t > 0:
        tot[ev.name][0] += 1
        tot[ev.name][1] += self_t
    if ev.name in want:
        key = (ev.name, str([tuple(s) for s in ev.input_shapes[:2]]) if ev.input_shapes else "")
        by[key][0] += 1
        by[key][1] += t
print("device self time by aten operator (us), top 25:")
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t:9.1f}  {c:5d}  {n}")
print("selected operators by shape:")
for (n, shp), (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"  {t:9.1f}  {c:4d}  {n:14s} {shp}")
