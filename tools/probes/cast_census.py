#!/usr/bin/env python3
"""Diagnostic: where the dtype-cast launches of one bf16 train_llm step (BASELINE configs[2] per GPU) come from.
torch.profiler over one eager step; aten::_to_copy / aten::copy_ events grouped by (shape, dtypes) and by the innermost
frame inside the package."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hopmi
from oracle.golden_util import Accel, step_args
import test_gpu_parity as T
from torch.profiler import profile, ProfilerActivity

epoch = int(os.environ.get("EPOCH", "0"))
dev = torch.device("cuda:0")
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
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()

by = collections.Counter()
for ev in prof.events():
    if ev.name not in ("aten::_to_copy", "aten::contiguous", "aten::clone"):
        continue
    if ev.name != "aten::_to_copy" and not ev.cpu_children:
        continue
    frame = "?"
    for fr in ev.stack or []:
        if "_amd/" in fr or "hopmi" in fr:
            frame = fr.split("_amd/")[-1][:70]
            break
    else:
        frame = (ev.stack[0][-70:] if ev.stack else "(backward / no stack)")
    shp = tuple(ev.input_shapes[0]) if ev.input_shapes else ()
    by[(ev.name, shp, frame)] += 1
tot = sum(v for (n, _, _), v in by.items() if n == "aten::_to_copy")
print(f"aten::_to_copy events in one step: {tot}")
for (name, shp, frame), c in sorted(by.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{c:4d}  {name:16s} {str(shp):28s} {frame}")
