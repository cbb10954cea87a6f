#!/usr/bin/env python3
"""Diagnostic: after 5 steps of train_llm (eager) and GraphedTrainStep on copies of one model, which parameters / buffers differ."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hopmi
from hopmi import steps
from oracle.golden_util import Accel, step_args
import test_gpu_graph as T

V, epoch = int(os.environ.get("V", "9")), int(os.environ.get("EPOCH", "11"))
dev = torch.device("cuda:0")
steps._randn_like = lambda t: torch.full_like(t, 0.5)
steps._randperm = lambda n, device: torch.arange(n - 1, -1, -1, device=device)
m1, d1, inp = T._pair(V, dev)
m2, d2 = copy.deepcopy(m1), copy.deepcopy(d1)
m2._randn_like = m1._randn_like
mk = lambda m, d: (torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999)),
                   torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999)))
g1, o1 = mk(m1, d1)
g2, o2 = mk(m2, d2)
args = step_args(V)
batch = (inp["in_audio"], inp["log_melspec"], inp["text"], inp["target_dir_vec"], inp["vid_indices"])
graphed = hopmi.GraphedTrainStep(args, m2, d2, g2, o2, eager_calls=2)
for it in range(5):
    want = hopmi.train_llm(args, epoch, *batch, m1, d1, g1, o1, Accel())
    got = graphed(epoch, *batch)
    print(it, {k: (round(want[k], 6), round(got[k], 6)) for k in want}, flush=True)
    torch.cuda.synchronize()
    rows = []
    for (n, a), (_, b) in zip(list(m1.named_parameters()) + list(d1.named_parameters()),
                              list(m2.named_parameters()) + list(d2.named_parameters())):
        df = (a.detach() - b.detach()).abs()      # (detached: a live autograd node here would be reused by the capture)
        rows.append((df.max().item(), df.mean().item(), n, a.numel()))
    rows.sort(reverse=True)
    print("   params:", [(f"{r[0]:.1e}", f"{r[1]:.1e}", r[2], r[3]) for r in rows[:6]])
    rows = []
    for (n, a), (_, b) in zip(m1.named_buffers(), m2.named_buffers()):
        if a.is_floating_point():
            rows.append(((a - b).abs().max().item() / max(a.abs().max().item(), 1.0), n))
    rows.sort(reverse=True)
    print("   buffers:", [(f"{r[0]:.1e}", r[1]) for r in rows[:4]])
