#!/usr/bin/env python3
"""Diagnostic: which generator parameters does rounding decide?  Four eager Adam steps of the test model, run twice, the
second time with the mapping-layer weights perturbed by 1e-7 relative (what computing the prototypes in two halves does):
per-parameter mean |difference| afterwards, and the relative change / sign flips of the last step's gradient."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hopmi
from hopmi import steps
from oracle.golden_util import Accel, step_args
from test_gpu_graph import _pair

epoch = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda:0")
steps._randn_like = lambda t: torch.full_like(t, 0.5)
steps._randperm = lambda n, device: torch.arange(n - 1, -1, -1, device=device)
runs = []
for rep in range(2):
    torch.manual_seed(0)
    m, d, inp = _pair(9, dev)
    if rep == 1:
        with torch.no_grad():
            m.mapping_layer.weight.mul_(1.0 + 1e-7 * torch.sign(torch.randn_like(m.mapping_layer.weight)))
    g_opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999))
    d_opt = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999))
    for it in range(4):
        hopmi.train_llm(step_args(9), epoch, inp["in_audio"], inp["log_melspec"], inp["text"], inp["target_dir_vec"],
                        inp["vid_indices"], m, d, g_opt, d_opt, Accel())
    runs.append(({n: p.detach().clone() for n, p in m.named_parameters()},
                 {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
rows = []
for n in runs[0][1]:
    a, b = runs[0][1][n], runs[1][1][n]
    pa, pb = runs[0][0][n], runs[1][0][n]
    rows.append(((pa - pb).abs().mean().item(), (a - b).abs().mean().item() / max(a.abs().mean().item(), 1e-30), a.abs().mean().item(),
                 (torch.sign(a) != torch.sign(b)).float().mean().item(), n))
for r in sorted(rows, reverse=True)[:14]:
    print("param mean diff %.2e | last grad: rel change %.2e  mean|g| %.2e  sign flips %.3f  %s" % r)
