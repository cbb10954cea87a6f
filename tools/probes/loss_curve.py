#!/usr/bin/env python3
"""Diagnostic: per-step losses of eager train_llm (BASELINE configs[1] size, dropout off) with the fused loss op on / off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hopmi
from hopmi import steps
from oracle.golden_util import Accel, step_args
import test_gpu_parity as T

dev = torch.device("cuda:0")
for fused in (True, False):
    steps.FUSED_LOSSES = fused
    torch.manual_seed(0)
    m, d, bcfg, inp = T._full_size_setup(9, 128)
    m.to(dev).train(); d.to(dev).train()
    g_opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=5e-4, betas=(0.5, 0.999))
    d_opt = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999))
    x = {k: v.to(dev) for k, v in inp.items()}
    curve = []
    for it in range(30):
        r = hopmi.train_llm(step_args(9), 0, x["in_audio"], x["log_melspec"], x["text"], x["target_dir_vec"], x["vid_indices"],
                            m, d, g_opt, d_opt, Accel())
        curve.append(round(r["loss"], 3))
    print("fused" if fused else "torch", curve, flush=True)
