#!/usr/bin/env python3
"""Diagnostic: what ANY single launch moving the graded kernel's algorithmic bytes achieves on this device -- a plain copy
(read n bytes, write n bytes) at the sizes of the fused WaveNet layer's launches (7.08 MB moved on average at V=9, B=128;
16.4 MB at V=42, B=64) and larger.  Run under rocprofv3 --kernel-trace --stats to read the kernels' own durations."""
import os, sys
import torch

dev = torch.device("cuda:0")
for mb in (3.54, 8.2, 32.0, 256.0):          # bytes read = bytes written = mb
    n = int(mb * 1e6 / 4)
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    for _ in range(5):
        y.copy_(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        y.copy_(x)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 200 * 1e3
    print(f"copy {mb:7.2f} MB in + {mb:7.2f} MB out: {us:7.2f} us per launch back to back = {2 * mb / us * 1e-3 * 1e3:7.0f} GB/s "
          f"= {2 * mb / us / 8e3 * 1e3:.3f} of 8 TB/s", flush=True)
