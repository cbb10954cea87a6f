#!/usr/bin/env python3
"""Diagnostic: hopmi_colsum against torch's column sum on the bias-gradient shapes of a configs[1] step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import hopmi
from hopmi import ops

dev = torch.device("cuda:0")


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for dt in (torch.float32, torch.bfloat16):
    for M, N in [(4352, 2100), (4352, 768), (4352, 175), (4608, 173), (1500, 1024), (2048, 1700), (4352, 1024), (4608, 512), (18432, 64), (4352, 27)]:
        x = torch.randn(M, N, device=dev).to(dt)
        t_h = timeit(lambda: ops.colsum(x))
        t_t = timeit(lambda: x.sum(0))
        gb = x.numel() * x.element_size() / 1e9
        print(f"{str(dt):15s} {M:6d} x {N:5d}  hopmi {t_h:7.1f} us ({gb / t_h * 1e6:6.0f} GB/s)   torch {t_t:7.1f} us", flush=True)
