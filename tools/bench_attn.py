#!/usr/bin/env python3
"""Micro-benchmark of the reprogramming-attention kernels at the BASELINE.json shape
(B=128 -> N = 4352 query rows, S = 1500 prototypes, 8 heads x 128)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hopmi
from hopmi import ops

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=128)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--p", type=float, default=0.1)
a = ap.parse_args()
dev = torch.device("cuda:0")
q = torch.randn(a.B, 34, 8, 128, device=dev, requires_grad=True)
k = torch.randn(1500, 8, 128, device=dev, requires_grad=True)
v = torch.randn(1500, 8, 128, device=dev, requires_grad=True)
go = torch.randn(a.B, 34, 8, 128, device=dev)
for _ in range(3):
    o = ops.reprog_attention(q, k, v, 128 ** -0.5, a.p, 7)
    torch.autograd.grad(o, [q, k, v], go)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
e[0].record()
for _ in range(a.iters):
    o = ops.reprog_attention(q, k, v, 128 ** -0.5, a.p, 7)
e[1].record()
for _ in range(a.iters):
    torch.autograd.grad(o, [q, k, v], go, retain_graph=True)
e[2].record()
torch.cuda.synchronize()
N = a.B * 34
fl = 4 * N * 8 * 1500 * 128
print(f"fwd {1e3 * e[0].elapsed_time(e[1]) / a.iters:8.1f} us  {fl / (e[0].elapsed_time(e[1]) / a.iters * 1e-3) / 1e12:6.1f} TFLOP/s")
print(f"bwd {1e3 * e[1].elapsed_time(e[2]) / a.iters:8.1f} us  {3.5 * fl / (e[1].elapsed_time(e[2]) / a.iters * 1e-3) / 1e12:6.1f} TFLOP/s")
