#!/usr/bin/env python3
"""Micro-benchmark of the fused WaveNet-layer forward kernel (hopmi_wn_layer_fwd) at the BASELINE.json shapes.

    python tools/bench_wn.py [--V 9 --B 128] [--iters 200] [--saves]

Per layer i (T_in = 16,15,13,12,10,9,7,6; dilation 1,2,...): (a) the dispatch's own begin/end timestamps
(hopmi_time_next_launch, median over --iters launches: what rocprofv3's kernel trace reports and bench.py's roofline
uses) and (b) back-to-back launches between two HIP events (includes the ~1.5 us dependent-launch boundary); reports
algorithmic GB/s (x in + y out + skip tail) and fp32-equivalent TFLOP/s.  HOPMI_WN_GRID / HOPMI_WN_MAXMT
select the tile geometry for experiments."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import hopmi
from hopmi import ops

DIL = (1, 2, 1, 2, 1, 2, 1, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--V", type=int, default=9)
    ap.add_argument("--B", type=int, default=128)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--saves", action="store_true", help="also write the gate values (diagnostic output)")
    ap.add_argument("--warm", type=float, default=2.0)
    ap.add_argument("--lib", default=None, help="alternative libhopmi.so (A/B runs)")
    a = ap.parse_args()
    if a.lib:
        hopmi._lib._LIB_PATH = os.path.abspath(a.lib)
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    A = torch.softmax(torch.randn(a.V, a.V, generator=g), 1).to(dev)
    prep = ops.gcn_prepare(A, A @ A)
    wf = (torch.randn(64, 64, 1, 2, generator=g) / 11).to(dev)
    wg = (torch.randn(64, 64, 1, 2, generator=g) / 11).to(dev)
    bf, bg = torch.randn(64, generator=g).to(dev), torch.randn(64, generator=g).to(dev)
    Wm = (torch.randn(64, 192, generator=g) / 14).to(dev)
    bm = torch.randn(64, generator=g).to(dev)
    scsh = torch.cat([torch.ones(64), torch.zeros(64)]).to(dev)
    wa = torch.randn(8192, 8192, device=dev)
    t_end = time.time() + a.warm
    while time.time() < t_end:
        for _ in range(10):
            wa @ wa
        torch.cuda.synchronize()
    L = hopmi._lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    wimg = ops.wn_prepare_weights([(wf, wg, Wm)])[0]
    tot_t = tot_b = tot_f = tot_k = 0.0
    T_in = 16
    for li, d in enumerate(DIL):
        T_out = T_in - d
        do_gcn = 1
        x = torch.randn(a.B, T_in, a.V, 64, generator=g).to(dev)
        y = torch.empty(a.B, T_out, a.V, 64, device=dev)
        fs = torch.empty(a.B, T_out, a.V, 128, device=dev) if a.saves else None
        tails = torch.empty(a.B, 4, a.V, 512, device=dev)
        ut = tails[..., 64 * li:64 * li + 64]
        ws = torch.empty(L.hopmi_wn_layer_ws_floats(a.B, T_in, a.V, d), device=dev)
        fn = lambda: L.hopmi_wn_layer_fwd(x.data_ptr(), scsh.data_ptr(), wimg.data_ptr(), bf.data_ptr(), bg.data_ptr(), prep.data_ptr(),
                                          bm.data_ptr(), y.data_ptr(), fs.data_ptr() if a.saves else None,
                                          ut.data_ptr(), ut.stride(2), ws.data_ptr(), a.B, T_in, a.V, d, do_gcn, st)
        n_out = a.B * T_out * a.V
        nbytes = 4 * 64 * (a.B * T_in * a.V + n_out + 4 * a.B * a.V)
        flops = n_out * (2 * 2 * 2 * 64 * 64 + 2 * 192 * 64 + 4 * 64 * a.V)
        for _ in range(10):
            hopmi._lib.check(fn(), "wn_layer_fwd")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        durs = []
        for _ in range(min(a.iters, 50)):
            k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            k0.record(); k1.record()
            L.hopmi_time_next_launch(k0.cuda_event, k1.cuda_event)
            fn()
            durs.append((k0, k1))
        torch.cuda.synchronize()
        ds = sorted(k0.elapsed_time(k1) * 1e3 for k0, k1 in durs)
        kus = ds[len(ds) // 2]
        print(f"layer {li} T_in={T_in:2d} d={d}: kernel {kus:6.2f} us ({nbytes / kus / 1e3:7.1f} GB/s, {flops / kus / 1e6:6.1f} TF)   "
              f"back-to-back {us:6.2f} us", flush=True)
        tot_t += us; tot_b += nbytes; tot_f += flops; tot_k += kus
        T_in = T_out
    print(f"stack: kernels {tot_k:.1f} us = {tot_b / tot_k / 1e3:.1f} GB/s algorithmic ({tot_b / tot_k / 8e6 * 100:.1f} % of 8 TB/s), "
          f"{tot_f / tot_k / 1e6:.1f} TF fp32-equivalent; back-to-back {tot_t:.1f} us")


if __name__ == "__main__":
    main()
