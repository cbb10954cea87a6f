#!/usr/bin/env python3
"""hopmi_gemm_f16x2 (fp16 hi/lo, three terms) and hopmi_gemm_split (bf16 parts: six / three terms) against frozen weights vs the
library's fp32 GEMM at the frozen BERT's shapes: time per call (HIP events over back-to-back launches) and error against a float64
product.  The f16x2 column is the GEMM with the row scales given; `+rs` adds hopmi_row_scales on the A operand (what a call pays
when no producer wrote the scales)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hopmi
from hopmi import ops

dev = torch.device("cuda:0")
if "--tuned" in sys.argv:
    hopmi.use_tuned_gemms()
g = torch.Generator().manual_seed(0)
M = int(sys.argv[sys.argv.index('--m') + 1]) if '--m' in sys.argv else 4352
for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304)):
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    ref = (x.double() @ w.double().t() + b.double())
    def timed(fn, iters=50):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    t_lib = timed(lambda: torch.nn.functional.linear(x, w, b))
    err_lib = ((torch.nn.functional.linear(x, w, b).double() - ref).abs().max() / ref.abs().max()).item()
    line = f"M={M} N={N} K={K}: library fp32 {t_lib:7.1f} us ({2*M*N*K/t_lib/1e6:6.1f} TF) err {err_lib:.1e}"
    for parts in (16, 3, 2):
        img = ops.split_weight_image(w, parts)
        if parts == 16:
            sc = ops.row_scales(x)
            t = timed(lambda: ops._split_gemm(x, img, b, N, K, parts, a_part=sc))
            t_rs = timed(lambda: ops.row_scales(x))
        else:
            t = timed(lambda: ops._split_gemm(x, img, b, N, K, parts))
        err = ((ops._split_gemm(x, img, b, N, K, parts).double() - ref).abs().max() / ref.abs().max()).item()
        line += f" | parts={parts}: {t:7.1f} us ({2*M*N*K/t/1e6:6.1f} TF-equiv) err {err:.1e}" + (f" (+rs {t_rs:4.1f} us)" if parts == 16 else "")
        if "--ab" in sys.argv and parts != 16:
            t_pre = timed(lambda: ops.split_rows_image(x, parts))
            a_img = ops.split_rows_image(x, parts)
            t_ab = timed(lambda: ops._split_gemm_ab(a_img, M, img, b, N, K, parts))
            err = ((ops._split_gemm_ab(a_img, M, img, b, N, K, parts).double() - ref).abs().max() / ref.abs().max()).item()
            line += f" [ab: split {t_pre:5.1f} + gemm {t_ab:6.1f} us, err {err:.1e}]"
    print(line, flush=True)
