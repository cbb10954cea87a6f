#!/usr/bin/env python3
"""ops.linear on hopmi_gemm_f16x2 (trainable weights: GRU input projections, align layer, beat MLP, reprogramming projections, the
graph-wavenet block's 1x1 convolutions) against the library's fp32 GEMM at the shapes of BASELINE.json configs[1] (TED, B = 128) and
configs[3] (TED-Expressive, B = 64): error against float64 and time per call -- forward (GEMM alone / + the row-scales pass), the two
weight images (per optimizer step), dX."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hopmi
from hopmi import ops

dev = torch.device("cuda:0")
hopmi.use_tuned_gemms()
g = torch.Generator().manual_seed(0)
SHAPES = {"ted": [(4352, 2100, 700), (4352, 2100, 992), (4352, 768, 1536), (4352, 768, 1024), (4352, 1024, 128), (1500, 1024, 768),
                  (2048, 1700, 3400), (2048, 170, 1700), (4608, 256, 512), (4608, 512, 256), (4608, 173, 512)],
          "expr": [(2176, 2100, 700), (2176, 768, 1536), (2176, 768, 1024), (2176, 1024, 128), (1024, 1700, 3400), (1024, 170, 1700),
                   (10752, 256, 512), (10752, 512, 256), (10752, 173, 512)]}


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for tag in (sys.argv[1:] or ["ted", "expr"]):
    for M, N, K in SHAPES[tag]:
        x = torch.randn(M, K, generator=g).to(dev)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        gy = torch.randn(M, N, generator=g).to(dev)
        ref = x.double() @ w.double().t() + b.double()
        refdx = gy.double() @ w.double()
        rel = lambda a, r: ((a.double() - r).abs().max() / r.abs().max()).item()
        lib = torch.nn.functional.linear(x, w, b)
        t_lib = timed(lambda: torch.nn.functional.linear(x, w, b))
        t_libdx = timed(lambda: gy @ w)
        img = ops.f16_weight_image(w)
        sc = ops.row_scales(x)
        y = ops._split_gemm(x, img, b, N, K, 16, a_part=sc)
        t_g = timed(lambda: ops._split_gemm(x, img, b, N, K, 16, a_part=sc))
        t_rs = timed(lambda: ops.row_scales(x))
        L = hopmi._lib.lib()
        t_img = timed(lambda: ops.f16_weight_image(w))
        line = (f"{tag} M={M} N={N} K={K}: fwd lib {t_lib:6.1f} us err {rel(lib, ref):.1e} | f16x2 {t_g:6.1f} (+rs {t_rs:4.1f}, image {t_img:4.1f}) err {rel(y, ref):.1e}")
        if N % 4 == 0 and K >= 128:
            imgt = ops.f16_weight_image(w, transpose=True)
            scg = ops.row_scales(gy)
            dx = ops._split_gemm(gy, imgt, None, K, N, 16, a_part=scg)
            t_dx = timed(lambda: ops._split_gemm(gy, imgt, None, K, N, 16, a_part=scg))
            t_imgt = timed(lambda: ops.f16_weight_image(w, transpose=True))
            line += f" | dX lib {t_libdx:6.1f} err {rel(gy @ w, refdx):.1e} | f16x2 {t_dx:6.1f} (image^T {t_imgt:4.1f}) err {rel(dx, refdx):.1e}"
        # the weight gradient dW = dY^T X: library vs hopmi_gemm_f16x2_tn (the row scales of gy / x are at hand from the products above)
        refdw = gy.double().t() @ x.double()
        t_libdw = timed(lambda: gy.t() @ x)
        if N % 4 == 0 and K % 4 == 0:
            scg = ops.row_scales(gy)
            dw = ops.f16_mm_tn(gy, x, scg, sc)
            t_dw = timed(lambda: ops.f16_mm_tn(gy, x, scg, sc))
            line += f" | dW lib {t_libdw:6.1f} err {rel(gy.t() @ x, refdw):.1e} | f16x2_tn {t_dw:6.1f} err {rel(dw, refdw):.1e}"
        print(line, flush=True)
