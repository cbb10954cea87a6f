#!/usr/bin/env python3
"""Debug: fused WaveNet-stack backward vs the oracle's autograd over a range of batch sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hopmi
from oracle import fill, ref_cpu, spec
dev = torch.device("cuda:0")
V = int(os.environ.get("V", "9"))
for B in [int(b) for b in os.environ.get("BS", "2,8,16,29,37,64,128").split(",")]:
    m = hopmi.gwnet(None, V, dropout=0, supports=None, gcn_bool=True, addaptadj=True, aptinit=None, in_dim=173,
                    out_dim=173, residual_channels=64, dilation_channels=64, skip_channels=256, end_channels=512)
    fill.fill_state_(m)
    m.to(dev).train()
    x0 = fill.uniform("gwnet.x0", (B, 173, V, 16))
    gout = fill.uniform("gwnet.gout", (B, 173, V, 4))
    xg = x0.to(dev).requires_grad_()
    out = m(xg)
    (out * gout.to(dev)).sum().backward()
    sd = spec.build_sd(spec.gwnet_spec(V, prefix=""))
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    xo = x0.clone().requires_grad_()
    want, upd = ref_cpu.gwnet_forward(sd, xo, prefix="", training=True)
    (want * gout).sum().backward()
    rel = lambda a, b: ((a.detach().cpu().double() - b.detach().double()).abs().max() / b.detach().double().abs().max().clamp_min(1e-30)).item()
    worst = sorted(((rel(p.grad, sd[n].grad), n) for n, p in m.named_parameters() if sd[n].grad is not None and not n.endswith("mlp.mlp.bias")), reverse=True)[:4]
    print(f"B={B:4d} out {rel(out, want):.1e} dx0 {rel(xg.grad, xo.grad):.1e} worst param grads: " + ", ".join(f"{n} {e:.1e}" for e, n in worst), flush=True)

# ---- which side is off?  compare both fp32 results with a float64 evaluation of the oracle
if os.environ.get("F64", "0") == "1":
    for B in [int(b) for b in os.environ.get("BS", "29,64,128").split(",")]:
        m = hopmi.gwnet(None, V, dropout=0, supports=None, gcn_bool=True, addaptadj=True, aptinit=None, in_dim=173,
                        out_dim=173, residual_channels=64, dilation_channels=64, skip_channels=256, end_channels=512)
        fill.fill_state_(m)
        m.to(dev).train()
        x0 = fill.uniform("gwnet.x0", (B, 173, V, 16))
        gout = fill.uniform("gwnet.gout", (B, 173, V, 4))
        xg = x0.to(dev).requires_grad_()
        (m(xg) * gout.to(dev)).sum().backward()
        res = {}
        for dt in (torch.float32, torch.float64):
            sd = {k: (v.detach().clone().to(dt) if v.is_floating_point() else v) for k, v in spec.build_sd(spec.gwnet_spec(V, prefix="")).items()}
            for v in sd.values():
                if v.is_floating_point():
                    v.requires_grad_(True)
            xo = x0.detach().clone().to(dt).requires_grad_()
            want, _ = ref_cpu.gwnet_forward(sd, xo, prefix="", training=True)
            (want * gout.to(dt)).sum().backward()
            res[dt] = (xo.grad, sd["end_conv_1.weight"].grad)
        rel = lambda a, b: ((a.detach().cpu().double() - b.detach().double()).abs().max() / b.detach().double().abs().max()).item()
        print(f"B={B}: dx0  gpu32-vs-f64 {rel(xg.grad, res[torch.float64][0]):.1e}  cpu32-vs-f64 {rel(res[torch.float32][0], res[torch.float64][0]):.1e} | "
              f"end_conv_1.w gpu32-vs-f64 {rel(m.end_conv_1.weight.grad, res[torch.float64][1]):.1e} cpu32-vs-f64 {rel(res[torch.float32][1], res[torch.float64][1]):.1e}", flush=True)
