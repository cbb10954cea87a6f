"""Where does the reprogramming attention's backward differ from float64?  (probe; prints per-tensor error and the worst elements)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hopmi import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(23)
B, L, H, E, S = 16, 34, 8, 128, 1500
q, k, v = torch.randn(B, L, H, E, generator=g), torch.randn(S, H, E, generator=g), torch.randn(S, H, E, generator=g)
go = torch.randn(B, L, H, E, generator=g)
scale = 1.0 / E ** 0.5


def attn(cast):
    qq, kk, vv = (t.detach().clone().to(cast).requires_grad_() for t in (q, k, v))
    p = torch.softmax(scale * torch.einsum("blhe,she->bhls", qq, kk), dim=-1)
    o = torch.einsum("bhls,she->blhe", p, vv)
    (o * go.to(cast)).sum().backward()
    return o.detach(), qq.grad, kk.grad, vv.grad


ref64, ref32 = attn(torch.float64), attn(torch.float32)
qd, kd, vd = (t.detach().clone().to(dev).requires_grad_() for t in (q, k, v))
o = ops.reprog_attention(qd, kd, vd, scale, 0.0, 0)
(o * go.to(dev)).sum().backward()
torch.cuda.synchronize()
for name, got, r32, r64 in zip(("o", "dq", "dk", "dv"), (o, qd.grad, kd.grad, vd.grad), ref32, ref64):
    got = got.detach().cpu().double()
    err = (got - r64).abs()
    e32 = (r32.double() - r64).abs()
    mx = r64.abs().max().item()
    print(f"{name}: dev {err.max().item() / mx:.3e}  fp32 {e32.max().item() / mx:.3e}  ratio {err.max().item() / e32.max().item():.2f}  "
          f"rms dev {err.pow(2).mean().sqrt().item() / mx:.3e} rms fp32 {e32.pow(2).mean().sqrt().item() / mx:.3e}")
    flat = err.flatten()
    top = torch.topk(flat, 5)
    for val, idx in zip(top.values.tolist(), top.indices.tolist()):
        coords = list(torch.unravel_index(torch.tensor(idx), err.shape))
        print("     worst", [int(c) for c in coords], f"err {val:.3e} ref {r64.flatten()[idx].item():.4e}")
    if name in ("dk", "dv"):
        per_key = err.amax(dim=(1, 2))
        bad = (per_key > 20 * e32.max()).nonzero().flatten().tolist()
        print("     keys with error > 20 x fp32's max:", bad[:40], "count", len(bad))

# ---- GRU: which of dx / dW errors are the recurrence kernel's and which the GEMMs around it? (default vs strict forms)
import copy
import hopmi
torch.manual_seed(5)
Bg, T, I, Hh, Lyr = 128, 34, 992, 350, 2
gru = torch.nn.GRU(I, Hh, num_layers=Lyr, batch_first=True, bidirectional=True)
x, gy = torch.randn(Bg, T, I), torch.randn(Bg, T, 2 * Hh)
names = ("weight_hh_l0", "weight_hh_l1_reverse", "weight_ih_l1")


def run(mod, xx, gg):
    xx = xx.clone().requires_grad_()
    y, _ = mod(xx)
    (y * gg).sum().backward()
    return [y.detach(), xx.grad] + [mod.get_parameter(n).grad for n in names]


r32, r64 = run(copy.deepcopy(gru), x, gy), run(copy.deepcopy(gru).double(), x.double(), gy.double())
for strict in (False, True):
    prev = hopmi.strict_fp32(strict)
    gd = copy.deepcopy(gru).to(dev)
    xd = x.to(dev).requires_grad_()
    yd = ops.gru_bidirectional(xd, gd)
    (yd * gy.to(dev)).sum().backward()
    torch.cuda.synchronize()
    got = [yd, xd.grad] + [gd.get_parameter(n).grad for n in names]
    for name, a, b32, b64 in zip(("y", "dx") + names, got, r32, r64):
        mx = b64.abs().max().item()
        e = (a.detach().cpu().double() - b64).abs().max().item() / mx
        e32 = (b32.double() - b64).abs().max().item() / mx
        print(f"gru strict={strict} {name}: dev {e:.3e} fp32 {e32:.3e} ratio {e / e32:.2f}")
    hopmi.strict_fp32(prev)
