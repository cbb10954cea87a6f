"""Small-case error map of the reprogramming attention backward (probe)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hopmi import ops
dev = torch.device("cuda:0")
torch.set_printoptions(linewidth=220, precision=2, sci_mode=True)


def run(B, L, H, S, seed, qmul=1.0):
    E = 128
    g = torch.Generator().manual_seed(seed)
    q, k, v = torch.randn(B, L, H, E, generator=g) * qmul, torch.randn(S, H, E, generator=g), torch.randn(S, H, E, generator=g)
    go = torch.randn(B, L, H, E, generator=g)
    scale = 1.0 / E ** 0.5

    def attn(cast):
        qq, kk, vv = (t.detach().clone().to(cast).requires_grad_() for t in (q, k, v))
        p = torch.softmax(scale * torch.einsum("blhe,she->bhls", qq, kk), dim=-1)
        o = torch.einsum("bhls,she->blhe", p, vv)
        (o * go.to(cast)).sum().backward()
        return o.detach(), qq.grad, kk.grad, vv.grad, p.detach()

    r64, r32 = attn(torch.float64), attn(torch.float32)
    qd, kd, vd = (t.detach().clone().to(dev).requires_grad_() for t in (q, k, v))
    o = ops.reprog_attention(qd, kd, vd, scale, 0.0, 0)
    (o * go.to(dev)).sum().backward()
    torch.cuda.synchronize()
    print(f"--- B={B} L={L} H={H} S={S} qmul={qmul}   max P {r64[4].max().item():.3f}")
    for name, got, a32, a64 in zip(("o", "dq", "dk", "dv"), (o, qd.grad, kd.grad, vd.grad), r32, r64):
        err = (got.detach().cpu().double() - a64).abs()
        e32 = (a32.double() - a64).abs()
        mx = a64.abs().max().item()
        print(f"{name}: dev {err.max().item() / mx:.3e} fp32 {e32.max().item() / mx:.3e}")
        if name in ("dk", "dv") and S <= 64 and H == 1:
            per_key = (err.amax(dim=2) / mx).flatten()
            print("   per key:", per_key)
            kmax = int(per_key.argmax())
            print(f"   key {kmax} per e (first 32):", (err[kmax, 0] / mx)[:32])


run(1, 32, 1, 32, 1)
run(1, 32, 1, 32, 1, qmul=4.0)
run(2, 34, 1, 64, 2, qmul=3.0)
run(16, 34, 8, 1500, 23)
