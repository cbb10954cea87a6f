import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hopmi import ops
dev = torch.device("cuda:0")


def run(B, L, H, S, seed, qmul=1.0, tag=""):
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
        return o.detach(), qq.grad, kk.grad, vv.grad

    r64, r32 = attn(torch.float64), attn(torch.float32)
    qd, kd, vd = (t.detach().clone().to(dev).requires_grad_() for t in (q, k, v))
    o = ops.reprog_attention(qd, kd, vd, scale, 0.0, 0)
    (o * go.to(dev)).sum().backward()
    torch.cuda.synchronize()
    out = []
    for name, got, a32, a64 in zip(("o", "dq", "dk", "dv"), (o, qd.grad, kd.grad, vd.grad), r32, r64):
        err = (got.detach().cpu().double() - a64).abs().max().item()
        e32 = (a32.double() - a64).abs().max().item()
        out.append(f"{name} {err / e32:6.2f}")
    print(f"{tag} B={B} L={L} H={H} S={S} qmul={qmul}: ratios to fp32  " + "  ".join(out), flush=True)


tag = os.environ.get("TAG", "")
run(1, 32, 1, 32, 1, 3.0, tag)
run(1, 64, 1, 32, 1, 3.0, tag)
run(1, 96, 1, 32, 1, 3.0, tag)
run(1, 128, 1, 32, 1, 3.0, tag)
run(2, 34, 1, 64, 2, 3.0, tag)
run(16, 34, 8, 1500, 23, 1.0, tag)
