import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hopmi import ops
dev = torch.device("cuda:0")
torch.set_printoptions(linewidth=200, precision=3, sci_mode=True)
B, L, H, S, E, qmul = 1, 32, 1, 32, 128, 3.0
g = torch.Generator().manual_seed(1)
q, k, v = torch.randn(B, L, H, E, generator=g) * qmul, torch.randn(S, H, E, generator=g), torch.randn(S, H, E, generator=g)
go = torch.randn(B, L, H, E, generator=g)
scale = 1.0 / E ** 0.5
Q, K, V, dO = q.reshape(-1, E).double(), k[:, 0].double(), v[:, 0].double(), go.reshape(-1, E).double()
Sc = scale * Q @ K.T
P = torch.softmax(Sc, -1)
O = P @ V
delta = (dO * O).sum(-1, keepdim=True)
dP = dO @ V.T
dS = P * (dP - delta)
dbg = int(os.environ["HOPMI_ATTN_DBG"])
qd, kd, vd = (t.detach().clone().to(dev).requires_grad_() for t in (q, k, v))
o = ops.reprog_attention(qd, kd, vd, scale, 0.0, 0)
(o * go.to(dev)).sum().backward()
torch.cuda.synchronize()
got_v = vd.grad[:, 0, :32].cpu().double()      # [key][row]
got_k = kd.grad[:, 0, :32].cpu().double()
want_v = {1: P.T, 2: Sc.T, 3: dP.T}[dbg]
print("dbg", dbg, "max |got - want| (dV slot):", (got_v - want_v).abs().max().item(), "max |want|", want_v.abs().max().item())
print("   dS^T slot err:", (got_k - dS.T).abs().max().item(), "max |dS|", dS.abs().max().item())
err = (got_v - want_v).abs()
i = int(err.argmax())
print("   worst at key,row", i // 32, i % 32, "got", got_v.flatten()[i].item(), "want", want_v.flatten()[i].item())
print("   lse check: o err", (o.detach().cpu().double().reshape(-1, E) - O).abs().max().item())
