import os, sys
sys.path.insert(0, "/root/repo")
import torch, hopmi
hopmi.use_tuned_gemms()
dev = torch.device("cuda:0")
M, K, N = 1500, 30522, 768
W = torch.randn(M, K, device=dev) / 170; E = torch.randn(K, N, device=dev); b = torch.randn(M, device=dev)
def timed(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for ks in (1, 2, 3, 6):
    kc = K // ks
    if ks == 1:
        f = lambda: torch.addmm(b.unsqueeze(1), W, E)
    else:
        f = lambda: torch.bmm(W.view(M, ks, kc).transpose(0, 1), E.view(ks, kc, N)).sum(0).add_(b.unsqueeze(1))
    print(ks, f"{timed(f):.1f} us")
