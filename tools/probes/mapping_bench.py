import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, hopmi
from hopmi.model import _SplitKAffine
if os.environ.get("TUNED", "1") == "1":
    hopmi.use_tuned_gemms()
dev = torch.device("cuda:0")
W = torch.randn(1500, 30522, device=dev, requires_grad=True); E = torch.randn(30522, 768, device=dev); b = torch.randn(1500, device=dev, requires_grad=True)
g = torch.randn(1500, 768, device=dev)
def old():
    part = torch.bmm(W.view(1500, 6, 5087).transpose(0, 1), E.view(6, 5087, 768)); S = part.sum(0) + b.unsqueeze(1); S.backward(g)
def new():
    _SplitKAffine.apply(W, E, b, 6).backward(g)
for name, fn in (("old", old), ("new", new)):
    for _ in range(5): fn(); W.grad = None; b.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn(); W.grad = None; b.grad = None
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t0) / 20 * 1e3, "ms")
