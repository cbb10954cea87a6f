"""How far torch's fused (capturable) Adam over the generator's parameter list is from a bandwidth floor: the same update on the
real list (143 tensors, 65.7 M elements), on ONE flat tensor of the same size, and a plain copy of 7 x the bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hopmi
from hopmi import synth
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = hopmi.Model(synth.model_configs("TED"), synth.build_bert(6), synth.SyntheticTokenizer(), synth.SpeakerVocab(1370)).float().to(dev)
ps = [p for p in model.parameters() if p.requires_grad]
n = sum(p.numel() for p in ps)
for p in ps:
    p.grad = torch.randn_like(p) * 1e-3


def timed(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


opt = torch.optim.Adam(ps, lr=1e-2, betas=(0.5, 0.999), fused=True, capturable=True)
t_list = timed(opt.step)
flat = torch.nn.Parameter(torch.randn(n, device=dev))
flat.grad = torch.randn(n, device=dev) * 1e-3
opt1 = torch.optim.Adam([flat], lr=1e-2, betas=(0.5, 0.999), fused=True, capturable=True)
t_flat = timed(opt1.step)
big = [p for p in ps if p.numel() > 1_000_000]
opt2 = torch.optim.Adam(big, lr=1e-2, betas=(0.5, 0.999), fused=True, capturable=True)
t_big = timed(opt2.step)
src = torch.randn(n * 4, device=dev)[: n * 4]
dst = torch.empty(n * 3, device=dev)
t_copy = timed(lambda: (dst.copy_(src[: n * 3]), src[n * 3:].sum()))
print(f"{len(ps)} tensors, {n / 1e6:.1f} M elements, 7 x {4 * n / 1e6:.0f} MB per step")
print(f"fused Adam, the parameter list : {t_list:7.1f} us  ({7 * 4 * n / t_list / 1e6:.2f} TB/s)")
print(f"fused Adam, one flat tensor    : {t_flat:7.1f} us  ({7 * 4 * n / t_flat / 1e6:.2f} TB/s)")
print(f"fused Adam, the {len(big)} tensors > 1 M elements ({sum(p.numel() for p in big) / 1e6:.1f} M): {t_big:7.1f} us")
print(f"copy 3 n + read n (a 4 n read / 3 n write stream): {t_copy:7.1f} us")
