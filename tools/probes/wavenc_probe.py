import sys, os
sys.path.insert(0, os.getcwd())
import torch, hopmi
from hopmi.model import WavEncoder
from oracle import fill
torch.manual_seed(0)
m = WavEncoder()
for n, p in m.named_parameters():
    p.data.copy_(fill.fill_value("audio_encoder." + n, p.shape) if hasattr(fill, "fill_value") else p.data)
x = fill.hot_path_inputs(2, 9, 100, 11)["in_audio"]
g = torch.randn(2, 34 + 0, 32) if False else None
res = {}
for dev in ("cpu", "cuda"):
    mm = __import__("copy").deepcopy(m).to(dev).train()
    xx = x.to(dev)
    o = mm(xx)
    if g is None:
        g = torch.randn(o.shape)
    (o * g.to(dev)).sum().backward()
    res[dev] = (o.detach().cpu(), {n: p.grad.cpu() for n, p in mm.named_parameters()})
    # per-layer activations
    h = xx.unsqueeze(1); acts = []
    with torch.no_grad():
        for layer in mm.feat_extractor:
            h = layer(h) if not isinstance(layer, torch.nn.LeakyReLU) else torch.nn.functional.leaky_relu(h, 0.3)
            acts.append(h.cpu())
    res[dev + "_acts"] = acts
print("out rel", ((res["cpu"][0] - res["cuda"][0]).abs().max() / res["cpu"][0].abs().max()).item())
for n in res["cpu"][1]:
    a, b = res["cpu"][1][n], res["cuda"][1][n]
    print(f"{n:28s} |cpu|={a.abs().sum():.5f} |gpu|={b.abs().sum():.5f} maxdiff/max={((a-b).abs().max()/a.abs().max().clamp_min(1e-30)).item():.2e}")
for i, (a, b) in enumerate(zip(res["cpu_acts"], res["cuda_acts"])):
    print(i, "act rel", ((a - b).abs().max() / a.abs().max()).item(), "near-zero frac", ((a.abs() < 1e-4).float().mean()).item())
