import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, hopmi
from hopmi import ops
from oracle import fill, ref_cpu, spec
dev = torch.device("cuda:0")
B, S, d_llm, p_drop = 128, 1500, 768, 0.1
m = hopmi.ReprogrammingLayer(128, 8, 128, d_llm, attention_dropout=p_drop); fill.fill_state_(m); m.to(dev).train()
tgt = fill.normal("reprog.target", (B, 34, 128)); src = fill.uniform("reprog.source", (S, d_llm), 0.5); gout = fill.uniform("reprog.gout", (B, 34, d_llm))
tg, sg = tgt.to(dev).requires_grad_(), src.to(dev).requires_grad_()
hopmi.ReprogrammingLayer._calls = 100
relu_in = []
h = m.activation.register_forward_hook(lambda mod, inp, outp: relu_in.append((inp[0] > 0).detach().cpu().float()))
out = m(tg, sg, sg); h.remove(); (out * gout.to(dev)).sum().backward()
seed = (torch.initial_seed() * 2654435761 + 101 * 40503) & 0xFFFFFFFF
mask = ops.attn_keep_mask(seed, B * 34, 8, S, p_drop, "cpu").view(B, 34, 8, S).permute(0, 2, 1, 3).float()
res = {}
for dt in (torch.float32, torch.float64):
    sd = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in spec.build_sd(spec.reprog_spec(d_llm, prefix="")).items()}
    to, so = tgt.clone().to(dt).requires_grad_(), src.clone().to(dt).requires_grad_()
    want = ref_cpu.reprogramming_layer(sd, to, so, so, 8, prefix="", drop_mask=mask.to(dt), p_drop=p_drop, relu_mask=relu_in[0].reshape(B, 34, -1).to(dt))
    (want * gout.to(dt)).sum().backward()
    res[dt] = (want.detach(), to.grad, so.grad)
rel = lambda a, b: ((a.detach().cpu().double() - b.double()).abs().max() / b.double().abs().max()).item()
f64 = res[torch.float64]; f32 = res[torch.float32]
print("out   gpu-vs-f64 %.2e  cpu32-vs-f64 %.2e" % (rel(out, f64[0]), rel(f32[0], f64[0])))
print("dtgt  gpu-vs-f64 %.2e  cpu32-vs-f64 %.2e" % (rel(tg.grad, f64[1]), rel(f32[1], f64[1])))
print("dsrc  gpu-vs-f64 %.2e  cpu32-vs-f64 %.2e" % (rel(sg.grad, f64[2]), rel(f32[2], f64[2])))
d = (tg.grad.cpu().double() - f64[1]).abs().view(B * 34, -1).max(1).values / f64[1].abs().max()
print("rows with err > 1e-4:", int((d > 1e-4).sum()), "of", d.numel(), "worst rows", d.topk(5).indices.tolist(), d.topk(5).values.tolist())
