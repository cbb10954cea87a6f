#!/usr/bin/env python3
"""Diagnostic: which ingredient of tests/test_gpu_graph.py::test_graphed_step_baseline_size_vs_oracle breaks the recording?
   python tools/probes/graph_capture_bisect.py draws|nodraws hook|nohook tuned|notuned fused|nofused [V B epoch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hopmi
from hopmi import steps
from oracle.golden_util import step_args
from test_gpu_parity import _full_size_setup
from test_gpu_graph import _Draws

flags = set(sys.argv[1:5])
V, B, epoch = (int(x) for x in (sys.argv[5:8] if len(sys.argv) >= 8 else (9, 128, 0)))
dev = torch.device("cuda:0")
m, d, bcfg, inp = _full_size_setup(V, B)
m.to(dev).train(); d.to(dev).train()
class MP:
    def setattr(self, obj, name, val): setattr(obj, name, val)
draws = _Draws(B, V, epoch > 10, dev)
if "draws" in flags:
    draws.install(m, MP())
else:
    m._randn_like = lambda t: torch.full_like(t, 0.25)
    steps._randn_like = lambda t: torch.full_like(t, 0.5)
    steps._randperm = lambda n, device: torch.arange(n - 1, -1, -1, device=device)
fused = "fused" in flags
g_opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999), fused=fused)
d_opt = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999), fused=fused)
gin = {k: v.to(dev) for k, v in inp.items()}
batch = (gin["in_audio"], gin["log_melspec"], gin["text"], gin["target_dir_vec"], gin["vid_indices"])
graded = []
if "hook" in flags:
    m.register_forward_hook(lambda mod, args, out: graded.append(out[0]) if torch.is_grad_enabled() else None)
if "tuned" in flags:
    assert hopmi.use_tuned_gemms()
graphed = hopmi.GraphedTrainStep(step_args(V), m, d, g_opt, d_opt, eager_calls=1)
torch.manual_seed(777)
for it in range(3):
    draws.refill()
    print(it, graphed(epoch, *batch), flush=True)
torch.cuda.synchronize()
print("OK", sorted(flags), "replays", graphed.n_replay)
