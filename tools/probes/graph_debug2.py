"""Debug probe: the new-batches scenario with a second eager model interleaved."""
import sys, os, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hopmi
from hopmi import ops, steps, graph
sys.path.insert(0, "tests")
from test_gpu_graph import _pair
from oracle.golden_util import step_args, Accel

dev = torch.device("cuda:0")
steps._randn_like = lambda t: torch.full_like(t, 0.5)
steps._randperm = lambda n, device: torch.arange(n - 1, -1, -1, device=device)
m1, d1, inp = _pair(9, dev)
m2, d2 = copy.deepcopy(m1), copy.deepcopy(d1)
m2._randn_like = m1._randn_like
mk = lambda m, d: (torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999)),
                   torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999)))
g1, o1 = mk(m1, d1)
g2, o2 = mk(m2, d2)
orig_take = graph._Capture.take_status
words = []
def take(self):
    words.extend(self.status)
    return orig_take(self)
graph._Capture.take_status = take
interleave = len(sys.argv) > 1 and sys.argv[1] == "1"
gs = hopmi.GraphedTrainStep(step_args(9), m2, d2, g2, o2, eager_calls=1)
names = ("in_audio", "log_melspec", "text", "target_dir_vec", "vid_indices")
for it in range(5):
    b = tuple(inp[k] for k in names)
    want = hopmi.train_llm(step_args(9), 0, *b, m1, d1, g1, o1, Accel()) if interleave else None
    t0 = time.perf_counter()
    try:
        r = gs(0, *b)
    except RuntimeError as e:
        r = str(e)[:60]
    torch.cuda.synchronize()
    print(it, f"{time.perf_counter() - t0:.3f}s", r, want, "words", [int(w.item()) for w in words], "host", gs._host[:6].tolist() if gs._built else None, flush=True)
