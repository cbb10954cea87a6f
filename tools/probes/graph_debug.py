"""Debug probe: per-iteration timing and GRU status words of a GraphedTrainStep."""
import sys, os, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hopmi
from hopmi import ops, steps, graph
sys.path.insert(0, "tests")
from test_gpu_graph import _pair
from oracle.golden_util import step_args

epoch = int(sys.argv[1]) if len(sys.argv) > 1 else 11
dev = torch.device("cuda:0")
steps._randn_like = lambda t: torch.full_like(t, 0.5)
steps._randperm = lambda n, device: torch.arange(n - 1, -1, -1, device=device)
m, d, inp = _pair(9, dev)
g = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999))
o = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999))
batch = (inp["in_audio"], inp["log_melspec"], inp["text"], inp["target_dir_vec"], inp["vid_indices"])
# keep every status word for inspection
orig_take = graph._Capture.take_status
words = []
def take(self):
    words.extend(self.status)
    return orig_take(self)
graph._Capture.take_status = take
gs = hopmi.GraphedTrainStep(step_args(9), m, d, g, o, eager_calls=1)
for it in range(5):
    t0 = time.perf_counter()
    try:
        r = gs(epoch, *batch)
    except RuntimeError as e:
        r = str(e)[:60]
    torch.cuda.synchronize()
    print(it, f"{time.perf_counter() - t0:.3f}s", r, "words", [int(w.item()) for w in words], "bwd", (float(gs._bwd_status.item()) if gs._built else None), flush=True)
