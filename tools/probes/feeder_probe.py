"""Probe: host-side cost of HostFeeder.refill() and next() at the bench's batch size."""
import sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, hopmi
from hopmi import synth
dev = torch.device("cuda:0")
hb = []
for k in range(3):
    b = synth.synthetic_batch(128, 9, 4321 + k, "cpu")
    hb.append(dict(audio_padded=b["in_audio"], text_token_padded=b["text"].double(), vec_seq=b["target_dir_vec"], vid_indices=b["vid_indices"]))
f = hopmi.HostFeeder(itertools.cycle(hb), dev)
x = torch.randn(8192, 8192, device=dev)
for it in range(8):
    t0 = time.perf_counter(); b = next(f); t1 = time.perf_counter()
    for _ in range(6): x @ x            # ~20 ms of GPU work standing in for the step
    t2 = time.perf_counter(); f.refill(); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print(it, f"next {1e3*(t1-t0):.2f} ms  issue {1e3*(t2-t1):.2f}  refill {1e3*(t3-t2):.2f}  drain {1e3*(t4-t3):.2f}", flush=True)
