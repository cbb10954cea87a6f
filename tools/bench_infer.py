#!/usr/bin/env python3
"""Batch-1 inference latency of hopmi.Model (eval, no_grad) and of the windowed generate_long loop
(test_checkpoint.py:395-472) on synthetic inputs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hopmi
from hopmi import synth

dev = torch.device("cuda:0")
V = 9
torch.manual_seed(0)
m = hopmi.Model(synth.model_configs("TED"), synth.build_bert(6), synth.SyntheticTokenizer(), synth.SpeakerVocab(1370)).float().to(dev)
m.eval()
if os.environ.get("TUNED", "1") == "1":
    hopmi.use_tuned_gemms()
for B in (1, 8, 64):
    b = synth.synthetic_batch(B, V, 1, dev)
    args = (b["in_audio"], b["log_melspec"], b["text"], b["target_dir_vec"][:, :16], b["vid_indices"])
    with torch.no_grad():
        for _ in range(5):
            m(*args)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            m(*args)
        torch.cuda.synchronize()
    print(f"eval forward B={B}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per call")
W = 8
b = synth.synthetic_batch(W, V, 2, dev)
pre0 = b["target_dir_vec"][:1, :16]
vid = b["vid_indices"][:1]
hopmi.generate_long(m, b["in_audio"], b["log_melspec"], b["text"], pre0, vid)
torch.cuda.synchronize(); t0 = time.perf_counter()
out = hopmi.generate_long(m, b["in_audio"], b["log_melspec"], b["text"], pre0, vid)
torch.cuda.synchronize()
print(f"generate_long {W} windows -> {tuple(out.shape)}: {(time.perf_counter() - t0) * 1e3:.1f} ms ({(time.perf_counter() - t0) / W * 1e3:.2f} ms per window)")

hopmi.generate_long(m, b["in_audio"], b["log_melspec"], b["text"], pre0, vid, use_graph=True)      # capture
torch.cuda.synchronize(); t0 = time.perf_counter()
out = hopmi.generate_long(m, b["in_audio"], b["log_melspec"], b["text"], pre0, vid, use_graph=True)
torch.cuda.synchronize()
print(f"generate_long {W} windows, hipGraph window forward: {(time.perf_counter() - t0) * 1e3:.1f} ms ({(time.perf_counter() - t0) / W * 1e3:.2f} ms per window)")
