#!/usr/bin/env python3
"""Host-side cost of one train_llm step: cProfile over a few steps (tottime by function)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import trace_aten
run = trace_aten.make_step(torch.device("cuda:0"), epoch=int(os.environ.get("EPOCH", "0")))
for _ in range(3):
    run()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    run()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
