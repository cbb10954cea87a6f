#!/usr/bin/env python3
"""Micro-benchmark of the training-mode WaveNet stack forward at the BASELINE.json shapes:

    python tools/bench_wn_stack.py [--V 9 --B 128] [--iters 100] [--stamps]

(a) hopmi_wn_stack_fwd, ONE persistent launch for the 8 layers incl. the BatchNorm statistics exchange: the dispatch's own
    begin / end timestamps (hopmi_time_next_launch; what rocprofv3's kernel trace reports) and back-to-back launches;
(b) the same forward as 8 x (hopmi_wn_layer_fwd + hopmi_wn_bn_finalize) launches, back-to-back and summed kernel durations.
Reports the algorithmic GB/s of SURVEY.md 8(d) (fused layers: x in + x out + the last-4-frames skip tail, summed over the stack).
--stamps: builds tools/probes/libhopmi_stamps_stack.so (-DHOPMI_STAMPS) and prints where a workgroup of the persistent launch
spends its cycles, per layer (s_memtime of wave 0: shares, not durations)."""
import argparse
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import hopmi
from hopmi import ops

DIL = (1, 2, 1, 2, 1, 2, 1, 2)
PKG = os.path.join(ROOT, "hop-heterogeneous-topology-based-multimodal-entanglement-for-co-speech-gesture-generation_amd")
SO = os.path.join(ROOT, "tools", "probes", "libhopmi_stamps_stack.so")


def build_stamps():
    src = [os.path.join(PKG, "csrc", f) for f in ("api.hip", "gcn.hip", "wavenet.hip", "wavenet_bwd.hip", "wavenet_stack.hip")]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DHOPMI_STAMPS", *(["-DSTK_EXP_XLAT"] if os.environ.get("STK_EXP_XLAT") else []),
                    "-I" + os.path.join(ROOT, "include"), *src, "-o", SO], check=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--V", type=int, default=9)
    ap.add_argument("--B", default="128", help="batch, or a comma-separated list (--B 128,256,512,1024: the sweep of DESIGN.md 7.1 -- the "
                                               "batch-independent exchange floor and the per-clip streaming rate as separate numbers)")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--warm", type=float, default=1.5)
    ap.add_argument("--stamps", action="store_true")
    ap.add_argument("--build-stamps", action="store_true")
    ap.add_argument("--lib", default=None, help="alternative libhopmi.so (timing experiments)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"], help="storage type of x0, the saved y_l and the skip tails")
    a = ap.parse_args()
    batches = [int(x) for x in str(a.B).split(",")]
    if len(batches) > 1 and not (a.stamps or a.build_stamps):
        # one child per batch (fresh workspaces and plans), then the two-parameter fit: time = floor + per_clip * B
        import re
        rows = []
        for b in batches:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--V", str(a.V), "--B", str(b), "--iters", str(a.iters), "--dtype", a.dtype]
                               + (["--lib", a.lib] if a.lib else []), capture_output=True, text=True)
            line = next((ln for ln in r.stdout.splitlines() if ln.startswith("one launch")), None)
            print(f"B={b}: {line}", flush=True)
            m = re.search(r"kernel\s+([0-9.]+) us", line or "")
            if m:
                rows.append((b, float(m.group(1))))
        if len(rows) >= 2:
            n = len(rows); sx = sum(b for b, _ in rows); sy = sum(t for _, t in rows)
            sxx = sum(b * b for b, _ in rows); sxy = sum(b * t for b, t in rows)
            slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
            print(f"least-squares fit over {[b for b, _ in rows]}: {(sy - slope * sx) / n:.1f} us batch-independent + {slope:.3f} us per clip")
        return
    a.B = batches[0]
    if a.build_stamps:
        build_stamps()
        return
    if a.lib:
        hopmi._lib._LIB_PATH = os.path.abspath(a.lib)
    if a.stamps:
        hopmi._lib._LIB_PATH = SO
        hopmi._lib.SIGNATURES = {k: v for k, v in hopmi._lib.SIGNATURES.items()
                                 if k.startswith(("hopmi_wn_", "hopmi_gcn_", "hopmi_version", "hopmi_last_error", "hopmi_reload_env",
                                                  "hopmi_time_next", "hopmi_noop"))}
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = hopmi.gwnet(None, a.V, dropout=0, supports=None, gcn_bool=True, addaptadj=True, aptinit=None, in_dim=173, out_dim=173,
                    residual_channels=64, dilation_channels=64, skip_channels=256, end_channels=512).to(dev).train()
    sdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    x0 = torch.randn(a.B, 16, a.V, 64, device=dev).to(sdt)
    A1, A2 = m.adjacency()
    A1, A2 = A1.detach(), A2.detach()
    prep = ops.gcn_prepare(A1, A2)
    wimg = m._weight_images()
    tails = torch.empty(a.B, 4, a.V, 512, device=dev, dtype=sdt)
    tcn_b = [(m.filter_convs[i].bias.detach(), m.gate_convs[i].bias.detach()) for i in range(8)]
    mlp_b = [m.gconv[i].mlp.mlp.bias.detach() for i in range(8)]
    bns = list(m.bn)
    wa = torch.randn(8192, 8192, device=dev)
    t_end = time.time() + a.warm
    while time.time() < t_end:
        for _ in range(10):
            wa @ wa
        torch.cuda.synchronize()
    L = hopmi._lib.lib()
    grid = ops.wn_stack_supported(a.B, 16, a.V, DIL)
    print(f"V={a.V} B={a.B}: persistent grid {grid}")
    nbytes, flops, T = 0, 0, 16
    for l, d in enumerate(DIL):
        nbytes += (2 if a.dtype == "bf16" else 4) * 64 * a.V * (a.B * T + (a.B * (T - d) if l < 7 else 0) + 4 * a.B)
        flops += a.B * (T - d) * a.V * (2 * 2 * 2 * 64 * 64 + 2 * 192 * 64 + 4 * 64 * a.V)
        T -= d
    stack = lambda: ops.wn_stack_fwd(x0, wimg, tcn_b, prep, mlp_b, bns, tails, DIL)

    if a.stamps:
        stamps = torch.zeros(1024 * 8 * 16, dtype=torch.int64, device=dev)
        assert L.hopmi_debug_set_stamps_stack(ctypes.c_void_p(stamps.data_ptr())) == 0
        for _ in range(3):
            stamps.zero_()
            stack()
            torch.cuda.synchronize()
        st = stamps.view(-1, 8, 16)[:grid].cpu().double()
        names = ["(statistics seen ->) normalise, split, LDS", "TCN mfma", "gate, u stores", "tail store, node mix", "contraction mfma",
                 "epilogue: y stores, sums", "reduce, granules out", "drain y, flags (+ combiner: sweep, add, publish)",
                 "neighbour flags, next tile's loads issued", "final sweep, statistics"]
        t0 = st[:, 0, 0].min()
        print(f"  launch span (first block start -> last stamp): {(st.max() - t0):.0f} cycles; block start skew {(st[:, 0, 0].max() - t0):.0f}")
        for layer in range(8):
            row = st[:, layer]
            has_tile = row[:, 1] > 0
            r = row[has_tile]
            print(f"  layer {layer}: {int(has_tile.sum())} blocks with a tile; layer start (median, rel. launch) {(row[:, 0].median() - t0):.0f}; "
                  f"last block into exchange {(row[:, 6].max() - t0):.0f}; " + (f"all out of exchange {(row[:, 10].max() - t0):.0f}" if layer < 7 else ""))
            for k in range(6):
                sgm = r[:, k + 1] - r[:, k]
                print(f"      {names[k]:56s} median {sgm.median().item():7.0f}  max {sgm.max().item():7.0f}")
            print(f"      commit detail: start -> first barrier {(r[:, 12] - r[:, 0]).median().item():.0f}; convert + LDS writes {(r[:, 13] - r[:, 12]).median().item():.0f}; "
                  f"weight-load issue + barrier {(r[:, 1] - r[:, 13]).median().item():.0f}")
            if layer < 7:
                print(f"      final detail: sweep (wave 0) {(row[:, 14] - row[:, 9]).median().item():.0f} (max {(row[:, 14] - row[:, 9]).max().item():.0f}); "
                      f"-> barrier (slowest wave's sweep) {(row[:, 15] - row[:, 14]).median().item():.0f}; statistics + barrier {(row[:, 10] - row[:, 15]).median().item():.0f}")
            if layer < 7:
                print(f"      final sweep passes (thread 0): median {row[:, 11].median().item():.0f}  max {row[:, 11].max().item():.0f}")
            for k in range(6, 10 if layer < 7 else 7):
                sgm = row[:, k + 1] - row[:, k]
                sgm = sgm[(row[:, k + 1] > 0) & (row[:, k] > 0)]
                if sgm.numel():
                    print(f"      {names[k]:56s} median {sgm.median().item():7.0f}  max {sgm.max().item():7.0f}")
        return

    for _ in range(5):
        stack()
    torch.cuda.synchronize()
    ops.check_status_now()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        stack()
    e1.record()
    torch.cuda.synchronize()
    b2b = e0.elapsed_time(e1) * 1e3 / a.iters
    ops.TIMER = ops.KernelTimer()
    for _ in range(a.iters):
        stack()
    torch.cuda.synchronize()
    spans = ops.TIMER.spans["wn_stack_fwd"]
    ops.TIMER = None
    ds = sorted(s[0].elapsed_time(s[1]) * 1e3 for s in spans)
    kus = ds[len(ds) // 2]
    ops.check_status_now()
    print(f"one launch  : kernel {kus:7.2f} us (min {ds[0]:.2f}, p90 {ds[int(0.9 * len(ds))]:.2f}) = {nbytes / kus / 1e3:7.1f} GB/s algorithmic "
          f"({nbytes / kus / 8e6 * 100:.1f} % of 8 TB/s), {flops / kus / 1e6:.1f} TF fp32-equivalent; back-to-back (host-issued) {b2b:.2f} us")

    # the same forward as per-layer launches
    ops.STACK_ENABLED = False
    with torch.no_grad():
        x = x0
        for _ in range(3):
            m._skip_tails_fused(x, prep, wimg)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            m._skip_tails_fused(x, prep, wimg)
        e1.record()
        torch.cuda.synchronize()
        b2b_l = e0.elapsed_time(e1) * 1e3 / a.iters
        ops.TIMER = ops.KernelTimer()
        for _ in range(a.iters):
            m._skip_tails_fused(x, prep, wimg)
        torch.cuda.synchronize()
        sp = ops.TIMER.spans["wn_layer_fwd"]
        ops.TIMER = None
    per = sum(s[0].elapsed_time(s[1]) for s in sp) * 1e3 / a.iters
    print(f"8 + 8 launches: layer kernels {per:7.2f} us summed (+ 8 finalisation launches); back-to-back (host-issued, eager) {b2b_l:.2f} us")
    ops.STACK_ENABLED = True


if __name__ == "__main__":
    main()
