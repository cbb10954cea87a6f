#!/usr/bin/env python3
"""Diagnostic: which host-side source lines of the package launch the small aten kernels of one train_llm step?
Runs a few steps under torch.profiler (with_stack) and prints, per (aten op, innermost hopmi frame), the launch
count and device time per step."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import hopmi
from hopmi import synth
from hopmi.parallel import GradSync


def make_step(dev, dataset="TED", B=128, epoch=0):
    V = 9 if dataset == "TED" else 42
    torch.manual_seed(0)
    model = hopmi.Model(synth.model_configs(dataset), synth.build_bert(6), synth.SyntheticTokenizer(),
                        synth.SpeakerVocab(1370)).float().to(dev)
    disc = hopmi.ConvDiscriminator(3 * V).to(dev)
    model.train(); disc.train()
    hopmi.mixed_precision(os.environ.get("DTYPE") or None)     # DTYPE=bf16: configs[2] / [4]
    lr = 0.01
    g_opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=lr, betas=(0.5, 0.999), fused=True)
    d_opt = torch.optim.Adam(disc.parameters(), lr=lr * 0.1, betas=(0.5, 0.999), fused=True)
    sync = GradSync([model, disc])
    sargs = synth.step_args(dataset)
    batch = synth.synthetic_batch(B, V, 1234, dev)
    return lambda: hopmi.train_llm(sargs, epoch, batch["in_audio"], batch["log_melspec"], batch["text"],
                                   batch["target_dir_vec"], batch["vid_indices"], model, disc, g_opt, d_opt, sync)


def host_sites(run):
    """aten ops of one step by the innermost package frame that issued them (forward / host-side code only:
    the dispatch mode is thread-local, the autograd engine's backward thread is not seen)."""
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    sites = collections.Counter()

    class Mode(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func).replace("aten.", "")
            if not any(k in name for k in ("view", "reshape", "t.default", "transpose", "permute", "expand", "slice", "select",
                                           "unsqueeze", "squeeze", "detach", "alias", "as_strided", "unbind", "split", "chunk",
                                           "narrow", "size", "stride", "is_")):
                fr = "?"
                for f in reversed(traceback.extract_stack()[:-1]):
                    if "_amd/" in f.filename:
                        fr = f"{os.path.basename(f.filename)}:{f.lineno}"
                        break
                sites[(name, fr)] += 1
            return func(*args, **(kwargs or {}))

    with Mode():
        run()
    torch.cuda.synchronize()
    print(f"host-issued aten ops with kernels (approx.): {sum(sites.values())}")
    for (name, fr), n in sites.most_common(150):
        print(f"{n:5d}x  {name:44s} {fr}")


def main():
    steps = 3
    dev = torch.device("cuda:0")
    run = make_step(dev, epoch=int(os.environ.get("EPOCH", "0")))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    if os.environ.get("HOST", "0") == "1":
        # host issue time (step entry -> the step's single device->host copy) vs wall time per step
        import time
        from hopmi import steps as _steps
        marks = []
        orig = _steps._LossFetch.result
        def ret(self):
            marks.append(time.perf_counter())
            return orig(self)
        _steps._LossFetch.result = ret
        t_host = t_wall = 0.0
        n = 20
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run()
            t1 = time.perf_counter()
            t_host += marks[-1] - t0
            t_wall += t1 - t0
        print(f"host issue {t_host / n * 1e3:.2f} ms/step, wall {t_wall / n * 1e3:.2f} ms/step")
        return
    if os.environ.get("SITES", "1") == "1":
        host_sites(run)
        return
    shapes = os.environ.get("SHAPES", "0") == "1"    # group by input shapes instead (python stacks come back empty on this build)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=shapes) as prof:
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        t = getattr(ev, "self_device_time_total", 0)
        if t <= 0 or ev.device_type != torch.autograd.DeviceType.CPU:
            continue
        frame = "?"
        for f in ev.stack or []:
            if "_amd/" in f or "hopmi" in f:
                frame = f.split("/")[-1]
                break
        if shapes:
            frame = str([tuple(x) if isinstance(x, (list, tuple)) else x for x in (ev.input_shapes or [])][:3])
        k = (ev.name, frame)
        agg[k][0] += 1
        agg[k][1] += t
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print(f"ops owning device time: {sum(v[0] for _, v in rows) / steps:.0f} per step, {tot / steps / 1e3:.2f} ms/step device time")
    for (name, frame), (n, t) in rows[:120]:
        print(f"{t / steps:9.1f} us/step  {n / steps:6.1f}x  {name:40s} {frame}")


if __name__ == "__main__":
    main()
