#!/usr/bin/env python3
"""Bench-regime divergence hunt (round 6): bench.py's exact set-up (default init under seed 0, lr 1e-2, dropout on, fused Adam,
tuned GEMM table, 1 + K instrumented eager steps, then the recorded step) with a finiteness census after EVERY step: losses,
parameters, gradients, buffers, Adam moments.  At the first step with a non-finite value it names the tensors, in backward order,
and stops.  Switches are taken from the environment (HOPMI_*), so variants are separate child processes:

    python3 tools/nan_hunt.py [--steps 45] [--eager] [--bert-gemm library] [--tag NAME]

Prints one summary line `HUNT <tag> first_bad=<step or none> ...` at the end."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=45)
    ap.add_argument("--kernel-steps", type=int, default=6)
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--dataset", default="TED")
    ap.add_argument("--epoch", type=int, default=0)
    ap.add_argument("--bert-gemm", default="f16x2")
    ap.add_argument("--tag", default="default")
    ap.add_argument("--every", type=int, default=1, help="census every n-th step")
    ap.add_argument("--lr", type=float, default=None)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    args = ap.parse_args()
    import torch
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    import hopmi
    from hopmi import ops, synth

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    V = 9 if args.dataset == "TED" else 42
    B = args.batch
    hopmi.mixed_precision(None if args.dtype == "fp32" else args.dtype)
    hopmi.gemm_parts({"library": 0, "f16x2": 16, "split3": 3, "split2": 2}[args.bert_gemm])
    hopmi.use_tuned_gemms(None)
    torch.manual_seed(0)
    model = hopmi.Model(synth.model_configs(args.dataset), synth.build_bert(6), synth.SyntheticTokenizer(),
                        synth.SpeakerVocab(1370)).float().to(dev)
    disc = hopmi.ConvDiscriminator(3 * V).to(dev)
    model.train()
    disc.train()
    torch.manual_seed(1000)
    lr = args.lr if args.lr is not None else (0.01 if args.dataset == "TED" else 0.005)
    g_opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=lr, betas=(0.5, 0.999), fused=True)
    d_opt = torch.optim.Adam(disc.parameters(), lr=lr * 0.1, betas=(0.5, 0.999), fused=True)
    sargs = synth.step_args(args.dataset)
    batch = synth.synthetic_batch(B, V, 1234, dev)
    inputs = (batch["in_audio"], batch["log_melspec"], batch["text"], batch["target_dir_vec"], batch["vid_indices"])
    graphed = hopmi.GraphedTrainStep(sargs, model, disc, g_opt, d_opt, eager_calls=1, enabled=not args.eager,
                                     grad_dtype=torch.bfloat16 if args.dtype == "bf16" else None)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    diag = ops.split_status(reset=True) is not None   # diagnostic library (HOPMI_LIB=.../libhopmi_dbg.so): registers the status buffer
    split_reports = []

    def census(step, losses):
        """-> (bad: bool, text)"""
        torch.cuda.synchronize()
        ps = [p for _, p in named]
        gs = [(n, p.grad) for n, p in named if p.grad is not None]
        pn = torch.stack(torch._foreach_norm(ps, float("inf")))
        gn = torch.stack(torch._foreach_norm([g for _, g in gs], float("inf"))) if gs else torch.zeros(1, device=dev)
        bufs = [(n, b) for n, b in model.named_buffers() if b.is_floating_point() and not n.startswith("llm_model.")]
        bn = torch.stack([b.abs().max() for _, b in bufs]) if bufs else torch.zeros(1, device=dev)
        bad_l = [k for k, v in losses.items() if v != v or abs(v) == float("inf")]
        bad_p = [named[i][0] for i in torch.nonzero(~torch.isfinite(pn)).flatten().tolist()]
        bad_g = [gs[i][0] for i in torch.nonzero(~torch.isfinite(gn)).flatten().tolist()]
        bad_b = [bufs[i][0] for i in torch.nonzero(~torch.isfinite(bn)).flatten().tolist()]
        fin_g = gn[torch.isfinite(gn)]
        line = (f"step {step:3d} losses {({k: round(v, 4) for k, v in losses.items()})} max|p| {pn[torch.isfinite(pn)].max().item():.3e} "
                f"max|g| {(fin_g.max().item() if fin_g.numel() else float('nan')):.3e} max|buf| {bn[torch.isfinite(bn)].max().item():.3e}")
        st = ops.split_status(reset=True)          # (None with the production library)
        if st is not None:
            line += f" splits {st}"
            if any(v is not None for v in st.values()):
                split_reports.append((step, st))
        bad = bool(bad_l or bad_p or bad_g or bad_b)
        if bad:
            line += f"\n   NON-FINITE: losses {bad_l}; {len(bad_p)} params; {len(bad_g)} of {len(gs)} grads; buffers {bad_b}"
            if bad_g and len(bad_g) < len(gs):
                good = [n for n, _ in gs if n not in set(bad_g)]
                line += f"\n   bad grads: {bad_g[:40]}\n   finite grads: {good[:60]}"
            elif bad_g:
                line += "\n   every gradient is non-finite (the forward, the loss or the head of the backward)"
            # largest finite gradients / parameters by tensor (which tensor carries the extreme values)
            top = sorted(((gn[i].item(), gs[i][0]) for i in range(len(gs)) if gn[i] == gn[i]), reverse=True)[:8]
            line += f"\n   largest |g|: {[(n, f'{v:.3e}') for v, n in top]}"
            top = sorted(((pn[i].item(), named[i][0]) for i in range(len(named)) if pn[i] == pn[i]), reverse=True)[:8]
            line += f"\n   largest |p|: {[(n, f'{v:.3e}') for v, n in top]}"
        return bad, line

    step_no, first_bad = 0, None
    hist = []

    def run(fn, label):
        nonlocal step_no, first_bad
        losses = fn()
        step_no += 1
        if step_no % args.every == 0 or any(v != v for v in losses.values()):
            bad, line = census(step_no, losses)
            print(f"[{label}] {line}", flush=True)
            hist.append(losses.get("loss"))
            if bad and first_bad is None:
                first_bad = step_no
        return first_bad is not None

    from hopmi.graph import _PlainBackward
    accel = _PlainBackward()
    eager = lambda: hopmi.train_llm(sargs, args.epoch, *inputs, model, disc, g_opt, d_opt, accel)
    stop = run(eager, "eager0")
    ops.TIMER = ops.KernelTimer()
    for _ in range(args.kernel_steps):
        if stop:
            break
        stop = run(eager, "eagerT")
    torch.cuda.synchronize()
    ops.TIMER = None
    for _ in range(args.steps):
        if stop:
            break
        stop = run(lambda: graphed(args.epoch, *inputs), "eager" if args.eager else f"graph(r{graphed.n_replay})")
    import json
    print("RESULT " + json.dumps({"tag": args.tag, "first_bad": first_bad, "steps_run": step_no, "replays": graphed.n_replay,
                                  "losses": hist, "split_reports": split_reports, "diagnostic_library": diag}), flush=True)
    print(f"HUNT {args.tag} first_bad={first_bad} steps_run={step_no} replays={graphed.n_replay} last_loss={hist[-1] if hist else None} "
          f"env={ {k: v for k, v in os.environ.items() if k.startswith('HOPMI_')} }", flush=True)


if __name__ == "__main__":
    main()
