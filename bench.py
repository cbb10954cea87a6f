#!/usr/bin/env python3
"""Benchmark of the HOP generator training step on MI355X (contract: see the task statement).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one `train_llm` call (train_eval/train_llm.py:9-98 semantics, epoch <= 10: two
generator forwards, one backward, Adam on 65.7 M parameters, plus the RCCL gradient all-reduce
when N > 1) on one batch of synthetic 34-frame clips already resident in HBM.  Workload at every
N: BASELINE.json configs[1] per GPU (TED 10-joint = 9 graph nodes, batch 128 per GPU, fp32), so
scaling is weak.  Prints ONE JSON line on rank 0 carrying the `roofline` of the gwnet graph-conv
kernel (algorithmic bytes / HIP-event time, measured live in the timed region) and, at N = 1, the
`cpu_baseline` (the parity-pinned CPU oracle timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
F32_MFMA_PEAK_TFLOPS = 157.3   # exact-f32 MFMA == vector peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="clips per GPU (configs[1]: 128)")
    ap.add_argument("--dataset", default="TED", choices=["TED", "TED_expressive"])
    ap.add_argument("--epoch", type=int, default=0, help="> 10 adds the GAN discriminator step")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                    help="fp32 = the headline configuration (configs[1]); bf16 = library GEMMs under autocast (configs 2/4)")
    ap.add_argument("--no-tuned-gemms", action="store_true", help="library-default GEMM selection instead of the shipped table")
    ap.add_argument("--rehearse-sync", action="store_true",
                    help="N=1 only: run GradSync's bucketed RCCL path on a 1-rank group (overhead rehearsal of the N>1 path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-steps", type=int, default=3)
    return ap.parse_args()


def cpu_baseline(args, V):
    """The CPU oracle (oracle/ref_cpu.py, pinned to the reference by tests/golden) running the same
    train_llm step on this box's host cores: BASELINE.json configs[0] (batch 4, fp32, full step)."""
    import torch
    from transformers import BertConfig
    from hopmi import synth
    from oracle import ref_cpu, spec
    B = args.cpu_batch
    # the box's CPU share for one GPU is 16 cores; torch's default (all 128 hardware threads)
    # oversubscribes it 8x
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    bcfg = BertConfig(num_hidden_layers=6)
    g_sd = spec.build_sd(spec.model_spec(V, bcfg, 1370))
    d_sd = spec.build_sd(spec.disc_spec(3 * V), salt=1)
    for k, v in g_sd.items():
        if v.is_floating_point() and not k.startswith("llm_model.") and k != "word_embeddings" and "running_" not in k:
            v.requires_grad_(True)
    for k, v in d_sd.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    g_opt = torch.optim.Adam([v for v in g_sd.values() if v.requires_grad], lr=1e-2, betas=(0.5, 0.999))
    d_opt = torch.optim.Adam([v for v in d_sd.values() if v.requires_grad], lr=1e-3, betas=(0.5, 0.999))
    batch = synth.synthetic_batch(B, V, 1234, "cpu")
    batch["text"] = batch["text"]
    rng = lambda kind, shape: torch.randperm(shape[0]) if kind == "perm" else torch.randn(shape)
    cfg, sargs = synth.model_configs(args.dataset), synth.step_args(args.dataset)
    times = []
    for i in range(1 + args.cpu_steps):
        t0 = time.perf_counter()
        ref_cpu.train_llm_step(sargs, cfg, args.epoch, batch, g_sd, d_sd, g_opt, d_opt, rng, bert_heads=12)
        times.append(time.perf_counter() - t0)
    times = sorted(times[1:])
    med = times[len(times) // 2]
    return {"value": B / med, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{args.cpu_steps} train_llm steps (median, after 1 warm-up) at batch {B}, {args.dataset}, "
                      f"fp32, 6-layer BERT-base geometry; {med:.3f} s/step"}


def main():
    args = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL; must precede the first HIP call
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback for the product path)"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    # HOPMI_REHEARSE_ONE_GPU=1 (rehearsal on a single-GPU box only): every rank uses cuda:0 and the collectives go
    # through gloo, so the multi-process plumbing (rendezvous, plan agreement, timing reduction) can be exercised
    # without a second device; run it with HOPMI_GRU_PERSISTENT=0 (two persistent launches cannot share one GPU)
    one_gpu = os.environ.get("HOPMI_REHEARSE_ONE_GPU") == "1"
    dev = torch.device("cuda", 0 if one_gpu else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)    # backend "nccl" is RCCL on ROCm
    elif args.rehearse_sync:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    import hopmi
    from hopmi import ops, synth
    from hopmi.parallel import GradSync

    V = 9 if args.dataset == "TED" else 42
    B = args.batch
    hopmi.mixed_precision(None if args.dtype == "fp32" else args.dtype)
    # (the bf16 mode is host-bound at these sizes: the per-call table lookup costs more than the selection gains)
    tuned = (not args.no_tuned_gemms) and args.dtype == "fp32" and hopmi.use_tuned_gemms()
    torch.manual_seed(0)                                       # identical replicas
    model = hopmi.Model(synth.model_configs(args.dataset), synth.build_bert(6), synth.SyntheticTokenizer(),
                        synth.SpeakerVocab(1370)).float().to(dev)
    disc = hopmi.ConvDiscriminator(3 * V).to(dev)
    model.train()
    disc.train()
    lr = 0.01 if args.dataset == "TED" else 0.005              # run_ted.py:103 / run_expressive.py:100
    g_opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=lr, betas=(0.5, 0.999), fused=True)
    d_opt = torch.optim.Adam(disc.parameters(), lr=lr * 0.1, betas=(0.5, 0.999), fused=True)
    sync = GradSync([model, disc], force=args.rehearse_sync)
    sargs = synth.step_args(args.dataset)
    batch = synth.synthetic_batch(B, V, 1234 + rank, dev)

    def step():
        return hopmi.train_llm(sargs, args.epoch, batch["in_audio"], batch["log_melspec"], batch["text"],
                               batch["target_dir_vec"], batch["vid_indices"], model, disc, g_opt, d_opt, sync)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ops.TIMER = ops.KernelTimer()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    fence()
    elapsed = time.perf_counter() - t0
    timer, ops.TIMER = ops.TIMER, None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    if rank == 0:
        ks = timer.summary()
        gf = ks["wn_layer_fwd"]
        # HBM traffic of the same kernel from the PMC passes committed under profiles/ (rocprofv3 cannot run
        # inside this process): (2*FETCH_SIZE + WRITE_SIZE) KB per launch, gfx950 correction applied there
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
                traffic = json.load(f)["kernels"]["wn_layer_fwd"]["hbm_bytes_per_launch"] if (V == 9 and B == 128) else None
        except (OSError, KeyError, ValueError):
            traffic = None
        gbs = gf["bytes"] / (gf["kernel_ms"] * 1e-3) / 1e9
        tfl = gf["flops"] / (gf["kernel_ms"] * 1e-3) / 1e12
        out = {
            "metric": "training clips/sec (34-frame, 10-joint TED)" if V == 9 else "training clips/sec (34-frame, 43-joint TED-Expressive)",
            "value": world * B * args.steps / elapsed, "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.dtype == "fp32" else "bf16 GEMMs (autocast) + f32 HIP kernels, f32 master weights", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{1 if V == 9 else 3}] per GPU: {args.dataset} {V + 1}-joint ({V} graph nodes), "
                                   f"34-frame clips, batch {B}/GPU, {args.dtype}, full train_llm step "
                                   f"({'GAN phase' if args.epoch > 10 else 'epoch<=10: 2 generator forwards + backward + Adam'})",
                       "global_batch": world * B, "per_gpu_batch": B, "parallelism": f"dp{world}",
                       "llm": "BERT-base geometry, 6 layers, random init, frozen",
                       "library_gemm_selection": "shipped TunableOp table (replay only)" if tuned else "library default",
                       "losses": last},
            # the gwnet graph conv runs inside the fused WaveNet-layer kernel (BN-on-load, gated TCN, skip tail,
            # node mix, graph conv, residual, BN statistics): algorithmic bytes = xin read + y / saved gates /
            # skip tail written, per launch
            "roofline": {"kernel": "wn_layer_fwd_kernel (fused WaveNet layer incl. the gwnet graph conv; 8 layers of the training forward, gates saved)",
                         "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "launches": gf["launches"], "avg_us": 1e3 * gf["kernel_ms"] / gf["launches"],
                         "avg_us_with_event_pair": 1e3 * gf["total_ms"] / gf["launches"],
                         "event_pair_overhead_us": 1e3 * gf["event_overhead_ms"],
                         "algorithmic_bytes_per_launch": gf["bytes"] / gf["launches"],
                         "f32_mfma_tflops": tfl, "f32_mfma_frac": tfl / F32_MFMA_PEAK_TFLOPS,
                         "timing": "start/stop HIP events attached to each wn_layer_fwd dispatch (hipExtLaunchKernelGGL) on the launch "
                                   "stream, inside the timed region; the other kernels' *_avg_us are event pairs minus the "
                                   "smallest empty-pair interval"},
        }
        for name in ("wn_layer_bwd", "reprog_attn_fwd", "reprog_attn_bwd", "bert_attn_fwd", "bert_attn_bwd", "gru_fwd", "gru_bwd"):
            if name in ks and ks[name]["launches"]:
                k = ks[name]
                out["roofline"][name + "_avg_us"] = 1e3 * k["kernel_ms"] / k["launches"]
                if k["flops"]:
                    out["roofline"][name + "_tflops"] = k["flops"] / (k["kernel_ms"] * 1e-3) / 1e12
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, V)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
