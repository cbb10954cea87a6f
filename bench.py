#!/usr/bin/env python3
"""Benchmark of the HOP generator training step on MI355X (contract: see the task statement).

  python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE in the environment: this process starts the N
                                                          ranks itself as children under torch.distributed.run; with WORLD_SIZE set
                                                          -- the driver's own torch.distributed.run line -- it is one of the ranks)

A "step" is one `train_llm` call (train_eval/train_llm.py:9-98 semantics, epoch <= 10: two generator forwards, one
backward, Adam on 65.7 M parameters, plus the RCCL gradient exchange when N > 1) on one batch of synthetic 34-frame
clips already resident in HBM.  Workload at every N: BASELINE.json configs[1] per GPU (TED 10-joint = 9 graph nodes,
batch 128 per GPU, fp32), so scaling is weak.

Three separate regions, in this order:
  1. kernel region (`--kernel-steps`, untimed for `value`): eager `train_llm` steps with HIP events on every hand-written
     kernel launch -> the `roofline` object (algorithmic bytes / event time of the fused WaveNet-layer kernel);
  2. W warm-up steps, then EXACTLY K timed steps between two barrier + synchronize fences with NO instrumentation:
     `value` = world * B * K / elapsed (max over ranks).  The steps are hopmi.GraphedTrainStep calls (the recorded
     hipGraphs of the same train_llm; `--eager` times train_llm itself);
  3. rank 0, N = 1 only: `cpu_baseline`, the parity-pinned CPU oracle on this box's host cores.
Prints ONE JSON line on rank 0.
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
PROFILE_ROUND = "r06"
EXIT_NON_FINITE = 3            # the model left the timed region with a non-finite loss / parameter: no `value` is reported


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="clips per GPU (configs[1]: 128)")
    ap.add_argument("--dataset", default="TED", choices=["TED", "TED_expressive"])
    ap.add_argument("--epoch", type=int, default=0, help="> 10 adds the GAN discriminator step")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                    help="fp32 = the headline configuration (configs[1]); bf16 = configs 2/4 (bf16 GEMMs + bf16 gradient exchange)")
    ap.add_argument("--eager", action="store_true", help="time steps.train_llm itself instead of its recorded hipGraphs")
    ap.add_argument("--kernel-steps", type=int, default=6, help="instrumented eager steps for the roofline object (0 = none)")
    ap.add_argument("--bert-gemm", default="f16x2", choices=["library", "f16x2", "split3", "split2"],
                    help="the frozen BERT's linears: the library's fp32 GEMM; hopmi_gemm_f16x2 (two scaled fp16 parts per operand, "
                         "three MFMA terms: fp32-equivalent, default); hopmi_gemm_split with 3 bf16 parts per operand (six MFMA terms: "
                         "fp32-equivalent) or 2 parts (three terms: 2^-16-class products)")
    ap.add_argument("--strict-fp32", action="store_true",
                    help="the WaveNet forward, reprogramming attention and GRU recurrences as their composed fp32-exact forms "
                         "(hopmi.strict_fp32: the tests' second evaluation of those operators; slower -- less fused -- and since round 5 "
                         "of the same accuracy class as the default)")
    ap.add_argument("--no-tuned-gemms", action="store_true", help="library-default GEMM selection instead of the shipped table")
    ap.add_argument("--tuned-table", default=None, help="another TunableOp table than the shipped one (A/B runs)")
    ap.add_argument("--flat-exchange", action="store_true",
                    help="N > 1 (or --rehearse-sync): one all-reduce behind the whole generator backward instead of the overlapped form")
    ap.add_argument("--rehearse-sync", action="store_true",
                    help="N=1 only: run the N>1 exchange path on a 1-rank RCCL group (overhead rehearsal)")
    ap.add_argument("--feed-host", action="store_true",
                    help="every step takes a fresh batch from HOST memory through hopmi.HostFeeder (pinned double-buffered staging, "
                         "log-mel computed on the GPU) instead of the batch resident in HBM")
    ap.add_argument("--plumbing-check", action="store_true",
                    help="no GPU work: every rank joins a gloo group, the ranks run the timing reduction (barrier, MAX over ranks) and rank 0 "
                         "prints a stub line with n_gpus and the group size -- what tests/test_parallel_gloo.py uses to run the launch "
                         "path of `python bench.py --gpus N` on a box without GPUs")
    ap.add_argument("--poison", action="store_true",
                    help="TEST HOOK (tests/test_gpu_graph.py::test_bench_refuses_to_report_a_dead_model): a NaN is written into one "
                         "parameter behind the warm-up steps, so that the run must end with exit code 3 and no `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--no-cpu-b128", action="store_true", help="skip the one CPU step at the GPU's batch size")
    return ap.parse_args()


def cpu_baseline(args, V):
    """The CPU oracle (oracle/ref_cpu.py, pinned to the reference by tests/golden) running the same train_llm step on
    this box's host cores: BASELINE.json configs[0] (batch 4, fp32, full step) and, once, the GPU's batch size
    (the batch-independent mapping layer is half of the CPU step at batch 4)."""
    import torch
    from transformers import BertConfig
    from hopmi import synth
    from oracle import ref_cpu, spec
    # the box's CPU share for one GPU is 16 cores; torch's default (all hardware threads) oversubscribes it
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    bcfg = BertConfig(num_hidden_layers=6)
    cfg, sargs = synth.model_configs(args.dataset), synth.step_args(args.dataset)
    rng = lambda kind, shape: torch.randperm(shape[0]) if kind == "perm" else torch.randn(shape)

    def run(B, n_steps):
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
        times = []
        for _ in range(1 + n_steps):
            t0 = time.perf_counter()
            ref_cpu.train_llm_step(sargs, cfg, args.epoch, batch, g_sd, d_sd, g_opt, d_opt, rng, bert_heads=12)
            times.append(time.perf_counter() - t0)
        times = sorted(times[1:])
        return times[len(times) // 2]

    B = args.cpu_batch
    med = run(B, args.cpu_steps)
    out = {"value": B / med, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{args.cpu_steps} train_llm steps (median, after 1 warm-up) at batch {B}, {args.dataset}, "
                     f"fp32, 6-layer BERT-base geometry; {med:.3f} s/step"}
    if not args.no_cpu_b128 and args.batch != B:
        t = run(args.batch, 1)
        out["at_gpu_batch"] = {"value": args.batch / t, "unit": "clips/s", "batch": args.batch,
                               "sample": f"1 step after 1 warm-up at batch {args.batch}; {t:.2f} s/step"}
    return out


def roofline(ks, V, B, n_kernel_steps, dtype="fp32"):
    """The roofline object from the kernel region's event timings.  The graded kernel is the fused WaveNet forward: as ONE
    persistent launch per training forward (hopmi_wn_stack_fwd: 8 layers incl. the BatchNorm statistics exchange) when the
    run used it, else the per-layer launches (hopmi_wn_layer_fwd)."""
    stack = "wn_stack_fwd" in ks and ks["wn_stack_fwd"]["launches"] > 0
    name = "wn_stack_fwd" if stack else "wn_layer_fwd"
    gf = ks[name]
    sec = gf["kernel_ms"] * 1e-3
    gbs = gf["bytes"] / sec / 1e9
    moved = (gf["bytes"] + gf["extra_bytes"]) / sec / 1e9
    tfl = gf["flops"] / sec / 1e12
    traffic = None        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot run in here)
    try:
        with open(os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_traffic.json")) as f:
            t = json.load(f)
        if t.get("V") == V and t.get("B") == B and dtype == "fp32":     # (the PMC passes ran the fp32 configuration)
            traffic = t["kernels"][name]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        traffic = None
    floor_us = 1e3 * ks["noop"]["kernel_ms"] / ks["noop"]["launches"] if ks.get("noop", {}).get("launches") else None
    r = {"kernel": ("wn_stack_fwd_kernel: the WHOLE WaveNet stack of a training forward as one persistent launch -- per layer BN-on-load, "
                    "gated TCN, skip tail, node mix, graph conv, residual, BatchNorm batch statistics, plus the chip-wide exchange of "
                    "those statistics between layers (training-mode BatchNorm makes every layer depend on all clips) and the "
                    "BatchNorm finalisation" if stack else
                    "wn_layer_fwd_kernel (fused WaveNet layer incl. the gwnet graph conv: BN-on-load, gated TCN, skip tail, "
                    "node mix, graph conv, residual, BN statistics); the 8 layer launches of every training forward"),
         "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
         "traffic": traffic,
         "traffic_source": (None if traffic is None else
                            f"profiles/{PROFILE_ROUND}_traffic.json: HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this tree "
                            "on an MI355X (tools/profile_round.sh; corrected as MI355X_MICROARCH.md prescribes) -- a COMMITTED measurement, not one "
                            "taken inside this run: bench.py cannot run the profiler on itself"),
         "launches": gf["launches"], "avg_us": 1e3 * gf["kernel_ms"] / gf["launches"],
         "algorithmic_bytes_per_launch": gf["bytes"] / gf["launches"],
         "bytes_definition": ("SURVEY.md 8(d), fused layers, summed over the 8 layers of the launch: 4 B x 64 ch x V x (B*T_in read + "
                              "B*T_out written (not the dead last layer's) + 4*B skip-tail frames)" if stack else
                              "SURVEY.md 8(d), fused layer: 4 B x 64 ch x V x (B*T_in read + B*T_out written + 4*B skip-tail frames)"),
         "moved_bytes_per_launch": (gf["bytes"] + gf["extra_bytes"]) / gf["launches"],
         "moved_gbs": moved, "moved_frac": moved / HBM_PEAK_GBS,
         "f32_mfma_equiv_tflops": tfl, "f32_mfma_equiv_frac": tfl / F32_MFMA_PEAK_TFLOPS,
         "empty_kernel_us": floor_us,
         "empty_kernel_note": "duration the SAME timing method reports for an empty kernel of a layer launch's shape, launched next to "
                              "the graded kernel in every forward (rocprofv3's kernel trace shows the same ~3.6-3.9 us): "
                              "`achieved` / `frac` use the raw durations and so include it",
         "frac_net_of_empty_kernel": (gf["bytes"] / ((gf["kernel_ms"] - gf["launches"] * floor_us * 1e-3) * 1e-3) / 1e9 / HBM_PEAK_GBS
                                      if floor_us is not None and gf["kernel_ms"] > gf["launches"] * floor_us * 1e-3 else None),
         "timing": f"start/stop HIP events attached to each {name} dispatch (hipExtLaunchKernelGGL) on the launch stream, "
                   f"over {n_kernel_steps} instrumented eager train_llm steps run before the timed region (the timed region "
                   "itself carries no instrumentation); the other kernels' *_avg_us are event pairs minus the smallest "
                   "empty-pair interval"}
    if stack:
        r["previous_form"] = ("rounds 1-2 ran this forward as 8 x (wn_layer_fwd + wn_bn_finalize) launches: 165 us per forward at "
                              "TED / B = 128 (profiles/r02_bench_kernel_stats.csv), i.e. 0.043 of the HBM peak on the same bytes")
    for name2 in ("wn_layer_regate", "wn_layer_bwd", "reprog_attn_fwd", "reprog_attn_bwd", "bert_attn_fwd", "bert_attn_bwd", "gru_fwd", "gru_bwd"):
        if name2 in ks and ks[name2]["launches"]:
            k = ks[name2]
            r[name2 + "_avg_us"] = 1e3 * k["kernel_ms"] / k["launches"]
            if k["flops"]:
                r[name2 + "_tflops"] = k["flops"] / (k["kernel_ms"] * 1e-3) / 1e12
    return r


def launch_ranks(args):
    """`python bench.py --gpus N` typed as such (no WORLD_SIZE in the environment): start the N ranks as FRESH child processes
    under torch.distributed.run (one process per GPU, run_ted.py:110-112 / accelerate in the reference), pass rank 0's JSON line
    and the launcher's exit code through.  This process never touches the GPU (no torch import before this point), and nothing is
    re-executed in place."""
    import socket
    import subprocess
    with socket.socket() as s:                                 # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, len(os.sched_getaffinity(0)) // args.gpus))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL; must precede the first HIP call
    import torch
    import torch.distributed as dist

    # host threads: the box's CPU share per GPU is 16 cores; torch's default (every hardware thread) oversubscribes it, and
    # spinning intra-op workers starve the ROCm runtime's completion threads (seen as 60-160 ms stalls of the step)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} in the environment"
    if args.plumbing_check:
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"plumbing_check": True, "n_gpus": world, "ranks_in_group": dist.get_world_size() if world > 1 else 1,
                              "max_over_ranks": t.item(), "steps": args.steps, "warmup": args.warmup}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback for the product path)"
    # HOPMI_REHEARSE_ONE_GPU=1 (rehearsal on a single-GPU box only): every rank uses cuda:0 and the collectives go
    # through gloo, so the multi-process plumbing (rendezvous, exchange, timing reduction) can be exercised without a
    # second device (the persistent GRU kernels are withheld: two such launches cannot share one GPU)
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
    hopmi.strict_fp32(bool(args.strict_fp32))
    hopmi.gemm_parts({"library": 0, "f16x2": 16, "split3": 3, "split2": 2}[args.bert_gemm])      # (fp32 mode only: bf16 mode autocasts)
    tuned = (not args.no_tuned_gemms) and hopmi.use_tuned_gemms(args.tuned_table)      # (the table holds fp32 and bf16 shapes)
    torch.manual_seed(0)                                       # identical replicas
    model = hopmi.Model(synth.model_configs(args.dataset), synth.build_bert(6), synth.SyntheticTokenizer(),
                        synth.SpeakerVocab(1370)).float().to(dev)
    disc = hopmi.ConvDiscriminator(3 * V).to(dev)
    model.train()
    disc.train()
    torch.manual_seed(1000 + rank)                             # ... that draw different noise / dropout masks
    lr = 0.01 if args.dataset == "TED" else 0.005              # run_ted.py:103 / run_expressive.py:100
    g_opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=lr, betas=(0.5, 0.999), fused=True)
    d_opt = torch.optim.Adam(disc.parameters(), lr=lr * 0.1, betas=(0.5, 0.999), fused=True)
    grad_dtype = torch.bfloat16 if args.dtype == "bf16" else None      # configs[2]/[4]: bf16 RCCL exchange
    sync = GradSync([model, disc], force=args.rehearse_sync, grad_dtype=grad_dtype)
    sargs = synth.step_args(args.dataset)
    batch = synth.synthetic_batch(B, V, 1234 + rank, dev)
    inputs = (batch["in_audio"], batch["log_melspec"], batch["text"], batch["target_dir_vec"], batch["vid_indices"])

    def eager_step():
        return hopmi.train_llm(sargs, args.epoch, *inputs, model, disc, g_opt, d_opt, sync)

    graphed = hopmi.GraphedTrainStep(sargs, model, disc, g_opt, d_opt, accelerator=sync, eager_calls=1,
                                     grad_dtype=grad_dtype, enabled=not args.eager, force_exchange=args.rehearse_sync,
                                     overlap=not args.flat_exchange,
                                     group=dist.group.WORLD if dist.is_initialized() else None)

    feeder, feed_times = None, []
    if args.feed_host:
        # host batches in the reference collate's dtypes (lmdb_data_loader.py:47-62): raw audio, text as float64, no mel
        import itertools
        hb = []
        for k in range(int(os.environ.get("HOPMI_BENCH_FEED_BATCHES", "3"))):
            b = synth.synthetic_batch(B, V, 4321 + 17 * k + rank, "cpu")
            hb.append(dict(audio_padded=b["in_audio"], text_token_padded=b["text"].double(), vec_seq=b["target_dir_vec"],
                           vid_indices=b["vid_indices"]))
        feeder = hopmi.HostFeeder(itertools.cycle(hb), dev)

    def step():
        if feeder is None:
            return graphed(args.epoch, *inputs)
        t0 = time.perf_counter()
        b = next(feeder)
        t1 = time.perf_counter()
        ret = graphed(args.epoch, b["in_audio"], b["log_melspec"], b["text"], b["target_dir_vec"], b["vid_indices"])
        t2 = time.perf_counter()
        feeder.refill()
        feed_times.append((t1 - t0, t2 - t1, time.perf_counter() - t2))
        return ret

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- 1. kernel region: instrumented eager steps (these also bring up handles / workspaces / autograd threads) ----
    ks = None
    if args.kernel_steps > 0 and not args.strict_fp32:         # (--strict-fp32 does not run the graded kernel: no roofline object)
        eager_step()                                           # first call: lazy initialisation, GradSync's plan
        ops.TIMER = ops.KernelTimer()
        for _ in range(args.kernel_steps):
            eager_step()
        torch.cuda.synchronize()
        timer, ops.TIMER = ops.TIMER, None
        ks = timer.summary()

    # ---- 2. set-up (the one-time recording of the step's graphs: like building the model, not a warm-up step), W warm-up
    #         steps, then the timed region ------------------------------------------------------------------------------
    setup_calls = 0
    while not args.eager and graphed.n_replay < 1 and setup_calls < 4:
        step()
        setup_calls += 1
    for _ in range(args.warmup):
        step()
    if args.poison:
        with torch.no_grad():
            model.align_layer.bias[0] = float("nan")
    fence()
    t0 = time.perf_counter()
    stamps = [t0]
    for _ in range(args.steps):
        last = step()
        stamps.append(time.perf_counter())
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # ---- the run only counts if the model it trained is alive: every loss of the last step and every trainable parameter finite
    # (round 5's headline was timed on a model that had gone to NaN inside the recorded step, and nothing said so) -------------------
    import math
    bad_losses = sorted(k for k, v in last.items() if not math.isfinite(v))
    with torch.no_grad():
        pmax = torch.stack(torch._foreach_norm([p for p in model.parameters() if p.requires_grad], float("inf")))
        n_bad_params = int((~torch.isfinite(pmax)).sum().item())
    alive = not bad_losses and n_bad_params == 0
    if world > 1:
        t = torch.tensor([0.0 if alive else 1.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        alive_all = t.item() == 0.0
    else:
        alive_all = alive
    losses_json = {k: (v if math.isfinite(v) else repr(v)) for k, v in last.items()}      # (strict JSON: no bare NaN)

    if rank == 0 and not alive_all:
        print(json.dumps({"metric": "training clips/sec", "value": None, "unit": "clips/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "error": "non-finite model state after the timed region"
                                                          + ("" if alive else f" (rank 0: losses {bad_losses}, {n_bad_params} parameter tensors)")
                                                          + ": the throughput of a dead model is not a measurement",
                          "losses": losses_json}), flush=True)
    if not alive_all:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(EXIT_NON_FINITE)

    if rank == 0:
        per_step = sorted(b - a for a, b in zip(stamps, stamps[1:]))
        median_ms = 1e3 * per_step[len(per_step) // 2]
        if os.environ.get("HOPMI_BENCH_STEP_TIMES") == "1":          # diagnostic: the individual step intervals
            print("step intervals (ms):", [round(1e3 * (b - a), 2) for a, b in zip(stamps, stamps[1:])], file=sys.stderr)
            print("graphed: eager calls", graphed.n_eager, "replays", graphed.n_replay, file=sys.stderr)
            if feed_times:
                print("feeder (next, step, refill) ms:", [tuple(round(1e3 * x, 2) for x in t) for t in feed_times[-args.steps:]], file=sys.stderr)
        gan = args.epoch > 10
        out = {
            "metric": "training clips/sec (34-frame, 10-joint TED)" if V == 9 else "training clips/sec (34-frame, 43-joint TED-Expressive)",
            "value": world * B * args.steps / elapsed, "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "median_ms_per_step": median_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": (("f32 storage, accumulation and elementwise arithmetic; every contraction of the hand-written kernels -- the frozen BERT's "
                       "and the large trainable linears (forward, activation gradient AND weight gradient), the fused WaveNet forward, the "
                       "reprogramming attention, the GRU recurrences -- as three MFMA terms of power-of-two-scaled fp16 hi/lo operands (22 "
                       "significand bits per operand, f32 accumulation: f32-equivalent, error against float64 within 4 x plain f32's, "
                       "tests/test_gpu_parity.py::test_*_vs_float64); WaveNet backward and BERT self-attention on the exact-f32 MFMA; "
                       + ("the WaveNet forward / reprogramming attention / GRU recurrences as their composed f32-exact forms (--strict-fp32); "
                          if args.strict_fp32 else "")
                       + {"f16x2": "", "split3": "the frozen BERT's linears as SIX split-bf16 terms (--bert-gemm split3); ",
                          "split2": "the frozen BERT's linears as THREE split-bf16 terms, 2^-16-class products (--bert-gemm split2); ",
                          "library": "the frozen BERT's linears on the library's f32 GEMM (--bert-gemm library); "}[args.bert_gemm]
                       + "the remaining library GEMMs (mapping layer forward, small projections) f32") if args.dtype == "fp32" else
                      "bf16 library GEMMs (autocast), bf16 activations between them (dtype argument of the BERT / GRU / attention HIP "
                      "kernels) + bf16 gradient exchange; inside the HIP kernels f32 accumulation and elementwise arithmetic with "
                      "three-term fp16 hi/lo MFMA products (f32-equivalent); f32 WaveNet stack, master weights and optimizer"),
            "data": "synthetic",
            "losses": losses_json,                     # of the LAST timed step (finite, or the run exits with EXIT_NON_FINITE above)
            "config": {"workload": f"BASELINE.json configs[{1 if V == 9 else 3}] per GPU: {args.dataset} {V + 1}-joint ({V} graph nodes), "
                                   f"34-frame clips, batch {B}/GPU, {args.dtype}, one full train_llm step = "
                                   + ("GAN phase (epoch > 10): discriminator step (1 no-grad generator forward, 2 discriminator forwards, "
                                      "backward, Adam) + generator step (2 generator forwards, 1 discriminator forward, backward, Adam)"
                                      if gan else
                                      "2 generator forwards (the graded one + the no-grad forward of the diversity regulariser, "
                                      "train_llm.py:58) + 1 backward + Adam on 65.7 M parameters (in the recorded step: the caller's "
                                      "torch.optim.Adam state updated by one hopmi_adam_multi launch, torch's fused-Adam arithmetic, "
                                      "tests/test_gpu_parity.py::test_adam_multi_equals_torch_fused_adam)")
                                   + "; the pose decoder's GRU recurrences of the generator step's two forwards (same weights) run as ONE launch per "
                                     "layer over both batches (Model.forward_pair / hopmi_gru_fwd_pair_dt; values of two separate calls at the fp32 "
                                     "class, tests/test_gpu_parity.py::test_forward_pair_equals_two_forwards)"
                                   + "; inside one step the batch-independent prototype branch (mapping layer, K/V projections) and the "
                                     "dropout-free audio branch (beat MLP, gwnet: one persistent launch for the 8 fused layers) are computed ONCE and reused by the "
                                     "step's other forwards, whose BatchNorm running-statistics update is replayed on the same partial "
                                     "sums (bit-identical to recomputing, tests/test_gpu_parity.py::test_step_cache_audio_branch_equals_recompute)"
                                   + ("; in the discriminator step the discriminator's per-sample part (GRU, linears) runs once on the real and "
                                      "the generated batch side by side (pre_conv with its per-call BatchNorm statistics separately); in the "
                                      "generator step the discriminator's own parameter gradients, which train_llm.py never uses (only "
                                      "model_optim steps; the next discriminator step starts with zero_grad), are not computed (bit-identical "
                                      "losses, parameters and buffers, tests/test_gpu_parity.py::test_train_llm_unused_discriminator_grads_elision)"
                                      if gan else "")
                                   + ("" if gan else "; the discriminator score that train_llm.py:43-44 computes in every epoch and :81 never "
                                      "uses before epoch 11 is not computed, only its lasting effect (the BatchNorm statistics update of the "
                                      "discriminator's pre_conv) is (bit-identical losses, parameters and buffers, "
                                      "tests/test_gpu_parity.py::test_train_llm_unused_score_elision)"),
                       "execution": "steps.train_llm issued from Python every step (--eager)" if args.eager else
                                    "hopmi.GraphedTrainStep: the launches of one steps.train_llm call recorded once as hipGraphs "
                                    "(cut behind the loss copy and around every collective) and replayed; dropout advances through a "
                                    "device-side seed word",
                       "input": "one batch resident in HBM" if feeder is None else
                                "a fresh batch from host memory every step (HostFeeder: pinned double-buffered staging, the 19 MB "
                                "host -> device copy and the log-mel kernels enqueued in front of the step)",
                       "global_batch": world * B, "per_gpu_batch": B, "parallelism": f"dp{world}",
                       "exchange": None if world == 1 and not args.rehearse_sync else {
                           "form": ("eager GradSync: 64 MB buckets launched from post-accumulate-grad hooks in gradient-production order "
                                    "(the persistent GRU / WaveNet-stack kernels are withheld while RCCL kernels may run beside the "
                                    "backward: the eager N > 1 step is a slower program than the recorded one)" if args.eager else
                                    "flat: one all-reduce behind the whole generator backward" if args.flat_exchange else
                                    "overlapped: the generator's backward is cut at the decoder input, the all-reduce of the first half's "
                                    "gradients (decoder GRU, head) runs under the second half, one flat all-reduce for the rest"),
                           "backend": dist.get_backend() if dist.is_initialized() else None,
                           "ranks_in_group": dist.get_world_size() if dist.is_initialized() else 1,
                           "gradient_dtype": args.dtype if grad_dtype is not None else "fp32",
                           "prototype_rows": "mapping-layer rows sharded over ranks: all-gather of S (4.6 MB), all-reduce of dS instead of "
                                             "the 183 MB weight gradient, Adam on own rows",
                           "all_reduces": ({"eager_buckets": [[len(b.params), b.numel] for g in sync.groups for b in (g.buckets or ())]}
                                           if args.eager else graphed.exchange_plan),
                           "where": "eager RCCL calls between graph launches (nothing of RCCL is captured)"},
                       "llm": "BERT-base geometry, 6 layers, random init, frozen",
                       "arithmetic": "per-kernel error against float64 next to plain fp32 torch: tests/test_gpu_parity.py::test_*_vs_float64 "
                                     "(error within 4 x the plain-fp32 evaluation's and <= 2e-6 relative; measured 0.7-1.3 x)",
                       "bert_gemm": ("library bf16 GEMMs (autocast)" if args.dtype != "fp32" else
                                     {"library": "library fp32 GEMMs (hipBLASLt)",
                                      "f16x2": "hopmi_gemm_f16x2 / hopmi_gemm_f16x2_ab_ep (the LayerNorm kernels write the next GEMM's operand image: "
                                               "both operands by LDS-DMA), 2 power-of-two-scaled fp16 parts per operand, 3 MFMA terms: fp32-equivalent "
                                               "(error vs float64 equal to the library's fp32 GEMM, tests/test_gpu_parity.py::test_gemm_split_vs_float64)",
                                      "split3": "hopmi_gemm_split, 3 bf16 parts per operand, 6 MFMA terms: fp32-equivalent (error vs float64 "
                                                "equal to the library's fp32 GEMM, tools/bench_gemm.py)",
                                      "split2": "hopmi_gemm_split, 2 bf16 parts per operand, 3 MFMA terms (2^-16-class products)"}[args.bert_gemm]),
                       "library_gemm_selection": "shipped TunableOp table (replay only)" if tuned else "library default"},
        }
        if ks is not None:
            out["roofline"] = roofline(ks, V, B, args.kernel_steps, args.dtype)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, V)
        print(json.dumps(out, allow_nan=False), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
