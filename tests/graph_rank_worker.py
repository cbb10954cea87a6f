"""Worker of tests/test_gpu_graph.py::test_graphed_exchange_two_ranks_one_gpu (launched by torch.distributed.run, every
rank on cuda:0, gloo collectives): per rank, 4 training steps through hopmi.GraphedTrainStep with the N > 1 recording
(prototype rows sharded, flat gradient all-reduces between graph launches) and, on a second copy of the models, the
same steps through steps.train_llm + GradSync (the hook-driven bucketed exchange); prints one JSON line per rank."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch
import torch.distributed as dist


def main():
    os.environ["HOPMI_REHEARSE_ONE_GPU"] = "1"          # shared device: no persistent GRU launches
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    import hopmi
    from hopmi import steps
    from hopmi.parallel import GradSync
    from oracle.golden_util import step_args
    from test_gpu_graph import _pair
    epoch = int(sys.argv[1])
    steps._randn_like = lambda t: torch.full_like(t, 0.5)
    steps._randperm = lambda n, device: torch.arange(n - 1, -1, -1, device=device)
    if os.environ.get("HOPMI_WORKER_DUMMY_MB"):        # diagnostic: shift every later allocation
        _dummy = torch.empty(int(float(os.environ["HOPMI_WORKER_DUMMY_MB"]) * (1 << 20)), dtype=torch.uint8, device=dev)
    fullsize = len(sys.argv) > 4 and sys.argv[4] == "fullsize"
    if fullsize:
        # BERT-base geometry at 96 clips = 3 264 rows: past ops.IMG_MIN_ROWS / LINEAR_IMG_MIN_ROWS, so the operand-image paths (LayerNorm /
        # GEMM-epilogue / attention producers, the LDS-DMA GEMM form, the TN weight gradients) and their caches run BESIDE the exchange
        # -- the caches are keyed on version counters that graph._adam_own_rows and _mark_written move outside the optimizer
        from test_gpu_parity import _full_size_setup
        m1, d1, _, inp = _full_size_setup(9, 96)
        m1._randn_like = lambda t: torch.full_like(t, 0.25)
        m1, d1 = m1.to(dev).train(), d1.to(dev).train()
        inp = {k: v.to(dev) for k, v in inp.items()}
    else:
        m1, d1, inp = _pair(9, dev)
    # a different batch per rank (same replicas)
    names = ("in_audio", "log_melspec", "text", "target_dir_vec", "vid_indices")
    batch = tuple((inp[k] * (1.0 + 0.25 * rank)) if inp[k].is_floating_point() else inp[k].roll(rank, 0) for k in names)
    # Biases in front of a training-mode BatchNorm and the key-projection bias have analytically zero gradients: Adam would
    # move every element of them by +-lr per step on rounding noise, a random walk that differs between the two exchange paths
    # and, through the forward, perturbs every other gradient by per cent within a few steps (tools/probes/grad_sensitivity.py).
    # They are frozen here so that the comparison measures the exchange paths, not that walk.
    # (pre_conv.1.bias: the shift of the first BatchNorm, which the next conv turns into a per-channel constant in front of the second)
    noise = lambda n: n.endswith("mlp.mlp.bias") or n in ("pre_conv.0.bias", "pre_conv.1.bias", "pre_conv.3.bias") or n.endswith("key_projection.bias")
    for mod in (m1, d1):
        for n, p in mod.named_parameters():
            if noise(n):
                p.requires_grad_(False)
    m2, d2 = copy.deepcopy(m1), copy.deepcopy(d1)
    m2._randn_like = m1._randn_like
    mk = lambda m, d: (torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999)),
                       torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999)))
    g1, o1 = mk(m1, d1)
    g2, o2 = mk(m2, d2)
    args = step_args(9)
    sync1 = GradSync([m1, d1], bucket_mb=0.25)
    sync2 = GradSync([m2, d2], bucket_mb=0.25)
    # argv[3]: "flat" = one all-reduce behind the whole generator backward; default: the overlapped form (backward cut at the decoder input)
    overlap = not (len(sys.argv) > 3 and sys.argv[3] == "flat")
    graphed = hopmi.GraphedTrainStep(args, m2, d2, g2, o2, accelerator=sync2, eager_calls=1, group=dist.group.WORLD, overlap=overlap)
    # a status word this worker controls, recorded as if a persistent GRU launch of the backward had produced it (there are
    # none here: shared device): set on ONE rank at the end, it must stop EVERY rank at the same call
    fake_status = torch.zeros((), dtype=torch.int32, device=dev)
    plain_after_backward = graphed._after_backward

    def after_backward_with_status(cap, is_disc):
        cap.status.append(fake_status)
        return plain_after_backward(cap, is_disc)

    graphed._after_backward = after_backward_with_status
    losses1, losses2 = [], []
    # Two orders (argv[2]): "sequential" runs the two paths one after the other, "interleaved" queues an eager train_llm step
    # of the OTHER model copy between two replays.  Round 2 found a wrong gradient in the interleaved order (the GAN-phase
    # recording's value-projection bias gradient, a torch multi-block reduction behind a semaphore memset NODE, from the third
    # replay on); the recording holds no memset node any more (tests/test_gpu_graph.py::test_recorded_step_has_no_memset_nodes)
    # and the interleaved order is a required test.
    if len(sys.argv) > 2 and sys.argv[2] == "interleaved" or os.environ.get("HOPMI_WORKER_INTERLEAVE"):
        for it in range(4):
            losses1.append(hopmi.train_llm(args, epoch, *batch, m1, d1, g1, o1, sync1))
            if os.environ.get("HOPMI_WORKER_SYNC_BEFORE"):
                torch.cuda.synchronize()
            losses2.append(graphed(epoch, *batch))
            if os.environ.get("HOPMI_WORKER_SYNC_AFTER"):
                torch.cuda.synchronize()
    else:
        for it in range(4):
            losses2.append(graphed(epoch, *batch))
        for it in range(4):
            losses1.append(hopmi.train_llm(args, epoch, *batch, m1, d1, g1, o1, sync1))
    if os.environ.get("HOPMI_WORKER_POOLCHECK"):
        snap = torch.cuda.memory_snapshot()
        segs = [(sg["address"], sg["address"] + sg["total_size"], tuple(sg.get("segment_pool_id", (0, 0)))) for sg in snap]
        def pool_of(ptr):
            for a, b_, pid in segs:
                if a <= ptr < b_:
                    return pid
            return None
        rec = next(iter(graphed.records.values()))
        out = {}
        for n, p_ in list(m2.named_parameters()) + list(d2.named_parameters()):
            if p_.grad is not None:
                out.setdefault(str(pool_of(p_.grad.data_ptr())), []).append(n)
        print("POOLCHECK grads by pool:", {k: (len(v), v[:6]) for k, v in out.items()}, flush=True)
        print("POOLCHECK static:", [str(pool_of(t.data_ptr())) for t in rec["static"]], "keep:", [str(pool_of(t.data_ptr())) for t in rec["cap"].keep], flush=True)
    sharded_before = bool(graphed.sharded)
    own = (graphed.r0, graphed.r1)
    # rows this rank does not own have not moved since the recording began; unshard() fetches them from their owners
    stale = (m2.mapping_layer.weight - m1.mapping_layer.weight).abs().max().item()
    graphed.unshard()
    worst = {}
    for (n, a), (_, b) in zip(list(m1.named_parameters()) + list(d1.named_parameters()),
                              list(m2.named_parameters()) + list(d2.named_parameters())):
        diff = (a - b).abs()
        worst[n] = (diff.max().item(), diff.mean().item())
    # replicas must agree across ranks after the exchange: checksum of every parameter, all-gathered
    cs = torch.tensor([p.double().sum().item() for p in list(m2.parameters()) + list(d2.parameters())], dtype=torch.float64)
    allcs = [torch.empty_like(cs) for _ in range(world)]
    dist.all_gather(allcs, cs)
    spread = max((c - allcs[0]).abs().max().item() for c in allcs)
    # phase 2 -- replays AFTER an unshard() shard the mapping layer again (each rank only updates its rows); an eager call
    # that follows (a short last batch) and a later unshard() must both see that.  Same steps on the GradSync pair.
    for it in range(2):
        graphed(epoch, *batch)
    for it in range(2):
        hopmi.train_llm(args, epoch, *batch, m1, d1, g1, o1, sync1)
    resharded = bool(graphed.sharded)
    stale2 = (m2.mapping_layer.weight - m1.mapping_layer.weight).abs().max().item()
    short = tuple(t[:1].contiguous() for t in batch)
    n_eager_before = graphed.n_eager
    graphed(epoch, *short)                                   # another batch shape: the eager step, which unshards first
    hopmi.train_llm(args, epoch, *short, m1, d1, g1, o1, sync1)
    short_was_eager = graphed.n_eager == n_eager_before + 1 and not graphed.sharded
    graphed(epoch, *batch)                                   # ... and one more replay, then an explicit unshard()
    hopmi.train_llm(args, epoch, *batch, m1, d1, g1, o1, sync1)
    graphed.unshard()
    worst2 = {}
    for (n, a), (_, b) in zip(list(m1.named_parameters()) + list(d1.named_parameters()),
                              list(m2.named_parameters()) + list(d2.named_parameters())):
        diff = (a - b).abs()
        worst2[n] = (diff.max().item(), diff.mean().item())
    cs = torch.tensor([p.double().sum().item() for p in list(m2.parameters()) + list(d2.parameters())] +
                      [g2.state[p][k].double().sum().item() for p in (m2.mapping_layer.weight, m2.mapping_layer.bias) for k in ("exp_avg", "exp_avg_sq")],
                      dtype=torch.float64)
    allcs = [torch.empty_like(cs) for _ in range(world)]
    dist.all_gather(allcs, cs)
    spread2 = max((c - allcs[0]).abs().max().item() for c in allcs)
    # (after every comparison: these two calls train m2 further)
    # a hand-off time-out on rank 1 only: its status rides on the gradient exchange of that replay, and the NEXT call's loss
    # fetch raises on both ranks (rank 0 learns of it from the exchange alone)
    if rank == 1:
        fake_status.fill_(1)
    stopped_at = None
    for call in range(2):                  # epoch <= 10: the exchange follows the fetch, so the second call stops; GAN phase:
        try:                               # the discriminator's exchange precedes the fetch of the same call
            graphed(epoch, *batch)
        except RuntimeError as e:
            if "timed out" not in str(e):
                raise
            stopped_at = call
            break
    fake_status.zero_()
    graphed._bwd_status.zero_()
    graphed._peer_status.zero_()
    from hopmi import ops as _hops
    print("RANKJSON " + json.dumps(dict(rank=rank, fullsize=fullsize, img_min_rows=_hops.IMG_MIN_ROWS, losses_eager=losses1, losses_graph=losses2, sharded=sharded_before, own=own,
                                        stale_before_unshard=stale, n_plan=[k for k, _ in next(iter(graphed.records.values()))["cap"].plan],
                                        worst_max=max(v[0] for v in worst.values()), worst_mean=max(v[1] for k, v in worst.items() if not noise(k)),
                                        worst_mean_name=max((k for k in worst if not noise(k)), key=lambda k: worst[k][1]),
                                        worst_name=max(worst, key=lambda k: worst[k][0]), replica_spread=spread,
                                        stopped_at=stopped_at, resharded=resharded, stale_before_short_batch=stale2,
                                        short_was_eager=short_was_eager, replica_spread2=spread2,
                                        worst2_max=max(v[0] for v in worst2.values()),
                                        worst2_mean=max(v[1] for k, v in worst2.items() if not noise(k)),
                                        top2_mean=sorted(((round(v[1], 7), k) for k, v in worst2.items() if not noise(k)), reverse=True)[:6],
                                        top_mean=sorted(((round(v[1], 7), k) for k, v in worst.items() if not noise(k)), reverse=True)[:6])),
          flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
