"""CPU, world_size 2 over gloo: the gradient exchange of hopmi.parallel.GradSync (the only
cross-rank step of the path) gives every rank the mean gradient, skips parameters that never get
a gradient, keeps discriminator-only backwards to the discriminator's buckets, and ends a step
with bit-identical parameters on all ranks."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 8)
        self.b = torch.nn.Linear(8, 4)
        self.unused = torch.nn.Linear(8, 8)          # never gets a gradient (like gwnet.residual_convs)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hopmi.parallel import GradSync
    torch.manual_seed(0)
    G, D = Toy(), Toy()
    sync = GradSync([G, D], bucket_mb=0.0002)        # tiny buckets: several per group
    g_opt = torch.optim.Adam(G.parameters(), lr=1e-2)
    d_opt = torch.optim.Adam(D.parameters(), lr=1e-2)
    res = {}
    for step in range(3):
        x = torch.randn(5, 8, generator=torch.Generator().manual_seed(100 * step + rank))
        # "D step": only D gets gradients
        d_opt.zero_grad()
        before = sync.bytes_reduced
        sync.backward(D(x).square().mean())
        res[f"d_bytes{step}"] = sync.bytes_reduced - before
        if step == 0:
            res["d_grad0"] = D.a.weight.grad.clone()
        d_opt.step()
        # "G step": loss reaches G and D
        g_opt.zero_grad()
        loss = (D(G(x)[:, :4].repeat(1, 2))).mean() + G(x).square().mean()
        sync.backward(loss)
        if step == 2:
            res["g_grad"] = G.a.weight.grad.clone()
        g_opt.step()
    res["unused_grad_none"] = G.unused.weight.grad is None
    # buckets follow the gradient PRODUCTION order of the planning backward: b (last layer) before a
    g_group = sync.groups[0]
    flat = [id(p) for b in g_group.buckets for p in b.params]
    res["order_ok"] = flat.index(id(G.b.weight)) < flat.index(id(G.a.weight))
    res["params"] = torch.cat([p.detach().flatten() for p in list(G.parameters()) + list(D.parameters())])
    # reference: mean of per-rank local gradients for the first D step, recomputed without sync
    torch.manual_seed(0)
    G2, D2 = Toy(), Toy()
    grads = []
    for r in range(world):
        x = torch.randn(5, 8, generator=torch.Generator().manual_seed(r))
        D2.zero_grad()
        D2(x).square().mean().backward()
        grads.append(D2.a.weight.grad.clone())
    res["d_expect0"] = sum(grads) / world
    q.put((rank, {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in res.items()}))   # by value, not by fd
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_gradsync_world2_gloo():
    world, port = 2, 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    r0, r1 = ({k: (torch.from_numpy(v) if hasattr(v, "dtype") else v) for k, v in out[r].items()} for r in (0, 1))
    assert torch.allclose(r0["d_grad0"], r0["d_expect0"], atol=1e-7)           # mean over ranks
    assert torch.equal(r0["d_grad0"], r1["d_grad0"])
    assert torch.equal(r0["g_grad"], r1["g_grad"])
    assert torch.equal(r0["params"], r1["params"])                              # replicas stay in lock-step
    assert r0["unused_grad_none"] and r1["unused_grad_none"]
    assert r0["order_ok"] and r1["order_ok"]
    d_bytes = sum(p.numel() for n, p in Toy().named_parameters() if not n.startswith("unused")) * 4
    assert r0["d_bytes1"] == d_bytes and r0["d_bytes2"] == d_bytes              # D step never moves G buckets


# ---------------------------------------------------------------------------------------------------------------------
# the real generator's parameter list (constructed on the CPU, no kernels run) through GradSync, and the prototype-row
# sharding of graph.GraphedTrainStep as plain tensor functions
NEVER_TRAINED = ("audio_encoder.", "gwnet.residual_convs.", "gwnet.bn.7.", "gwnet.gconv.7.")


def _skeleton_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hopmi
    from hopmi import graph
    from hopmi.parallel import GradSync
    from transformers import BertModel
    from oracle.golden_util import SynthTok, SynthVocab, hop_cfg, tiny_bert_config
    torch.manual_seed(0)
    bcfg = tiny_bert_config()
    G = hopmi.Model(hop_cfg(9, bcfg.hidden_size), BertModel(bcfg), SynthTok(), SynthVocab(11)).float()
    D = hopmi.ConvDiscriminator(27)
    sync = GradSync([G, D], bucket_mb=0.5)
    g_opt = torch.optim.Adam([p for p in G.parameters() if p.requires_grad], lr=1e-2)
    d_opt = torch.optim.Adam(D.parameters(), lr=1e-2)
    names = {id(p): n for n, p in list(G.named_parameters()) + [("D." + n, p) for n, p in D.named_parameters()]}
    live = [p for n, p in G.named_parameters() if p.requires_grad and not n.startswith(NEVER_TRAINED)]
    gen = torch.Generator().manual_seed(7 + rank)

    def fake_loss(params, reverse=False):
        # a stand-in for the step's loss: every live parameter gets a rank-dependent gradient, produced in (reversed)
        # registration order like a real backward would
        ps = list(reversed(params)) if reverse else params
        return sum((p * torch.randn(p.shape, generator=gen)).sum() for p in ps)

    res = {}
    for step in range(2):
        # discriminator step: only D's gradients exist and are exchanged
        d_opt.zero_grad()
        before = sync.bytes_reduced
        sync.backward(fake_loss(list(D.parameters())), only=(D,))
        res[f"d_bytes{step}"] = sync.bytes_reduced - before
        d_opt.step()
        # generator step: the loss reaches D too (gen_error), but only G's buckets are exchanged
        g_opt.zero_grad()
        before = sync.bytes_reduced
        sync.backward(fake_loss(live) + fake_loss(list(D.parameters())), only=(G,))
        res[f"g_bytes{step}"] = sync.bytes_reduced - before
        g_opt.step()
    res["plan"] = [[names[id(p)] for p in b.params] for b in sync.groups[0].buckets]
    res["never_none"] = all(p.grad is None for n, p in G.named_parameters() if n.startswith(NEVER_TRAINED))
    res["g_params"] = torch.cat([p.detach().flatten() for p in live]).double().sum().item()
    res["g_checks"] = [p.detach().double().sum().item() for p in live]
    res["g_live_bytes"] = sum(p.numel() * 4 for p in live)
    res["d_total_bytes"] = sum(p.numel() * 4 for p in D.parameters())

    # ---- prototype-row sharding (graph.shard_rows / all_gather_rows / mapping_grad_rows) --------------------------
    n_rows, vocab, dllm = 11, 40, 6                                 # 11 rows over 2 ranks: shards of 6 and 5 (padded to 6)
    torch.manual_seed(1)
    W, b, E = torch.randn(n_rows, vocab), torch.randn(n_rows), torch.randn(vocab, dllm)
    r0, r1, per = graph.shard_rows(n_rows, rank, world)
    S_pad = torch.zeros(per * world, dllm)
    S_pad[r0:r1] = W[r0:r1] @ E + b[r0:r1, None]
    graph.all_gather_rows(S_pad, per)
    S = S_pad[:n_rows].clone().requires_grad_()
    x = torch.randn(5, dllm, generator=torch.Generator().manual_seed(50 + rank))       # this rank's batch
    local = (torch.tanh(x @ S.t()) ** 2).mean()
    local.backward()
    flat = S.grad.clone()
    graph.all_reduce_mean(flat)
    Wg, bg = torch.zeros_like(W), torch.zeros_like(b)
    graph.mapping_grad_rows(flat, E, r0, r1, Wg, bg)
    # reference: the mean over ranks of the unsharded gradient
    Wr, br = W.clone().requires_grad_(), b.clone().requires_grad_()
    tot = 0
    for r in range(world):
        xr = torch.randn(5, dllm, generator=torch.Generator().manual_seed(50 + r))
        tot = tot + (torch.tanh(xr @ (Wr @ E + br[:, None]).t()) ** 2).mean() / world
    tot.backward()
    res["S_ok"] = bool(torch.allclose(S_pad[:n_rows], W @ E + b[:, None], atol=1e-5))
    res["rows"] = (r0, r1, per)
    res["dW_err"] = (Wg[r0:r1] - Wr.grad[r0:r1]).abs().max().item()
    res["db_err"] = (bg[r0:r1] - br.grad[r0:r1]).abs().max().item()
    res["outside_zero"] = bool((Wg[:r0] == 0).all() and (Wg[r1:] == 0).all())
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_real_generator_skeleton_and_prototype_sharding_world2_gloo():
    world, port = 2, 31500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_skeleton_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=200) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    a, b = out[0], out[1]
    assert a["plan"] == b["plan"] and len(a["plan"]) >= 2                       # same bucket plan on both ranks
    planned = [n for bucket in a["plan"] for n in bucket]
    assert len(planned) == len(set(planned)) and not any(n.startswith(NEVER_TRAINED) for n in planned)
    assert "mapping_layer.weight" in planned and "gru.weight_hh_l3_reverse" in planned
    assert a["never_none"] and b["never_none"]                                  # grad-less tensors stay grad-less
    assert a["g_checks"] == b["g_checks"]                                       # step-end parameters identical
    for r in (a, b):
        assert r["d_bytes0"] == r["d_total_bytes"] and r["d_bytes1"] == r["d_total_bytes"]      # D step: D buckets only
        assert r["g_bytes0"] == r["g_live_bytes"] and r["g_bytes1"] == r["g_live_bytes"]        # G step: G buckets only
        assert r["S_ok"] and r["outside_zero"]
        assert r["dW_err"] <= 1e-6 and r["db_err"] <= 1e-6, r
    assert a["rows"] == (0, 6, 6) and b["rows"] == (6, 11, 6)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` typed as such (no WORLD_SIZE): the process starts two fresh ranks under torch.distributed.run,
    they rendezvous on 127.0.0.1, run the timing reduction and rank 0's JSON line comes back through the parent with exit code 0
    (`--plumbing-check`: no GPU work -- this box has none).  Without the check the ranks stop at the "needs MI355X GPUs"
    assertion -- not at the WORLD_SIZE one that used to fire in the parent -- and the parent forwards a non-zero code."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "3", "--warmup", "1", "--plumbing-check"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_in_group"] == 2 and out["max_over_ranks"] == 2.0 and out["steps"] == 3
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                           text=True, timeout=240)
        assert r.returncode != 0
        assert "needs MI355X GPUs" in r.stderr and "WORLD_SIZE=" not in r.stderr
