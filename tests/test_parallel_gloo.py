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
