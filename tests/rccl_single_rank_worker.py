"""Child process of tests/test_gpu_parity.py::test_gradsync_on_rccl_single_rank_group (a 1-rank RCCL group on cuda:0)."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    import hopmi
    from hopmi import steps
    from hopmi.parallel import GradSync
    from oracle import fill
    from oracle.golden_util import Accel, step_args
    from test_gpu_parity import _inputs, _make_model, _zero_grad_param
    dev = torch.device("cuda:0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        m1, bcfg = _make_model(9, dev)
        d1 = hopmi.ConvDiscriminator(27)
        d1.gru.dropout = 0.0
        fill.fill_state_(d1, salt=1)
        d1.to(dev)
        m2, d2 = copy.deepcopy(m1), copy.deepcopy(d1)
        inp = _inputs(9, bcfg, dev)
        steps._randn_like = lambda t: torch.randn(t.shape).to(t.device)
        steps._randperm = lambda n, device: torch.randperm(n).to(device)

        def run(m, d, acc):
            m.train(); d.train()
            g_opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999))
            d_opt = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999))
            torch.manual_seed(5)
            for _ in range(3):
                ret = hopmi.train_llm(step_args(9), 11, inp["in_audio"], inp["log_melspec"], inp["text"], inp["target_dir_vec"],
                                      inp["vid_indices"], m, d, g_opt, d_opt, acc)
            return ret

        sync = GradSync([m1, d1], bucket_mb=0.25, force=True)
        assert sync.active
        r1 = run(m1, d1, sync)
        r2 = run(m2, d2, Accel())
        assert sync.bytes_reduced > 0 and len(sync.groups[0].buckets) > 1
        # (not bitwise: under the exchange the GRU backward runs as per-time-step launches, whose partial sums are added
        # in another order than the persistent kernel's; 3 Adam steps of 1e-3 amplify that on analytically-zero gradients)
        assert sorted(r1) == sorted(r2)
        for k in r1:
            assert abs(r1[k] - r2[k]) <= 1e-4 * max(abs(r2[k]), 1e-6), (k, r1[k], r2[k])
        for (n, a), (_, b) in zip(list(m1.named_parameters()) + list(d1.named_parameters()),
                                  list(m2.named_parameters()) + list(d2.named_parameters())):
            # an element whose gradient is at rounding level may step the other way: <= 2 lr per step for few elements
            # (for every element of the analytically-zero-gradient biases)
            diff = (a - b).abs()
            assert diff.max().item() <= 6.5e-3, (n, diff.max().item())
            # (mean over the tensor: 1e-4 = 1.7 % of the element-steps taking the other sign of a rounding-level gradient at lr 1e-3 --
            # the bound the recorded-step comparison below uses; measured 6.7e-5 on gwnet.filter_convs.0.weight in round 5)
            assert _zero_grad_param(n) or diff.mean().item() <= 1e-4, (n, diff.mean().item())
        # ---- the recorded step's exchange on the same 1-rank group (force_exchange): gradients all-reduced IN PLACE under one RCCL
        # group call (`inplace=True`) and through the flat pack / all-reduce / unpack form (the default) must both reproduce the
        # recording without any exchange (a mean over one rank is the identity): same losses at every step, same parameters after
        steps._randn_like = lambda t: torch.full_like(t, 0.5)
        steps._randperm = lambda n, device: torch.arange(n - 1, -1, -1, device=device)
        from hopmi import graph as hgraph
        assert hgraph.inplace_group_ok(None)
        base_m, base_d = copy.deepcopy(m2), copy.deepcopy(d2)
        batch = (inp["in_audio"], inp["log_melspec"], inp["text"], inp["target_dir_vec"], inp["vid_indices"])

        def recorded(**kw):
            m, d = copy.deepcopy(base_m), copy.deepcopy(base_d)
            m._randn_like = lambda t: torch.full_like(t, 0.25)
            m.train(); d.train()
            g_opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999))
            d_opt = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999))
            step = hopmi.GraphedTrainStep(step_args(9), m, d, g_opt, d_opt, eager_calls=1, **kw)
            rets = [step(11, *batch) for _ in range(4)]
            assert step.n_replay == 3, step.n_replay
            kinds = [k for rec in step.records.values() for k, _ in rec["cap"].plan]
            return rets, m, d, kinds

        r_plain, m_p, d_p, k_plain = recorded()
        r_inpl, m_i, d_i, k_inpl = recorded(force_exchange=True, inplace=True)
        r_flat, m_f, d_f, k_flat = recorded(force_exchange=True)
        assert "eager" not in k_plain[:-1] or k_plain.count("eager") <= 1, k_plain      # (only the loss-fetch cut)
        assert k_inpl.count("eager") >= 4 and k_flat.count("eager") == k_inpl.count("eager"), (k_inpl, k_flat)
        for name, rr, mm, dd in (("in place", r_inpl, m_i, d_i), ("flat", r_flat, m_f, d_f)):
            for it, (a, b) in enumerate(zip(r_plain, rr)):
                assert sorted(a) == sorted(b), (name, it, a, b)
                for k in a:
                    assert abs(a[k] - b[k]) <= 2e-4 * max(abs(a[k]), 1e-6), (name, it, k, a[k], b[k])
            for (n, a), (_, b) in zip(list(m_p.named_parameters()) + list(d_p.named_parameters()),
                                      list(mm.named_parameters()) + list(dd.named_parameters())):
                diff = (a - b).abs()
                assert diff.max().item() <= 8.5e-3, (name, n, diff.max().item())
                assert _zero_grad_param(n) or diff.mean().item() <= 1e-4, (name, n, diff.mean().item())
    finally:
        dist.destroy_process_group()
    print("RCCL_WORKER_OK", flush=True)


if __name__ == "__main__":
    main()
