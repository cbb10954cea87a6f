"""hopmi.GraphedTrainStep (the recorded hipGraphs of train_llm) against steps.train_llm itself, on the device."""
import copy
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return torch.device("cuda:0")


def _pair(V, dev):
    import hopmi
    from transformers import BertModel
    from oracle import fill
    from oracle.golden_util import SynthTok, SynthVocab, hop_cfg, tiny_bert_config
    bcfg = tiny_bert_config()
    m = hopmi.Model(hop_cfg(V, bcfg.hidden_size), BertModel(bcfg), SynthTok(), SynthVocab(11)).float()
    m.reprogramming_layer.dropout.p = 0.0
    fill.fill_state_(m)
    m._randn_like = lambda t: torch.full_like(t, 0.25)               # capturable, deterministic speaker sample
    d = hopmi.ConvDiscriminator(3 * V)
    d.gru.dropout = 0.0
    fill.fill_state_(d, salt=1)
    inp = fill.hot_path_inputs(2, V, bcfg.vocab_size, 11)
    return m.to(dev).train(), d.to(dev).train(), {k: v.to(dev) for k, v in inp.items()}


def _deterministic_draws(monkeypatch):
    from hopmi import steps
    monkeypatch.setattr(steps, "_randn_like", lambda t: torch.full_like(t, 0.5))
    monkeypatch.setattr(steps, "_randperm", lambda n, device: torch.arange(n - 1, -1, -1, device=device))


@pytest.mark.parametrize("V,epoch", [(9, 0), (9, 11), (42, 11)])
def test_graphed_step_equals_eager(V, epoch, monkeypatch):
    """Five steps: train_llm itself on one copy, GraphedTrainStep (2 eager calls, 1 capture + replay, 2 replays) on the
    other, dropout off and the random draws replaced by constants: same loss dicts and same parameters after every step
    (the recorded step differs only in who forms the mapping layer's weight gradient and in Adam's capturable form)."""
    import hopmi
    from oracle.golden_util import Accel, step_args
    dev = _dev()
    _deterministic_draws(monkeypatch)
    m1, d1, inp = _pair(V, dev)
    m2, d2 = copy.deepcopy(m1), copy.deepcopy(d1)
    m2._randn_like = m1._randn_like
    mk = lambda m, d: (torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999)),
                       torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999)))
    g1, o1 = mk(m1, d1)
    g2, o2 = mk(m2, d2)
    args = step_args(V)
    batch = (inp["in_audio"], inp["log_melspec"], inp["text"], inp["target_dir_vec"], inp["vid_indices"])
    graphed = hopmi.GraphedTrainStep(args, m2, d2, g2, o2, eager_calls=2)
    for it in range(5):
        want = hopmi.train_llm(args, epoch, *batch, m1, d1, g1, o1, Accel())
        got = graphed(epoch, *batch)
        assert sorted(got) == sorted(want), (it, got, want)
        for k in want:
            assert abs(got[k] - want[k]) <= 2e-4 * max(abs(want[k]), 1e-6), (it, k, got[k], want[k])   # (Adam noise, see below)
    assert len(graphed.records) == 1
    for (n, a), (_, b) in zip(list(m1.named_parameters()) + list(d1.named_parameters()),
                              list(m2.named_parameters()) + list(d2.named_parameters())):
        # Adam turns rounding-level gradient differences into +-lr steps: an element whose gradient is at rounding level
        # may step the other way (5 steps of 1e-3: <= 1e-2 apart, few elements -- all elements for the analytically-zero
        # gradients of biases in front of a training-mode BatchNorm / the key-projection bias)
        # (the per-step losses above are the sharp check; this one catches a missing or doubled update)
        diff = (a - b).abs()
        assert diff.max().item() <= 1.1e-2 and diff.mean().item() <= 2.5e-3, (n, diff.max().item(), diff.mean().item())
    for (n, a), (_, b) in zip(m1.named_buffers(), m2.named_buffers()):
        if a.is_floating_point():
            # (a bias in front of a training-mode BatchNorm has an analytically zero gradient: Adam moves it by +-lr per
            # step on rounding noise, DESIGN.md 2, and it shifts that BatchNorm's batch mean by as much)
            # (and a few weight elements whose gradient is at rounding level step the other way -- see above: the batch
            # variances downstream of them move by ~1e-3 relative)
            slack = 6e-3 if n.endswith("running_mean") else 0.0
            rtol = 2e-3 if n.endswith("running_var") else 1e-4
            assert (a - b).abs().max().item() <= rtol * max(a.abs().max().item(), 1.0) + slack, n
        else:
            assert torch.equal(a, b), n


def _copy_state_(src_modules, dst_modules, src_opts, dst_opts):
    """Parameters, buffers and Adam state of the eager twin INTO the recorded one, in place (a replay reads and writes
    static storage): after it both sides start the next step from the same state."""
    with torch.no_grad():
        for ms, md in zip(src_modules, dst_modules):
            for a, b in zip(ms.parameters(), md.parameters()):
                if a.requires_grad:                          # (a modified frozen parameter drops the recordings: test below)
                    b.copy_(a)
            for a, b in zip(ms.buffers(), md.buffers()):
                b.copy_(a)
        for os_, od in zip(src_opts, dst_opts):
            ps = [p for g in os_.param_groups for p in g["params"]]
            pd = [p for g in od.param_groups for p in g["params"]]
            for a, b in zip(ps, pd):
                sa, sb = os_.state.get(a, {}), od.state.get(b, {})
                if not sa:                                   # never had a gradient: the recorded side's state is still its zeros
                    continue
                assert sorted(sa) == sorted(sb), (sorted(sa), sorted(sb))
                for k, v in sa.items():
                    if torch.is_tensor(sb[k]):
                        sb[k].copy_(v if torch.is_tensor(v) else torch.tensor(float(v)))
                    else:
                        sb[k] = v


def test_graphed_step_bf16_follows_eager(monkeypatch):
    """The bf16 mode through the recording, step by step: five steps of train_llm on one copy and of GraphedTrainStep (one eager
    call, recording + replay, three replays) on the other, the recorded copy's state (parameters, buffers, Adam moments and step
    counts) RESET to the eager copy's after every step.  From equal states the forwards are the same kernels on the same numbers,
    so the loss dicts must agree to rounding; the updated parameters differ only by Adam's two forms (fused / capturable) turning
    rounding-level gradients into +-lr steps.  (Letting the two copies run free instead compares two chaotic trajectories: in
    bf16 a +-lr step moves a weight across bf16 roundings and the losses drift 1 % apart within three GAN-phase steps, by an
    amount that changes with every kernel's summation order.)"""
    import hopmi
    from oracle.golden_util import Accel, step_args, zero_grad_param
    dev = _dev()
    _deterministic_draws(monkeypatch)
    m1, d1, inp = _pair(9, dev)
    m2, d2 = copy.deepcopy(m1), copy.deepcopy(d1)
    m2._randn_like = m1._randn_like
    lr_g, lr_d = 1e-3, 1e-4
    mk = lambda m, d: (torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=lr_g, betas=(0.5, 0.999)),
                       torch.optim.Adam(d.parameters(), lr=lr_d, betas=(0.5, 0.999)))
    g1, o1 = mk(m1, d1)
    g2, o2 = mk(m2, d2)
    args = step_args(9)
    batch = (inp["in_audio"], inp["log_melspec"], inp["text"], inp["target_dir_vec"], inp["vid_indices"])
    prev = hopmi.mixed_precision("bf16")
    try:
        graphed = hopmi.GraphedTrainStep(args, m2, d2, g2, o2, eager_calls=1)
        for it in range(5):
            want = hopmi.train_llm(args, 11, *batch, m1, d1, g1, o1, Accel())
            got = graphed(11, *batch)
            assert sorted(got) == sorted(want), (it, got, want)
            for k in want:
                # (the GAN phase updates the discriminator before the generator's forward: its Adam noise -- +-1e-4 on a few
                # weights -- is inside this step's generator losses already)
                # -- 'gen' (the adversarial term, read through the freshly stepped discriminator) and the total that carries it
                # five-fold see that noise; every other term is computed from the common state)
                tol = {"DIV_REG": 2e-2, "gen": 1e-2, "loss": 5e-3}.get(k, 1e-3)
                assert abs(got[k] - want[k]) <= tol * max(abs(want[k]), 1e-6), (it, k, got[k], want[k])
            # (no_grad: a tensor with a graph over the parameters that is still alive when the step is recorded keeps their
            # gradient accumulators -- created here, on the default stream -- and the recorded backward would run them there:
            # a legacy-stream launch inside a capture, which ends the process in hipStreamEndCapture)
            with torch.no_grad():
                for mods, lr in (((m1, m2), lr_g), ((d1, d2), lr_d)):
                    for (n, a), (_, b) in zip(mods[0].named_parameters(), mods[1].named_parameters()):
                        diff = (a - b).abs()
                        # one step from equal states: an element whose gradient sits at rounding level may take its step the other
                        # way on one side (a step is lr m / sqrt(v): a few lr while v still lags m, beta2 = 0.999); the mean says
                        # that these are few
                        # (analytically-zero gradients -- a bias in front of a training-mode BatchNorm, the key-projection bias --
                        # are rounding noise in EVERY element: Adam moves all of them by +-lr on both sides)
                        mean_ok = (2.0 if zero_grad_param(n) else 0.1) * lr
                        assert diff.max().item() <= 10 * lr and diff.mean().item() <= mean_ok, (it, n, diff.max().item(), diff.mean().item())
            _copy_state_((m1, d1), (m2, d2), (g1, o1), (g2, o2))
        assert graphed.n_replay == 4
    finally:
        hopmi.mixed_precision(prev)


def test_graphed_step_rerecords_after_a_frozen_weight_changes(monkeypatch):
    """A recording replays tensors derived from the frozen parameters (the fused QKV weight, bf16 copies / part images of the
    BERT's weights: built once, in an eager call): after an in-place change of a frozen weight (what load_state_dict does) the
    next call must NOT replay -- it runs the eager step again, re-records, and keeps following an eager twin that received the
    same change.  A third copy with the check switched off shows what the check is for: its replay misses the change."""
    import hopmi
    from oracle.golden_util import Accel, step_args
    dev = _dev()
    _deterministic_draws(monkeypatch)
    m1, d1, inp = _pair(9, dev)
    m2, d2 = copy.deepcopy(m1), copy.deepcopy(d1)
    m3, d3 = copy.deepcopy(m1), copy.deepcopy(d1)
    m2._randn_like = m3._randn_like = m1._randn_like
    mk = lambda m, d: (torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999)),
                       torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999)))
    g1, o1 = mk(m1, d1)
    g2, o2 = mk(m2, d2)
    g3, o3 = mk(m3, d3)
    args = step_args(9)
    batch = (inp["in_audio"], inp["log_melspec"], inp["text"], inp["target_dir_vec"], inp["vid_indices"])
    graphed = hopmi.GraphedTrainStep(args, m2, d2, g2, o2, eager_calls=1)
    unchecked = hopmi.GraphedTrainStep(args, m3, d3, g3, o3, eager_calls=1)
    unchecked._frozen_versions = lambda: ()
    stale = None
    for it in range(7):
        if it == 3:
            for m in (m1, m2, m3):
                with torch.no_grad():
                    w = m.llm_model.encoder.layer[0].attention.self.query.weight
                    assert not w.requires_grad
                    w.mul_(-3.0)
        want = hopmi.train_llm(args, 0, *batch, m1, d1, g1, o1, Accel())
        got = graphed(0, *batch)
        for k in want:
            assert abs(got[k] - want[k]) <= 4e-4 * max(abs(want[k]), 1e-6), (it, k, got[k], want[k])
        # calls: 0 eager, 1 record + replay, 2 replay, 3 eager again (recordings dropped), 4 record + replay, 5-6 replay
        assert (graphed.n_eager, graphed.n_replay) == [(1, 0), (1, 1), (1, 2), (2, 2), (2, 3), (2, 4), (2, 5)][it], it
        if it <= 3:
            miss = unchecked(0, *batch)
            if it == 3:
                stale = abs(miss["loss"] - want["loss"]) / abs(want["loss"])
    assert unchecked.n_eager == 1 and stale > 4e-4, stale       # (the stale replay is off by more than the tolerance above)


def test_graphed_step_rerecords_after_the_caches_are_invalidated(monkeypatch):
    """ops.invalidate_weight_images() (= reset_all_caches("all")) drops cached images of FROZEN weights too, and a recording holds
    their addresses: the recorded step keeps ops.CACHE_EPOCH beside the frozen parameters' version counters, so the call after an
    invalidation runs the eager step and records again instead of replaying freed memory -- and after a frozen weight was
    rescaled through `.data` (no version counter moves) followed by the documented invalidate call, it follows an eager twin."""
    import hopmi
    from hopmi import ops
    from oracle.golden_util import Accel, step_args
    dev = _dev()
    _deterministic_draws(monkeypatch)
    m1, d1, inp = _pair(9, dev)
    m2, d2 = copy.deepcopy(m1), copy.deepcopy(d1)
    m2._randn_like = m1._randn_like
    mk = lambda m, d: (torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999)),
                       torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999)))
    g1, o1 = mk(m1, d1)
    g2, o2 = mk(m2, d2)
    args = step_args(9)
    batch = (inp["in_audio"], inp["log_melspec"], inp["text"], inp["target_dir_vec"], inp["vid_indices"])
    graphed = hopmi.GraphedTrainStep(args, m2, d2, g2, o2, eager_calls=1)
    for it in range(6):
        if it == 3:
            for m in (m1, m2):
                m.llm_model.encoder.layer[0].intermediate.dense.weight.data.mul_(2.0)      # behind every version counter's back
            ops.invalidate_weight_images()
        want = hopmi.train_llm(args, 0, *batch, m1, d1, g1, o1, Accel())
        got = graphed(0, *batch)
        for k in want:
            assert abs(got[k] - want[k]) <= 4e-4 * max(abs(want[k]), 1e-6), (it, k, got[k], want[k])
        assert (graphed.n_eager, graphed.n_replay) == [(1, 0), (1, 1), (1, 2), (2, 2), (2, 3), (2, 4)][it], it


def test_graphed_step_new_batches_and_other_shapes(monkeypatch):
    """Replays read the batch from static buffers (a new batch of the recorded shape is copied in), another batch size
    falls back to the eager step, and both keep training the same model."""
    import hopmi
    from oracle.golden_util import Accel, step_args
    dev = _dev()
    _deterministic_draws(monkeypatch)
    m1, d1, inp = _pair(9, dev)
    m2, d2 = copy.deepcopy(m1), copy.deepcopy(d1)
    m2._randn_like = m1._randn_like
    mk = lambda m, d: (torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999)),
                       torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999)))
    g1, o1 = mk(m1, d1)
    g2, o2 = mk(m2, d2)
    args = step_args(9)
    graphed = hopmi.GraphedTrainStep(args, m2, d2, g2, o2, eager_calls=1)
    names = ("in_audio", "log_melspec", "text", "target_dir_vec", "vid_indices")
    gen = torch.Generator().manual_seed(5)
    for it in range(5):
        if it == 3:                                   # a short batch
            b = tuple(inp[k][:1].clone() for k in names)
        else:                                         # fresh tensors of the recorded shape
            b = tuple((inp[k] + 0.01 * it * torch.randn(inp[k].shape, generator=gen).to(dev)) if inp[k].is_floating_point()
                      else inp[k].roll(it, 0) for k in names)
        want = hopmi.train_llm(args, 0, *b, m1, d1, g1, o1, Accel())
        got = graphed(0, *b)
        for k in want:
            assert abs(got[k] - want[k]) <= 2e-4 * max(abs(want[k]), 1e-6), (it, k, got[k], want[k])   # (Adam noise, see below)
    assert len(graphed.records) == 1


def test_inference_between_graphed_steps_sees_the_trained_weights(monkeypatch):
    """A no-grad forward keeps its K/V prototypes across calls until a weight changes (Model._kv_infer, keyed on the
    parameters' version counters).  Replays of the recorded step update the weights on the device without touching those
    counters: an evaluation between replays must still see the current weights (= the same forward with the cache
    dropped), not the prototypes of the evaluation before."""
    import hopmi
    from oracle.golden_util import step_args
    dev = _dev()
    _deterministic_draws(monkeypatch)
    m, d, inp = _pair(9, dev)
    g_opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-2, betas=(0.5, 0.999))
    d_opt = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999))
    batch = (inp["in_audio"], inp["log_melspec"], inp["text"], inp["target_dir_vec"], inp["vid_indices"])
    graphed = hopmi.GraphedTrainStep(step_args(9), m, d, g_opt, d_opt, eager_calls=1)

    def evaluate():
        m.eval()
        with torch.no_grad():
            out = m(inp["in_audio"], inp["log_melspec"], inp["text"], inp["target_dir_vec"][:, 0:16], inp["vid_indices"])
        m.train()
        return (out[0] if isinstance(out, (tuple, list)) else out).clone()

    graphed(0, *batch)                                 # eager call
    graphed(0, *batch)                                 # capture + first replay
    before = evaluate()                                # fills the prototype cache
    for _ in range(3):
        graphed(0, *batch)                             # replays only
    assert graphed.n_replay >= 4
    after = evaluate()
    m._kv_infer = None
    fresh = evaluate()
    assert torch.equal(after, fresh)
    assert not torch.equal(after, before)


def test_seed_word_advances_dropout_masks():
    """The seeded kernels add *ops.SEED_DEV to their seed: same word -> same mask, another word -> another mask, and the
    backward regenerates the forward's mask (gradient of sum(o) w.r.t. v counts the kept keys)."""
    from hopmi import ops
    dev = _dev()
    g = torch.Generator().manual_seed(2)
    q = torch.randn(2, 34, 8, 128, generator=g).to(dev)
    k = torch.randn(96, 8, 128, generator=g).to(dev)
    v = torch.randn(96, 8, 128, generator=g).to(dev)
    prev = ops.SEED_DEV
    try:
        ops.SEED_DEV = torch.zeros(1, dtype=torch.int64, device=dev)
        a = ops.reprog_attention(q, k, v, 0.1, 0.3, 1234)
        ops.SEED_DEV = None
        a0 = ops.reprog_attention(q, k, v, 0.1, 0.3, 1234)
        assert torch.equal(a, a0)                                      # word 0 == no word
        ops.SEED_DEV = torch.full((1,), 77, dtype=torch.int64, device=dev)
        b = ops.reprog_attention(q, k, v, 0.1, 0.3, 1234)
        b2 = ops.reprog_attention(q, k, v, 0.1, 0.3, 1234)
        c = ops.reprog_attention(q, k, v, 0.1, 0.3, 1234 + 77)
        assert torch.equal(b, b2) and not torch.equal(a, b)
        ops.SEED_DEV = None
        c0 = ops.reprog_attention(q, k, v, 0.1, 0.3, 1234 + 77)
        assert torch.equal(b, c0)                                      # seed + word is what is hashed
        # BERT epilogue + attention honour the word as well
        x = torch.randn(40, 768, generator=g).to(dev)
        res = torch.randn(40, 768, generator=g).to(dev)
        one, zero = torch.ones(768, device=dev), torch.zeros(768, device=dev)
        e0 = ops.bias_dropout_residual_layernorm(x, zero, res, one, zero, 1e-12, 0.2, 9)
        ops.SEED_DEV = torch.full((1,), 5, dtype=torch.int64, device=dev)
        e1 = ops.bias_dropout_residual_layernorm(x, zero, res, one, zero, 1e-12, 0.2, 9)
        ops.SEED_DEV = None
        e2 = ops.bias_dropout_residual_layernorm(x, zero, res, one, zero, 1e-12, 0.2, 14)
        assert not torch.equal(e0, e1) and torch.equal(e1, e2)
        qkv = torch.randn(2, 34, 3, 12, 64, generator=g).to(dev)
        f0 = ops.bert_attention(qkv, 0.2, 3)
        ops.SEED_DEV = torch.full((1,), 8, dtype=torch.int64, device=dev)
        f1 = ops.bert_attention(qkv, 0.2, 3)
        ops.SEED_DEV = None
        f2 = ops.bert_attention(qkv, 0.2, 11)
        assert not torch.equal(f0, f1) and torch.equal(f1, f2)
        # backward regenerates the forward's mask under a non-zero word
        ops.SEED_DEV = torch.full((1,), 31, dtype=torch.int64, device=dev)
        qq, vv = q.clone().requires_grad_(), v.clone().requires_grad_()
        o = ops.reprog_attention(qq, k, vv, 0.1, 0.3, 99)
        o.sum().backward()
        ops.SEED_DEV = None
        qq2, vv2 = q.clone().requires_grad_(), v.clone().requires_grad_()
        o2 = ops.reprog_attention(qq2, k, vv2, 0.1, 0.3, 99 + 31)
        o2.sum().backward()
        assert torch.equal(o, o2) and torch.equal(vv.grad, vv2.grad) and torch.equal(qq.grad, qq2.grad)
    finally:
        ops.SEED_DEV = prev


def test_graphed_step_dropout_advances(monkeypatch):
    """With dropout on, consecutive replays of the recorded step draw different masks: at learning rate 0 (weights and
    batch fixed) the loss still changes from replay to replay, and stays finite."""
    import hopmi
    from oracle.golden_util import step_args
    dev = _dev()
    _deterministic_draws(monkeypatch)
    m, d, inp = _pair(9, dev)
    m.reprogramming_layer.dropout.p = 0.3
    g_opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=0.0, betas=(0.5, 0.999))
    d_opt = torch.optim.Adam(d.parameters(), lr=0.0, betas=(0.5, 0.999))
    graphed = hopmi.GraphedTrainStep(step_args(9), m, d, g_opt, d_opt, eager_calls=1)
    batch = (inp["in_audio"], inp["log_melspec"], inp["text"], inp["target_dir_vec"], inp["vid_indices"])
    # BatchNorm running statistics move, but training-mode forwards do not read them: only the masks change the loss
    losses = [graphed(0, *batch)["loss"] for _ in range(6)]
    assert all(l == l and abs(l) < 1e9 for l in losses)
    assert len(set(losses[1:])) >= 4, losses


def test_graphed_exchange_two_ranks_one_gpu_image_paths():
    """The same two-rank comparison at BERT-base geometry and 96 clips per rank (3 264 rows >= ops.IMG_MIN_ROWS): the operand-image
    producers, the LDS-DMA GEMM form, the TN weight gradients and every version-keyed operand cache run beside the row-sharded
    exchange (whose Adam hook and _mark_written move version counters outside the optimizer).  Held to: the same loss dict as the
    GradSync path every step, replicas identical across the ranks after unshard(), the shard bookkeeping of the small case."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29777", os.path.join(root, "tests", "graph_rank_worker.py"), "0", "sequential", "overlap", "fullsize"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    dec = json.JSONDecoder()
    rows = [dec.raw_decode(chunk)[0] for chunk in r.stdout.split("RANKJSON ")[1:]]
    assert len(rows) == 2, r.stdout[-2000:]
    for row in rows:
        assert row["fullsize"] and row["img_min_rows"] <= 3264, row["img_min_rows"]
        assert row["sharded"] and row["own"] in ([0, 750], [750, 1500]), row
        assert row["stale_before_unshard"] > 1e-4
        for a, b in zip(row["losses_eager"], row["losses_graph"]):
            assert sorted(a) == sorted(b)
            for k in a:
                assert abs(a[k] - b[k]) <= 5e-4 * max(abs(a[k]), 1e-6), (k, a[k], b[k])
        assert row["replica_spread"] <= 1e-6 and row["replica_spread2"] <= 1e-6, (row["replica_spread"], row["replica_spread2"])
        assert row["resharded"] and row["short_was_eager"], row
        # (a skipped or doubled update of a tensor would show as ~lr = 1e-3 on its mean)
        assert row["worst_mean"] <= 5e-4 and row["worst2_mean"] <= 8e-4, {k: row[k] for k in ("worst_mean", "worst_mean_name", "worst2_mean", "top2_mean")}
        assert row["stopped_at"] == 1, row["stopped_at"]


@pytest.mark.parametrize("epoch,order,exchange", [(0, "sequential", "overlap"), (11, "sequential", "overlap"), (0, "interleaved", "overlap"),
                                                  (11, "interleaved", "overlap"), (0, "sequential", "flat")])
def test_graphed_exchange_two_ranks_one_gpu(epoch, order, exchange):
    """The N > 1 recording (prototype rows sharded over ranks, S all-gathered, dS + the other gradients all-reduced by
    eager collectives between graph launches) with two ranks sharing cuda:0 over gloo, against the hook-driven GradSync
    exchange under steps.train_llm on copies of the same models: same losses every step, same parameters after
    unshard(), and identical replicas across the ranks (tests/graph_rank_worker.py).  `order`: the two paths one after the
    other, or an eager step of the other model copy queued between every two replays (the order in which round 2 saw a wrong
    gradient).  Then the sequence replay / unshard() / replay / short batch through the eager step / replay / unshard():
    every replay shards the mapping layer again, the eager step and unshard() must notice, the replicas stay identical.
    `exchange`: "overlap" = the generator's backward recorded in two halves, cut at the decoder's input, the first half's
    all-reduce started between two graph launches and finished behind the second half (the default for N > 1); "flat" = one
    all-reduce behind the whole backward."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + epoch + (50 if order == "interleaved" else 0) + (100 if exchange == "flat" else 0)),
           os.path.join(root, "tests", "graph_rank_worker.py"), str(epoch), order, exchange]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    dec = json.JSONDecoder()          # (the two ranks' lines can arrive glued together)
    rows = [dec.raw_decode(chunk)[0] for chunk in r.stdout.split("RANKJSON ")[1:]]
    assert len(rows) == 2, r.stdout[-2000:]
    for row in rows:
        assert row["sharded"] and row["own"] in ([0, 750], [750, 1500]), row
        assert "eager" in row["n_plan"] and row["n_plan"].count("graph") >= (5 if exchange == "overlap" else 4), row["n_plan"]
        assert row["stale_before_unshard"] > 1e-4                       # the other rank's rows really were not updated here
        for a, b in zip(row["losses_eager"], row["losses_graph"]):
            assert sorted(a) == sorted(b)
            for k in a:
                assert abs(a[k] - b[k]) <= 2e-4 * max(abs(a[k]), 1e-6), (k, a[k], b[k])
        # (parameters with analytically zero gradients are frozen in the worker; single elements of the others whose gradient
        # is at rounding level may still take a +-lr Adam step the other way)
        assert row["worst_max"] <= 4.5e-3 and row["worst_mean"] <= 2e-5, {k: row[k] for k in ("worst_max", "worst_mean", "worst_mean_name", "worst_name", "top_mean")}
        assert row["replica_spread"] <= 1e-6, row
        # a status word set on rank 1 alone stopped BOTH ranks, at the same call (rank 0 learns of it from the exchange only)
        assert row["stopped_at"] == (0 if epoch > 10 else 1), row["stopped_at"]
        # phase 2: the replays after the first unshard() sharded the copies again, and both later consumers noticed
        assert row["resharded"] and row["stale_before_short_batch"] > 1e-4 and row["short_was_eager"], row
        assert row["replica_spread2"] <= 1e-6, row["replica_spread2"]
        # (7 more free-running steps, one of them at batch 1: the two paths' Adam noise on rounding-level gradients compounds -- the
        # per-step losses above are the sharp check, this one catches a missing or doubled update, which would show as ~lr = 1e-3)
        # (measured 0.9-1.2e-4 on all of the audio branch's tensors alike; a skipped step would be 1e-3 on one of them)
        assert row["worst2_max"] <= 9e-3 and row["worst2_mean"] <= 3e-4, {k: row[k] for k in ("worst2_max", "worst2_mean", "top2_mean")}


@pytest.mark.parametrize("mode", ["pyerr", "sync", "devsync"])
def test_recording_failure_is_not_fatal(mode):
    """An exception while a step is being recorded (a Python error under a healthy capture; an illegal host read-back; a
    device-wide synchronize, which INVALIDATES the capture) comes back as that exception, the object gives up recording
    (`enabled` False, `broken` says why) and keeps training through the eager step with train_llm's results, and the process
    exits cleanly -- in a child process, so that whatever the runtime does is a return code here (graph.EXIT_CAPTURE_FAILED = 86
    would mean: no device work was possible after the failed recording; -6: an abort at interpreter exit, what a capture left
    open does)."""
    import json
    from conftest import ROOT, run_isolated
    r = run_isolated([os.path.join(ROOT, "tests", "capture_failure_worker.py"), mode], timeout=240)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    res = json.loads(r.stdout.split("RESULT ")[-1])
    assert res["raised"] in (("ValueError",) if mode == "pyerr" else ("AcceleratorError", "RuntimeError")), res
    assert res["enabled"] is False and res["broken"], res
    assert all(res["losses_match"]) and res["n_replay"] == 0 and res["n_eager"] == 4, res
    assert res["after"] == 32 * 1024 * 1024
    assert "recording abandoned" in r.stderr


# ----------------------------------------------------------------- the recorded step at the sizes bench.py times
class _Draws:
    """The step's random draws as STATIC device tensors: a recording reads them by address, so each call's values are put
    there before the call, taken from torch's CPU generator in the oracle's order (ref_cpu.train_llm_step: per generator
    forward one (B,16) `eps`; in the GAN phase two `noise` tensors for add_noise; one `perm`)."""

    def __init__(self, B, V, gan, dev):
        self.spec = ([("eps", (B, 16)), ("noise", (B, 34, 3 * V)), ("noise", (B, 34, 3 * V))] if gan else []) + \
                    [("eps", (B, 16)), ("perm", (B,)), ("eps", (B, 16))]
        self.slots = [torch.empty(shape, dtype=torch.int64 if kind == "perm" else torch.float32, device=dev) for kind, shape in self.spec]
        self.i = 0

    def refill(self):
        for (kind, shape), t in zip(self.spec, self.slots):
            t.copy_(torch.randperm(shape[0]) if kind == "perm" else torch.randn(shape))
        self.i = 0

    def take(self, kind, shape):
        k, want = self.spec[self.i]
        assert k == kind and tuple(shape) == tuple(want), (self.i, kind, tuple(shape), self.spec[self.i])
        t = self.slots[self.i]
        self.i += 1
        return t

    def install(self, model, monkeypatch):
        from hopmi import steps
        model._randn_like = lambda t: self.take("eps", t.shape)
        monkeypatch.setattr(steps, "_randn_like", lambda t: self.take("noise", t.shape))
        monkeypatch.setattr(steps, "_randperm", lambda n, device: self.take("perm", (n,)))


@pytest.mark.parametrize("V,B,epoch", [(9, 128, 0), (42, 64, 11)])
def test_graphed_step_baseline_size_vs_reference(V, B, epoch, monkeypatch):
    """What bench.py times, against the REAL reference: hopmi.GraphedTrainStep at BASELINE.json configs[1] (TED, B = 128, epoch 0)
    and configs[3] in the GAN phase (TED-Expressive, V = 42, B = 64, epoch 11) with bench.py's set-up -- fused Adam (its
    capturable form under the recording), the shipped TunableOp GEMM table, one eager call (bench.py's eager_calls = 1), then
    the recording and three more replays.  The reference advanced the same five steps in the build container from the same random
    stream (tests/golden/train_llm_full_*.npz, tools/make_golden.py::golden_step_full).  Compared as
    test_train_llm_baseline_size_vs_reference compares the eager step: every step's loss dict, every recorded step's graded
    outputs, the BatchNorm running statistics after all forwards of all steps, post-step checksums of generator and discriminator
    parameters."""
    import hopmi
    from oracle.golden_util import step_args
    from test_gpu_parity import RTOL, _full_golden, _full_size_setup, assert_close
    dev = _dev()
    n_steps = 5
    m, d, bcfg, inp = _full_size_setup(V, B)
    g = _full_golden(V, B, epoch)
    assert g.n_steps == n_steps
    m.to(dev).train(); d.to(dev).train()
    gan = epoch > 10
    draws = _Draws(B, V, gan, dev)
    draws.install(m, monkeypatch)
    g_opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999), fused=True)
    d_opt = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999), fused=True)
    gin = {k: v.to(dev) for k, v in inp.items()}
    batch = (gin["in_audio"], gin["log_melspec"], gin["text"], gin["target_dir_vec"], gin["vid_indices"])
    # the graded forward's output tensor as the recording holds it (rewritten in place by every replay): what the loss op is
    # handed while the step is being recorded.  (A forward hook on the model that keeps the eager call's output alive made the
    # recording fail -- "operation would make the legacy stream depend on a capturing blocking stream" -- so nothing of the
    # eager call is kept.)
    from hopmi import ops as hops, steps as hsteps
    graded = []
    real_losses = hops.hop_losses

    def spy_losses(out, *a, **k):
        if hsteps._CAPTURE is not None:
            graded.append(out.detach())
        return real_losses(out, *a, **k)

    monkeypatch.setattr(hops, "hop_losses", spy_losses)
    tuned = hopmi.use_tuned_gemms()
    assert tuned, "the shipped TunableOp table is missing"
    graphed = hopmi.GraphedTrainStep(step_args(V), m, d, g_opt, d_opt, eager_calls=1)
    torch.manual_seed(777)
    rets, eps = [], []
    try:
        for it in range(n_steps):
            draws.refill()
            rets.append(graphed(epoch, *batch))
            if graded:                                  # the recording's static output tensor, as this step left it
                torch.cuda.synchronize()
                eps.append((it, g.out_err(graded[0], it)))
    finally:
        import torch.cuda.tunable as tunable
        tunable.enable(False)
    assert graphed.n_eager == 1 and graphed.n_replay == n_steps - 1 and len(graded) == 1 and len(eps) == n_steps - 1
    # Tolerance per step.  Step 1 is the north_star's 1e-3 (what test_train_llm_baseline_size_vs_reference holds the eager step
    # to).  From step 2 on the forward sees parameters that Adam has moved: an element whose gradient is at rounding level steps
    # by +-lr in a direction that rounding decides, differently on the two sides (tools/probes/grad_sensitivity.py),
    # and the outputs drift apart by a few 1e-4 per step on that account alone.  Rounds 3-4 (three-term split-bf16 products, ~2^-16,
    # in the WaveNet / attention / GRU kernels) measured 1.7e-3 / 1.9e-3 at step 5 and allowed 1e-3 + 4e-4 per Adam step; with
    # every contraction fp32-class (round 5) the same runs measure 2.7e-4, 6.1e-4, 2.3e-4, 4.6e-4 (V = 9) and 2.2e-4, 3.1e-4, 4.6e-4,
    # 8.6e-4 (V = 42) at steps 2-5 (gpurun_out/graphed_vs_reference_drift.txt): about half of the old drift was the narrow
    # products', the rest is Adam's.  Allowed now: 1e-3 + 1.5e-4 per Adam step taken.  That the recorded step itself adds nothing
    # to the eager step's error is the soak test's business (bit-identical gradients, 52 steps, this size); a wrong launch, a
    # stale buffer or a reset optimizer state shows up here as per cent, not as 1e-3.
    tol = lambda it: RTOL + 1.5e-4 * it
    # (the BatchNorm running statistics and the parameters after five Adam steps keep the allowance of rounds 3-4: they integrate the
    # sign-undetermined steps of the rounding-level gradients directly -- running_var of V = 42 measures 1.0e-3 elementwise at step 5)
    tol_state = lambda it: RTOL + 4e-4 * it
    try:                                               # (the measured drift, kept for DESIGN.md 2)
        from conftest import ROOT
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "graphed_vs_reference_drift.txt"), "a") as f:
            f.write(f"V={V} B={B} epoch={epoch} output rel err per step: {[(i + 1, float(f'{e:.3e}')) for i, e in eps]}\n")
    except OSError:
        pass
    for it, e in eps:
        assert e <= tol(it), f"outputs of step {it + 1}: rel err {e:.3e} > {tol(it):.1e}; all steps: {eps}"
    by_step = dict(eps)
    for it, (ret, want) in enumerate(zip(rets, g.rets)):
        assert sorted(ret.keys()) == sorted(want.keys()), (it, ret, want)
        div_tol, cond = g.div_reg_tol(by_step.get(it, RTOL), it)
        for k in want:
            t = tol(it) if k != "DIV_REG" else div_tol
            assert abs(ret[k] - want[k]) <= t * max(abs(want[k]), 1e-6), (it, k, ret[k], want[k], f"eps {eps} cond {cond:.1f}")
    sd = m.state_dict()
    for k, v in g.bn["last"].items():
        if k.endswith("running_mean"):
            # the graph-conv bias in front of this BatchNorm has an analytically zero gradient: Adam walks it by +-lr per step on
            # rounding noise, differently on the two sides, and the batch mean moves with it (the normalised output does not)
            diff = (sd[k].cpu() - v).abs().max().item()
            assert diff <= tol_state(n_steps - 1) * v.abs().max().item() + n_steps * 1e-3, (k, diff)
        else:
            assert_close(sd[k], v, tol_state(n_steps - 1), what=k)
    # (the absolute slack of n_steps sign-undetermined Adam steps is given to the analytically-zero-gradient tensors only)
    g.check_params(sd, d.state_dict(), "last", tol_state(n_steps - 1), n_steps, gan)
    if gan:
        for p in d.parameters():
            assert float(d_opt.state[p]["step"]) == n_steps
    for p in m.parameters():
        if p.requires_grad and p.grad is not None:
            assert float(g_opt.state[p]["step"]) == n_steps


def _bench_regime(extra_args=(), env=None, timeout=400):
    """tools/nan_hunt.py as a child process: bench.py's exact set-up with a finiteness census after every step -> its RESULT dict."""
    import json
    from conftest import ROOT, run_isolated
    r = run_isolated([os.path.join(ROOT, "tools", "nan_hunt.py")] + list(extra_args), timeout=timeout, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert r.returncode == 0 and lines, f"rc {r.returncode}\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    return json.loads(lines[-1][len("RESULT "):]), r.stdout


def test_graphed_step_bench_regime_stays_finite():
    """The regime bench.py times and no other test ran: default initialisation under seed 0, the reference's
    learning rate 1e-2 (run_ted.py:103), dropout ON (HOP.py:256,296; the BERT in train mode), fused Adam, the shipped GEMM table,
    7 eager steps (6 of them instrumented, as bench.py's kernel region), the recorder's eager call and then 35 replays of the recorded step at configs[1] size.
    After EVERY step: all losses, all trainable parameters, all gradients and all buffers finite.  The trajectory is chaotic at
    this learning rate (variants of the arithmetic end anywhere between 20 and 200 after 40 steps), so the comparison with the
    twin run on the library's fp32 GEMMs (--bert-gemm library) is a band, not an equality: the mean of the last 10 losses within
    a factor 10 of the twin's.  Red on the round-5 tree (NaN at the 2nd recorded step), green with the underflow-safe row norms."""
    got, log = _bench_regime(["--steps", "36", "--tag", "regime"])
    assert got["first_bad"] is None, log[-3000:]
    assert got["replays"] == 35 and got["steps_run"] == 43, got          # (1 + 6 eager, 1 eager call of the recorder, recording + 35 replays)
    twin, tlog = _bench_regime(["--steps", "36", "--tag", "twin", "--bert-gemm", "library"])
    assert twin["first_bad"] is None, tlog[-3000:]
    a, b = sum(got["losses"][-10:]) / 10, sum(twin["losses"][-10:]) / 10
    assert b / 10 <= a <= b * 10, (a, b)


def test_bench_refuses_to_report_a_dead_model():
    """bench.py must not time a dead model (round 5's headline was measured on one that had gone to NaN, printed inside `config`,
    exit code 0): with a NaN written into a parameter behind the warm-up (`--poison`, a test hook) the run ends with exit code 3,
    its one JSON line is strict JSON with `value: null`, an `error` and the non-finite losses as strings at the top level; the same
    command without the hook reports a finite value with `losses` at the top level."""
    import json
    from conftest import ROOT, run_isolated
    common = [os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--kernel-steps", "0", "--no-cpu-baseline"]
    r = run_isolated(common + ["--poison"], timeout=300)
    assert r.returncode == 3, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1], parse_constant=lambda c: pytest.fail(f"bare {c} in the line"))
    assert line["value"] is None and "non-finite" in line["error"] and any(isinstance(v, str) for v in line["losses"].values()), line
    r = run_isolated(common, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1], parse_constant=lambda c: pytest.fail(f"bare {c} in the line"))
    assert line["value"] > 0 and all(isinstance(v, float) and v == v for v in line["losses"].values()), line


def test_bench_regime_under_the_diagnostic_library_reports_no_split_overflow():
    """The same run against libhopmi_dbg.so (`make dbg`: -DHOPMI_CHECK_SPLIT, csrc/common.h): every conversion of a scaled value to
    its fp16 hi part, in every kernel, reports into a status buffer when it produces infinity / NaN.  No site may report in the
    bench regime -- neither an overflow on a finite input (an operand scale that does not hold for the data: what round 5's
    a-priori FFN bound did) nor a non-finite value entering a split.  The diagnostic build computes the same bits (asserted:
    identical losses to the production library's run is checked by the hunt's logs, here only that it ran the diagnostic code)."""
    from conftest import ROOT
    lib = os.path.join(ROOT, "hop-heterogeneous-topology-based-multimodal-entanglement-for-co-speech-gesture-generation_amd", "libhopmi_dbg.so")
    assert os.path.exists(lib), f"{lib} is missing: `make -C .../csrc all` builds it next to libhopmi.so"
    got, log = _bench_regime(["--steps", "36", "--tag", "regime_dbg"], env={"HOPMI_LIB": lib})
    assert got["diagnostic_library"], "the child did not load the diagnostic library"
    assert got["first_bad"] is None, log[-3000:]
    assert got["split_reports"] == [], got["split_reports"]


@pytest.mark.parametrize("eager_calls", [1, 0])
def test_graphed_step_epoch_10_to_11_transition(eager_calls, monkeypatch):
    """A training run crosses from epoch 10 to epoch 11 with the generator's recording already in use, and epoch 11 is the
    first time the discriminator's optimizer steps (train_llm.py:15-36).  Every phase gets its own eager call before it is
    recorded (also when the caller asks for none: a model's first call builds state that cannot be recorded), so the
    discriminator's Adam state is created by an eager step; and `_make_capturable` creates missing state before any
    recording (a state created under capture would be re-zeroed by every replay: step counters stuck at 1).  Checked: the
    step counters count the GAN-phase steps, losses, moments and parameters follow the eager path."""
    import hopmi
    from oracle.golden_util import Accel, step_args
    dev = _dev()
    _deterministic_draws(monkeypatch)
    m1, d1, inp = _pair(9, dev)
    m2, d2 = copy.deepcopy(m1), copy.deepcopy(d1)
    m2._randn_like = m1._randn_like
    mk = lambda m, d: (torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999)),
                       torch.optim.Adam(d.parameters(), lr=1e-3, betas=(0.5, 0.999)))
    args = step_args(9)
    batch = (inp["in_audio"], inp["log_melspec"], inp["text"], inp["target_dir_vec"], inp["vid_indices"])
    g1, o1 = mk(m1, d1)
    g2, o2 = mk(m2, d2)
    graphed = hopmi.GraphedTrainStep(args, m2, d2, g2, o2, eager_calls=eager_calls)
    epochs = [10, 10, 10, 11, 11, 11, 11]
    for it, epoch in enumerate(epochs):
        want = hopmi.train_llm(args, epoch, *batch, m1, d1, g1, o1, Accel())
        got = graphed(epoch, *batch)
        assert sorted(got) == sorted(want), (it, got, want)
        for k in want:
            # (Adam noise as in test_graphed_step_equals_eager; the discriminator trains at lr 1e-3 here, 10 x the usual.
            # DIV_REG is a ratio of two small differences between the step's two forwards: it amplifies that noise ~50 x)
            assert abs(got[k] - want[k]) <= (1e-3 if k != "DIV_REG" else 2e-2) * max(abs(want[k]), 1e-6), (it, k, got[k], want[k])
    assert graphed.n_eager == 2 and len(graphed.records) == 2
    n_gan = sum(e > 10 for e in epochs)
    for (n, a), (_, b) in zip(d1.named_parameters(), d2.named_parameters()):
        assert float(o2.state[b]["step"]) == n_gan == float(o1.state[a]["step"]), n
        ea, eb = o1.state[a]["exp_avg"], o2.state[b]["exp_avg"]
        if not (n.startswith("pre_conv.") and n.endswith(".bias")):      # (analytically zero gradients: rounding noise)
            # (a state re-created by every replay would hold (1 - beta1) g of the last step only: ~half of the accumulated moment;
            # the Adam noise between the two runs is per cent at most)
            assert (ea - eb).abs().max().item() <= 5e-2 * ea.abs().max().item() + 1e-9, n
        diff = (a - b).abs()
        assert diff.max().item() <= 4.5e-3 and diff.mean().item() <= 1e-3, (n, diff.max().item(), diff.mean().item())


@pytest.mark.parametrize("epoch,dtype", [(0, "fp32"), (11, "fp32"), (11, "bf16")])
def test_recorded_step_has_no_memset_nodes(epoch, dtype, tmp_path, monkeypatch):
    """Structural guard for the replay faults of DESIGN.md 5: in a replay, a memset NODE is not reliably ordered in front of
    the kernel node that depends on it once other work is queued on the stream between replays (seen twice: the persistent
    GRU's counters, and the semaphore memset of a torch multi-block reduction whose result then came out wrong).  The
    recording therefore holds no memset node at all: hopmi's own kernels initialise their state with kernels, and every
    torch reduction that would split over blocks (column sums of tall matrices, var_mean over a batch) goes through
    hopmi_colsum / hopmi's own kernels.  Checked at the sizes where torch's reductions do split: configs[1], B = 128,
    6-layer BERT-base, both phases, fp32 and bf16."""
    import hopmi
    from oracle.golden_util import step_args
    from test_gpu_parity import _full_size_setup
    dev = _dev()
    V, B = 9, 128
    m, d, bcfg, inp = _full_size_setup(V, B)
    m.to(dev).train(); d.to(dev).train()
    g_opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999), fused=True)
    d_opt = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999), fused=True)
    gin = {k: v.to(dev) for k, v in inp.items()}
    batch = (gin["in_audio"], gin["log_melspec"], gin["text"], gin["target_dir_vec"], gin["vid_indices"])
    prev = hopmi.mixed_precision(None if dtype == "fp32" else dtype)
    try:
        graphed = hopmi.GraphedTrainStep(step_args(V), m, d, g_opt, d_opt, eager_calls=1, debug=True)
        for _ in range(3):
            graphed(epoch, *batch)
        torch.cuda.synchronize()
    finally:
        hopmi.mixed_precision(prev)
    census = graphed.node_census()
    total = {}
    for c in census:
        for k, v in c.items():
            total[k] = total.get(k, 0) + v
    assert total.get("kernel", 0) > 400, total       # the census really lists the step's launches
    if total.get("memset", 0):
        # say which launches the memset nodes sit in front of (DOT dump of the segments that hold one)
        import re
        paths = graphed.dump_graphs(str(tmp_path))
        where = []
        for path, c in zip(paths, census):
            if c.get("memset", 0) and os.path.exists(path):
                text = open(path).read()
                where += re.findall(r'label\s*=\s*"([^"]*(?:memset|MEMSET)[^"]*)"', text)[:8]
        raise AssertionError(f"memset nodes in the recording: {total}; {where[:16]}")
    assert set(total) <= {"kernel", "memcpy", "empty", "event_record", "wait_event"}, total


@pytest.mark.parametrize("epoch", [0, 11])
def test_graphed_step_soak_interleaved_with_eager_work(epoch, monkeypatch):
    """50 replays of the recorded step at configs[1] size (B = 128, 6-layer BERT-base: the size at which torch's reductions
    split over blocks), with an eager evaluation forward of the SAME model and an eager training step of ANOTHER model copy
    queued between replays and no synchronisation in between -- what a training loop does around validation, short last
    batches and unshard().  Learning rate 0 and constant draws keep the weights fixed, so that every gradient tensor of every
    replay can be compared with an all-eager twin's on the same batch: bit for bit (same deterministic kernels, same
    inputs), except the mapping layer's weight gradient, which the recording forms with its own GEMM call (1e-6)."""
    import hopmi
    from oracle.golden_util import Accel, step_args
    from test_gpu_parity import _full_size_setup
    dev = _dev()
    _deterministic_draws(monkeypatch)
    V, B = 9, 128
    m, d, bcfg, inp = _full_size_setup(V, B)
    const = lambda t: torch.full_like(t, 0.25)
    m._randn_like = const
    m.to(dev).train(); d.to(dev).train()
    twin_m, twin_d = copy.deepcopy(m), copy.deepcopy(d)
    other_m, other_d = copy.deepcopy(m), copy.deepcopy(d)
    for mm in (twin_m, other_m):
        mm._randn_like = const
    mk = lambda mm, dd, lr: (torch.optim.Adam([p for p in mm.parameters() if p.requires_grad], lr=lr, betas=(0.5, 0.999)),
                             torch.optim.Adam(dd.parameters(), lr=lr, betas=(0.5, 0.999)))
    g_opt, d_opt = mk(m, d, 0.0)
    tg, td = mk(twin_m, twin_d, 0.0)
    og, od = mk(other_m, other_d, 1e-3)                       # the bystander really trains
    args = step_args(V)
    names = ("in_audio", "log_melspec", "text", "target_dir_vec", "vid_indices")
    gen = torch.Generator().manual_seed(11)
    batches = []
    for k in range(3):
        batches.append(tuple((inp[n] + 0.05 * k * torch.randn(inp[n].shape, generator=gen)).to(dev) if inp[n].is_floating_point()
                             else inp[n].roll(k, 0).to(dev) for n in names))
    graphed = hopmi.GraphedTrainStep(args, m, d, g_opt, d_opt, eager_calls=1)
    pairs = [(n, a, b) for (n, a), (_, b) in zip(list(m.named_parameters()) + list(d.named_parameters()),
                                                 list(twin_m.named_parameters()) + list(twin_d.named_parameters())) if a.requires_grad]
    bad = []
    n_iter = 52
    for it in range(n_iter):
        b = batches[it % 3]
        got = graphed(epoch, *b)
        # -- queued behind the replay, no synchronisation: a training step of another model copy, an evaluation of this one
        hopmi.train_llm(args, epoch, *batches[(it + 1) % 3], other_m, other_d, og, od, Accel())
        m.eval()
        with torch.no_grad():
            m(b[0], b[1], b[2], b[3][:, 0:16], b[4])
        m.train()
        want = hopmi.train_llm(args, epoch, *b, twin_m, twin_d, tg, td, Accel())
        torch.cuda.synchronize()
        for k in want:
            assert got[k] == want[k] or abs(got[k] - want[k]) <= 1e-6 * abs(want[k]), (it, k, got[k], want[k])
        for n, a, bb in pairs:
            if a.grad is None and bb.grad is None:
                continue
            assert a.grad is not None and bb.grad is not None, (it, n)
            if n.startswith("mapping_layer."):
                ok = (a.grad - bb.grad).abs().max().item() <= 1e-6 * bb.grad.abs().max().item() + 1e-12
            else:
                ok = torch.equal(a.grad, bb.grad)
            if not ok:
                bad.append((it, n, (a.grad - bb.grad).abs().max().item(), bb.grad.abs().max().item()))
        assert not bad, bad[:10]
    assert graphed.n_replay >= 50


# ------------------------------------------------------------------------------------------- input stage (feeder)
@pytest.mark.parametrize("which", ["recorded_step_baseline_size", "soak", "bf16"])
def test_operand_caches_hold_under_the_paranoid_mode(which, monkeypatch):
    """HOPMI_CACHE_CHECK (ops.cache_check): every hit of a derived-operand cache outside a stream capture -- row scales / operand
    images / row norms attached to tensors, parameter casts, fp16 weight images, the frozen BERT encoder's fused QKV weight, images
    and FFN bounds -- is verified against a fresh computation and raises on a mismatch.  Run once over the recorded step at
    configs[1] size (the image paths need >= 3072 rows), the 52-replay soak with its interleaved eager work, and the bf16 step: the
    tests' own assertions hold, nothing raises, and the mode did verify hits."""
    from hopmi import ops
    monkeypatch.setattr(ops, "CACHE_CHECK", True)
    n0 = ops.cache_checks_done()
    if which == "recorded_step_baseline_size":
        test_graphed_step_baseline_size_vs_reference(9, 128, 0, monkeypatch)
    elif which == "soak":
        test_graphed_step_soak_interleaved_with_eager_work(0, monkeypatch)
    else:
        test_graphed_step_bf16_follows_eager(monkeypatch)
    assert ops.cache_checks_done() >= n0 + 10, ops.cache_checks_done() - n0


def test_logmel_vs_float64_restatement():
    """hopmi_logmel (GPU: reflect-padded framing, periodic Hann, radix-2 FFT in LDS, Slaney mel filters, power_to_db with
    ref = max and an 80 dB floor) against oracle/mel_ref.py, the float64 restatement of librosa 0.8.1's published
    definitions of the reference's call (lmdb_data_loader.py:216-218).  librosa itself cannot be imported in the build container, so no
    fixture from it exists: the restatement is pinned stage by stage against torch.stft, scipy and transformers.audio_utils
    (tests/test_mel_pin.py).  Tolerance: 2e-2 dB absolute on the [-80, 0] dB
    range (fp32 FFT and log10), i.e. 2.5e-4 of the feature's range."""
    import numpy as np
    import hopmi
    from oracle import mel_ref
    dev = _dev()
    rng = np.random.default_rng(3)
    n = 36267
    t = np.arange(n) / 16000.0
    clips = np.stack([rng.standard_normal(n) * 0.1,                                   # noise
                      np.sin(2 * np.pi * 440.0 * t) + 0.3 * np.sin(2 * np.pi * 3000.0 * t),   # tones: most bands at the floor
                      np.concatenate([np.zeros(n // 2), rng.standard_normal(n - n // 2)]),       # half silence
                      np.zeros(n)]).astype(np.float32)                                 # silence: every value at amin
    got = hopmi.log_melspec(torch.from_numpy(clips).to(dev)).cpu().double().numpy()
    assert got.shape == (4, 34, 128)
    for i in range(clips.shape[0]):
        want = mel_ref.log_melspec(clips[i].astype(np.float64))
        err = np.abs(got[i] - want).max()
        assert err <= 2e-2, (i, err)
    assert np.abs(got[3]).max() <= 1e-5 and got[1].min() == -80.0 and abs(got[0].max()) <= 1e-5


def test_host_feeder_feeds_the_step(monkeypatch):
    """HostFeeder: host batches in the reference collate's dtypes (text as float64) arrive as device tensors equal to a
    plain copy, log_melspec equals the kernel on that audio, batches come in order, and train_llm runs on them."""
    import numpy as np
    import hopmi
    from oracle.golden_util import Accel, step_args
    dev = _dev()
    _deterministic_draws(monkeypatch)
    m, d, inp = _pair(9, dev)
    rng = np.random.default_rng(0)
    host = []
    for k in range(5):
        host.append(dict(audio_padded=rng.standard_normal((2, 36267)).astype(np.float32) * 0.1,
                         text_token_padded=rng.integers(0, 90, (2, 34)).astype(np.float64),      # np.zeros(34) rows -> float64
                         vec_seq=(rng.standard_normal((2, 34, 27)) * 0.1).astype(np.float32),
                         vid_indices=torch.from_numpy(rng.integers(0, 11, (2,)))))
    feeder = hopmi.HostFeeder(iter(host), dev)
    g_opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999))
    d_opt = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999))
    seen = 0
    for k, batch in enumerate(feeder):
        assert batch["text"].dtype == torch.float64 and batch["in_audio"].dtype == torch.float32
        assert batch["vid_indices"].dtype == torch.int64 and batch["log_melspec"].shape == (2, 34, 128)
        # (the device buffers are reused by the next batch: compare before asking for it)
        assert torch.equal(batch["in_audio"].cpu(), torch.from_numpy(host[k]["audio_padded"]))
        assert torch.equal(batch["text"].cpu(), torch.from_numpy(host[k]["text_token_padded"]))
        assert torch.equal(batch["target_dir_vec"].cpu(), torch.from_numpy(host[k]["vec_seq"]))
        assert torch.equal(batch["log_melspec"], hopmi.log_melspec(batch["in_audio"]))
        ret = hopmi.train_llm(step_args(9), 0, batch["in_audio"], batch["log_melspec"], batch["text"], batch["target_dir_vec"],
                              batch["vid_indices"], m, d, g_opt, d_opt, Accel())
        feeder.refill()
        assert ret["loss"] == ret["loss"]
        seen += 1
    assert seen == 5
