"""The training step of the reference (`train_eval/train_llm.py:9-98`) as replayed hipGraphs.

`steps.train_llm` issues ~1 500 kernel launches per step; on a slow or instrumented host the step is bound by the time
the host needs to issue them, not by the device.  `GraphedTrainStep` records the launches of one `train_llm` call once
(the very same Python code: `steps.train_llm` runs under stream capture with this object standing in for the
`accelerator`) and replays them:

    step = hopmi.GraphedTrainStep(args, model, discriminator, model_optim, dis_optimizer)      # Adam optimizers
    losses = step(epoch, in_audio, log_melspec, text_token_padded, target_dir_vec, vid_indices)   # = train_llm(...)

* The first `eager_calls` calls OF EACH PHASE (`epoch <= 10` / GAN phase) run `steps.train_llm` itself (library handles,
  workspaces, lazily built operand images and autograd threads come up outside a capture -- the discriminator's backward
  runs for the first time at epoch 11); the next call captures and replays; later calls copy the batch into the graph's
  static inputs and replay.  Every call is exactly one training step.  One recording per phase (`epoch <= 10` / GAN phase) and batch
  shape; a batch of another shape (the last, short one of an epoch) runs the eager step.
* The recording is cut into segments where the step talks to the outside: behind the loss copy (the replay waits for
  that event only and returns while backward + optimizer still run), and -- with more than one rank -- around every
  collective, which are ordinary eager RCCL calls between two graph launches (nothing of RCCL is captured).
* Dropout: the seeded kernels add a device-side stream position to their (frozen) seed argument
  (`ops.SEED_DEV`, bumped at the head of every replay); torch's own generators are graph-safe.
* Data parallelism: the batch-independent prototype matrix S = W_map E + b
  (HOP.py:116,200; 70 GFLOP forward + 70 GFLOP weight gradient, constant in the batch) is ROW-SHARDED over the ranks:
  each rank computes its ceil(1500 / N) rows, S is all-gathered (4.6 MB), the backward all-reduces dS (4.6 MB) instead
  of the 183 MB weight gradient, and each rank forms dW for and updates only its own rows.  The other gradients
  (~80 MB fp32) are exchanged as one flat all-reduce per module.  `unshard()` brings every rank's copy of the mapping
  layer (and its Adam moments) up to date, e.g. before `state_dict()`.
"""
import contextlib
import gc

import torch
import torch.distributed as dist

from . import ops as _ops
from . import steps as _steps

_STEP_INC = 0x9E3779B1            # added to ops.SEED_DEV at the head of every replay
EXIT_CAPTURE_FAILED = 86          # exit code when no device work is possible after a failed recording (see _recording_failed)
_GRAVEYARD = []                   # (graph, stream) of abandoned captures: kept for the life of the process


def shard_rows(n_rows: int, rank: int, world: int):
    """Row range [r0, r1) of a matrix of n_rows rows owned by `rank`, and the padded per-rank row count (equal shards
    for the all-gather; the last ranks' shards are short or empty)."""
    per = -(-n_rows // world)
    r0 = min(rank * per, n_rows)
    return r0, min(r0 + per, n_rows), per


def mapping_grad_rows(dS, E, r0, r1, out_w, out_b, bf16=False):
    """Rows [r0, r1) of the mapping layer's gradients from dS (1500 x d_llm): dW = dS E^T, db = sum_j dS (the backward
    of S = W E + b[:, None]); written into the same rows of the full-size gradient buffers."""
    if r1 <= r0:
        return
    g = dS[r0:r1]
    if bf16:
        out_w[r0:r1].copy_(g.to(torch.bfloat16) @ E.to(torch.bfloat16).t())
    elif _ops.f16_mm_nt_ok(g, E, (E,)):
        _ops.f16_mm_nt(g.contiguous(), E, (E,), out=out_w[r0:r1])       # (E frozen: its image is made once, in an eager call)
    else:
        torch.mm(g, E.t(), out=out_w[r0:r1])
    torch.sum(g, dim=1, out=out_b[r0:r1])


def all_gather_rows(buf, per, group=None):
    """In-place all-gather of equal row shards: rank r contributes buf[r*per:(r+1)*per]."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = buf[rank * per:(rank + 1) * per]
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(buf, mine, group=group)
    else:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine.contiguous(), group=group)
        for r, p in enumerate(parts):
            buf[r * per:(r + 1) * per].copy_(p)


def all_reduce_mean(flat, group=None):
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
    else:
        dist.all_reduce(flat, group=group)
        flat.div_(dist.get_world_size(group))


def all_reduce_mean_start(flat, group=None):
    """The same all-reduce, started and left running (RCCL: on its own stream, behind what the current stream holds now);
    returns what all_reduce_mean_finish needs."""
    if dist.get_backend(group) == "nccl":
        return dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group, async_op=True), None
    return dist.all_reduce(flat, group=group, async_op=True), dist.get_world_size(group)


def all_reduce_mean_finish(flat, pending):
    work, div = pending
    work.wait()                                   # (RCCL: the current stream waits; gloo: the host does)
    if div is not None:
        flat.div_(div)


def inplace_group_ok(group=None) -> bool:
    """Can a LIST of tensors be all-reduced in place under one RCCL group call (ncclGroupStart / End: one launch, no flat copy)?"""
    import os
    return (os.environ.get("HOPMI_EXCHANGE_INPLACE", "1") != "0" and hasattr(dist, "_coalescing_manager")
            and dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl")


def all_reduce_mean_group(tensors, group=None, async_op=False):
    """All-reduce (mean) every tensor of the list IN PLACE under one RCCL group call -- what the flat form does with two copies of
    all the gradients around one all-reduce (pack / unpack: 2 x 80 MB read + written per step at fp32).  Returns the manager
    (`async_op`: `.wait()` it: the current stream then waits for the collective)."""
    with dist._coalescing_manager(group=group, async_ops=async_op) as cm:
        for t in tensors:
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
    return cm


def _params_behind(tensors):
    """ids of the leaf tensors (parameters) that the autograd graphs of `tensors` reach."""
    seen, out = set(), set()
    stack = [t.grad_fn for t in tensors if t is not None and t.grad_fn is not None]
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        var = getattr(fn, "variable", None)
        if var is not None:
            out.add(id(var))
        stack.extend(f for f, _ in fn.next_functions)
    return out


class _Capture:
    """Stands in for the `accelerator` of `train_llm` while its launches are being recorded, and owns the recording:
    a list of ("graph", CUDAGraph) / ("eager", callable) entries that a replay walks in order."""
    takes_only = True

    def __init__(self, owner, gan):
        self.owner, self.gan = owner, gan
        self.plan = []
        self.status = []            # ops.STATUS_SINK: status words of the persistent GRU launches recorded so far
        self.graph = None
        self.fetch = None
        self.keep = []              # tensors the recording owns (flat exchange buffers, ...)
        self.cuts = []              # (tensor, detached leaf) pairs of the graded forward (ops.cut_point), overlapped exchange only
        self.early = None           # the first half of an overlapped exchange, between _exchange_early and _after_backward

    def begin(self):
        # debug: the hipGraph_t is kept behind the executable graph (node_census / dump_graphs)
        self.graph = torch.cuda.CUDAGraph(keep_graph=True) if self.owner.debug else torch.cuda.CUDAGraph()
        self.graph.capture_begin(pool=self.owner._pool)

    def end(self):
        self.graph.capture_end()
        if self.owner.debug:
            self.graph.instantiate()
        self.plan.append(("graph", self.graph))
        self.graph = None

    def cut(self, eager_fn=None):
        self.end()
        if eager_fn is not None:
            self.plan.append(("eager", eager_fn))
        self.begin()

    # -- hooks called from steps.train_llm / steps._LossFetch ------------------------------------------------------
    def take_status(self):
        parts = [w.float().reshape(()) for w in self.status]
        self.status.clear()
        parts.append(self.owner._bwd_status)          # the previous replay's backward launches
        parts.append(self.owner._peer_status)         # ... and what rode on the gradient exchanges since the last fetch
        total = torch.stack(parts).sum()
        self.owner._peer_status.zero_()
        return total

    def fetch_into(self, fetch, dev):
        o = self.owner
        host = o._host[:dev.numel()]
        host.copy_(dev, non_blocking=True)
        self.fetch = fetch
        ev = o._event
        self.cut(ev.record)                            # the replay's host side waits for this event only
        return host

    def cut_leaf(self, t):
        leaf = t.detach().requires_grad_()
        self.cuts.append((t, leaf))
        return leaf

    def backward(self, loss, only=None):
        is_disc = only is not None and len(only) == 1 and only[0] is self.owner.disc
        cuts, self.cuts = self.cuts, []
        loss.backward()
        if cuts and not is_disc:
            # The backward above ended at the leaves the graded forward was cut at (decoder input, VAE outputs): the gradients of
            # everything behind them are complete.  Their all-reduce is started and left running; the backward of everything in
            # front of the leaves is recorded into the next segment and replays while that exchange is on the wire.
            self.owner._exchange_early(self, [t for t, _ in cuts])
            live = [(t, leaf.grad) for t, leaf in cuts if leaf.grad is not None]
            torch.autograd.backward([t for t, _ in live], [g for _, g in live])
        self.owner._after_backward(self, is_disc)


class _AdamMultiStep:
    """While a step is being recorded, `opt.step()` of a plain torch.optim.Adam runs as ONE hopmi_adam_multi launch per parameter
    group (csrc/adam.hip: the arithmetic of torch's fused Adam at the rate of a copy -- torch's multi-tensor launch serves the
    generator's 172 small tensors badly, tools/probes/adam_floor.py) instead of torch's multi_tensor_apply launches.  The
    optimizer's own state tensors are what is updated (exp_avg, exp_avg_sq, the device-side step counters: state_dict() and later
    eager steps see nothing unusual) and its step hooks run as they would (the row-sharded mapping layer's among them).  Eager
    steps keep torch's own step.  `plans`: {group index: ops.AdamPlan}, made before the recording (GraphedTrainStep._build)."""

    def __init__(self, opt, plans):
        self.opt, self.plans = opt, plans
        self.active = bool(plans)

    def __enter__(self):
        if self.active:
            self.opt.step = self._step                   # (instance attribute: shadows the class's hooked step for the recording)
        return self

    def __exit__(self, *exc):
        if self.active:
            self.opt.__dict__.pop("step", None)
        return False

    @torch.no_grad()
    def _step(self, closure=None):
        opt = self.opt
        for hook in list(getattr(opt, "_optimizer_step_pre_hooks", {}).values()):
            hook(opt, (), {})
        for gi, group in enumerate(opt.param_groups):
            plan = self.plans.get(gi)
            if plan is None:
                continue
            live = [p for p in plan.params if p.grad is not None]
            if not live:
                continue
            steps = [opt.state[p]["step"] for p in live]
            torch._foreach_add_(steps, 1)
            st = [opt.state[p] for p in plan.params]
            plan.bind([p.grad for p in plan.params], [s_["exp_avg"] for s_ in st], [s_["exp_avg_sq"] for s_ in st])
            plan.step(group["lr"], group["betas"][0], group["betas"][1], group["eps"], steps[0])
        for hook in list(getattr(opt, "_optimizer_step_post_hooks", {}).values()):
            hook(opt, (), {})


class GraphedTrainStep:
    def __init__(self, args, model, discriminator, model_optim, dis_optimizer, accelerator=None, group=None,
                 eager_calls=2, grad_dtype=None, enabled=True, force_exchange=False, debug=False, overlap=True, inplace=False):
        """`accelerator`: what the eager calls hand to train_llm (a GradSync when more than one rank trains; default a
        plain backward).  `grad_dtype=torch.bfloat16` halves the bytes of the flat gradient exchanges (the sums still
        land in fp32 gradients).  `enabled=False` makes every call the eager step (for A/B runs).  `force_exchange` runs
        the collectives of the N > 1 recording on a 1-rank group too (rehearsal on a single-GPU box).  `debug=True` keeps
        the recorded hipGraphs inspectable (`dump_graphs`).  `overlap` (N > 1 only): the generator's backward is recorded in two
        halves, cut at the decoder's input, and the all-reduce of the first half's gradients (decoder GRU, head) runs under the
        second half (BERT, reprogramming attention, WaveNet, mapping layer); False: one all-reduce behind the whole backward."""
        self.args, self.model, self.disc = args, _steps._unwrap(model), _steps._unwrap(discriminator)
        self.g_opt, self.d_opt = model_optim, dis_optimizer
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.exchange = self.world > 1 or (force_exchange and dist.is_available() and dist.is_initialized())
        self.accel = accelerator if accelerator is not None else _PlainBackward()
        # at least one eager call per phase: besides library handles, every model builds state lazily in its first call
        # (the frozen BERT's operand images, the GRU's packed weights, selection matrices), and some of that cannot be recorded
        self.eager_calls = max(1, int(eager_calls))
        self.eager_left = {}                        # phase (gan flag) -> eager calls still to make before recording it
        self.grad_dtype = grad_dtype
        self.overlap = bool(overlap)
        # `inplace=True` (fp32 exchange over RCCL only): the gradient tensors are all-reduced IN PLACE under one RCCL group call instead
        # of pack -> one all-reduce -> unpack (2 x 80 MB of copies).  Measured on the 1-rank rehearsal (round 4, one box,
        # profiles/r04_rehearse_sync.txt): NOT faster -- 16.19 vs 16.11 ms (TED), 22.60 vs 22.41 (GAN), 13.19 vs 13.11 (V = 42) against 15.71 /
        # 21.93 / 12.70 without any exchange: a group call over ~150 tensors costs more than the two flat copies it saves, and what
        # the exchange adds to the step is mostly its cuts and launches, not the copies.  Kept as an option for a measurement on
        # real links; the flat form stays the default.
        self.inplace = bool(inplace)
        self.exchange_plan = {}                      # what each recorded all-reduce carries (filled while recording; bench.py reports it)
        self.enabled = enabled
        self.debug = debug
        self.records = {}
        self.broken = None                          # why recording was given up (an exception while recording), else None
        self.n_eager = self.n_replay = 0            # calls served by steps.train_llm itself / by a replay
        self.sharded = False
        self._pool = None
        self._built = False

    # -- one-time device state -------------------------------------------------------------------------------------
    def _build(self, dev):
        m = self.model
        self._pool = torch.cuda.graph_pool_handle()
        self._stream = torch.cuda.Stream(device=dev)
        self._event = torch.cuda.Event()
        self._host = torch.zeros(16, dtype=torch.float32).pin_memory()
        self._bwd_status = torch.zeros((), dtype=torch.float32, device=dev)
        self._peer_status = torch.zeros((), dtype=torch.float32, device=dev)   # status words of ANY rank (N > 1), see _after_backward
        self._seed_word = torch.zeros(1, dtype=torch.int64, device=dev)    # ops.SEED_DEV while recording
        self._has_proto = bool(getattr(m, "use_reprograme", False))
        if self._has_proto:
            n, d = m.mapping_layer.weight.shape[0], m.word_embeddings.shape[1]
            self.r0, self.r1, self.per = shard_rows(n, self.rank, self.world)
            self._S_pad = torch.zeros(self.per * self.world, d, dtype=torch.float32, device=dev)
            self._S = self._S_pad[:n].requires_grad_()                    # the leaf the graded forward reads
            self._Wg = torch.zeros_like(m.mapping_layer.weight)            # rows outside [r0, r1) stay zero
            self._bg = torch.zeros_like(m.mapping_layer.bias)
        for opt in (self.g_opt, self.d_opt):
            _make_capturable(opt)
        # the generator's optimizer step of a recording as one launch per parameter group (see _AdamMultiStep)
        self._adam_plans = {}
        if _ops.adam_multi_supported(self.g_opt):
            for gi, group in enumerate(self.g_opt.param_groups):
                ps = [p for p in group["params"] if p.requires_grad]
                if ps:
                    st = [self.g_opt.state[p] for p in ps]
                    self._adam_plans[gi] = _ops.AdamPlan(ps, [s_["exp_avg"] for s_ in st], [s_["exp_avg_sq"] for s_ in st])
        # more than one rank: every rank steps Adam on its own rows of the mapping layer only (see _after_backward)
        self._rows_adam = bool(self._has_proto and self.world > 1 and isinstance(self.g_opt, torch.optim.Adam)
                               and not isinstance(self.g_opt, torch.optim.AdamW))
        if self._rows_adam:
            self.g_opt.register_step_post_hook(self._adam_own_rows)
        # what a replay writes in place: the optimizers' parameters and the modules' buffers (BatchNorm statistics)
        seen, written = set(), []
        for t in [p for opt in (self.g_opt, self.d_opt) for g in opt.param_groups for p in g["params"]] + \
                 list(m.buffers()) + list(self.disc.buffers()):
            if id(t) not in seen:
                seen.add(id(t))
                written.append(t)
        self._written = tuple(written)
        # what a recording reads but no replay writes: the frozen parameters (the BERT, the word embeddings).  Tensors derived
        # from them (bf16 copies, part images of the split GEMMs, the fused QKV weight) are built once, in an eager call, and
        # a recording holds their addresses: if a frozen parameter is modified afterwards (load_state_dict, an in-place copy),
        # the recordings are dropped and the step runs eagerly again before it is re-recorded -- see __call__
        self._frozen = tuple(p for mod in (m, self.disc) for p in mod.parameters() if id(p) not in seen)
        self._built = True

    def _frozen_versions(self):
        # (+ ops.CACHE_EPOCH: ops.invalidate_weight_images() / reset_all_caches("all") drops the frozen weights' images, whose
        # addresses a recording holds -- such a recording is dropped and re-made like one whose frozen parameter changed)
        return tuple(p._version for p in self._frozen) + (_ops.CACHE_EPOCH,)

    def _mark_written(self):
        """A replay updates parameters and buffers on the device behind torch's back: advance their version counters as
        the in-place updates of the eager step do, so that everything keyed on them (the no-grad forward's K/V prototype
        cache, Model._kv_infer; autograd's saved-tensor checks) sees a modified tensor."""
        bump = getattr(torch._C._autograd, "_unsafe_set_version_counter", None)
        if bump is not None:
            bump(self._written, tuple(t._version + 1 for t in self._written))
        self.model._kv_infer = None

    # -- the three places where a recording differs from the eager step -------------------------------------------
    def _prototype_segment(self, cap, amp):
        """Head of every replay: advance the dropout stream, compute this rank's rows of S, all-gather."""
        self._seed_word.add_(_STEP_INC)
        if not self._has_proto:
            return
        with torch.no_grad(), amp:
            if self.r1 > self.r0:
                self._S_pad[self.r0:self.r1].copy_(self.model.prototype_rows(self.r0, self.r1))
        if self.exchange:
            buf, per, grp = self._S_pad, self.per, self.group
            cap.cut(lambda: all_gather_rows(buf, per, grp))

    def _pack(self, grads, extra=0):
        n = sum(g.numel() for g in grads)
        flat = torch.empty(n + extra, dtype=self.grad_dtype or torch.float32, device=grads[0].device)
        views, off = [], 0
        for g in grads:
            views.append(flat[off:off + g.numel()].view(g.shape))
            off += g.numel()
        torch._foreach_copy_(views, grads)
        return flat, views, n

    def _exchange_early(self, cap, cut_tensors):
        """First half of an overlapped exchange: the gradients that are complete when the backward has reached the cut (the
        parameters the graphs in front of the cut do NOT reach) go into a flat buffer whose all-reduce is started between two graph
        launches and finished in _after_backward, behind the rest of the backward."""
        late = _params_behind(cut_tensors)
        early = [p for p in self.model.parameters() if p.requires_grad and p.grad is not None and id(p) not in late]
        if not early:
            return
        grads = [p.grad for p in early]
        pending, grp = {}, self.group
        if self._inplace():
            # in place under one RCCL group call: the gradient tensors themselves are what the collective reads and writes
            cap.keep.append(grads)
            cap.cut(lambda: pending.__setitem__("w", all_reduce_mean_group(grads, grp, async_op=True)))
            cap.early = dict(ids={id(p) for p in early}, grads=grads, views=None, flat=None, pending=pending)
            return
        flat, views, _ = self._pack(grads)
        self.exchange_plan["generator_first_half"] = dict(tensors=len(grads), elements=flat.numel(), dtype=str(flat.dtype).replace("torch.", ""))
        cap.keep.append(flat)
        cap.cut(lambda: pending.__setitem__("w", all_reduce_mean_start(flat, grp)))
        cap.early = dict(ids={id(p) for p in early}, grads=grads, views=views, flat=flat, pending=pending)

    def _after_backward(self, cap, is_disc):
        """Called with the gradients of one module freshly produced: exchange them (world > 1), then -- for the
        generator -- turn dS into this rank's rows of the mapping layer's gradients."""
        module = self.disc if is_disc else self.model
        early, cap.early = (None, None) if is_disc else (cap.early, None)
        if self.exchange:
            grads = [p.grad for p in module.parameters() if p.requires_grad and p.grad is not None
                     and (early is None or id(p) not in early["ids"])]
            if not is_disc and self._has_proto and self._S.grad is not None:
                grads.append(self._S.grad)
            if self._inplace():
                # this rank's persistent-kernel status words ride as one more (1-element) tensor of the group call
                words = [w.float().reshape(()) for w in cap.status]
                word = torch.zeros(1, dtype=torch.float32, device=grads[0].device)
                if words:
                    word.copy_(torch.stack(words).sum().reshape(1))
                tensors = grads + [word]
                cap.keep.append(tensors)
                grp = self.group
                if early is None:
                    cap.cut(lambda: all_reduce_mean_group(tensors, grp))
                else:
                    e_pending = early["pending"]
                    cap.cut(lambda: (e_pending.pop("w").wait(), all_reduce_mean_group(tensors, grp)))
                self._peer_status.add_(word[0])
                self._mapping_rows(is_disc)
                return
            flat, views, n = self._pack(grads, extra=1)
            self.exchange_plan["discriminator" if is_disc else ("generator_second_half" if early is not None else "generator")] = dict(
                tensors=len(grads), elements=flat.numel(), dtype=str(flat.dtype).replace("torch.", ""))
            # one more element: this rank's persistent-GRU status words so far in the replay.  A hand-off time-out on one rank
            # then shows up in EVERY rank's next loss fetch, so all ranks raise at the same step instead of the healthy ones
            # waiting in the next collective for a rank that has stopped.
            words = [w.float().reshape(()) for w in cap.status]
            flat[n:].copy_((torch.stack(words).sum() if words else torch.zeros((), device=flat.device)).reshape(1))
            cap.keep.append(flat)
            grp = self.group
            if early is None:
                cap.cut(lambda: all_reduce_mean(flat, grp))
            else:
                e_flat, e_pending = early["flat"], early["pending"]
                cap.cut(lambda: (all_reduce_mean_finish(e_flat, e_pending.pop("w")), all_reduce_mean(flat, grp)))
                torch._foreach_copy_(early["grads"], early["views"])
            torch._foreach_copy_(grads, views)
            self._peer_status.add_(flat[n].float())
        self._mapping_rows(is_disc)

    def _inplace(self):
        return self.inplace and self.grad_dtype is None and inplace_group_ok(self.group)

    def _mapping_rows(self, is_disc):
        """Behind the generator's exchange: dS -> this rank's rows of the mapping layer's gradients."""
        if not is_disc and self._has_proto:
            m = self.model
            dS = self._S.grad
            with torch.no_grad():
                if dS is not None:
                    # (the step's precision, not the ambient autocast state: this runs behind train_llm's autocast block)
                    bf16 = (getattr(self.args, "mixed_precision", None) or _steps._MIXED) == "bf16"
                    mapping_grad_rows(dS, m.word_embeddings, self.r0, self.r1, self._Wg, self._bg, bf16=bf16)
            if self._rows_adam:
                # this rank's rows only: the optimizer's own step skips the two tensors (no gradient), the step hook below
                # runs Adam on the row views (and their moments) -- the other ranks' rows, whose gradient is identically zero
                # here, are not touched at all (183 MB of weights + two moments read and written per step otherwise)
                m.mapping_layer.weight.grad = m.mapping_layer.bias.grad = None
            else:
                m.mapping_layer.weight.grad, m.mapping_layer.bias.grad = self._Wg, self._bg

    def _adam_own_rows(self, opt, *_):
        """Optimizer step post-hook (recorded with the step): Adam on rows [r0, r1) of the mapping layer with the optimizer's own
        state tensors and hyper-parameters, through torch's functional form on row views."""
        if _steps._CAPTURE is None or not self._rows_adam or self.r1 <= self.r0:
            return
        from torch.optim.adam import adam as _adam
        m = self.model
        r0, r1 = self.r0, self.r1
        for p, g in ((m.mapping_layer.weight, self._Wg), (m.mapping_layer.bias, self._bg)):
            grp = next(gr for gr in opt.param_groups if any(q is p for q in gr["params"]))
            st = opt.state[p]
            with torch.no_grad():
                _adam([p[r0:r1]], [g[r0:r1]], [st["exp_avg"][r0:r1]], [st["exp_avg_sq"][r0:r1]],
                      [st["max_exp_avg_sq"][r0:r1]] if grp.get("amsgrad", False) else [], [st["step"]],
                      foreach=grp.get("foreach"), capturable=True, differentiable=False, fused=grp.get("fused"),
                      amsgrad=grp.get("amsgrad", False), beta1=grp["betas"][0], beta2=grp["betas"][1], lr=grp["lr"],
                      weight_decay=grp.get("weight_decay", 0.0), eps=grp["eps"], maximize=grp.get("maximize", False))

    # -- capture -----------------------------------------------------------------------------------------------------
    def _capture(self, epoch, batch):
        dev = batch[0].device
        gan = epoch > 10 and self.args.loss_gan_weight > 0.0
        static = [t.clone() for t in batch]
        cap = _Capture(self, gan)
        cur = torch.cuda.current_stream(dev)
        self._stream.wait_stream(cur)
        amp_factory = lambda: _steps._amp(self.args, static[3])
        m = self.model
        prev_sink, prev_cap, prev_seed, prev_cut = _ops.STATUS_SINK, _steps._CAPTURE, _ops.SEED_DEV, _ops.CUT_HOOK
        _ops.deferred_status()                                   # pending eager launches are not this recording's
        # nothing of the eager calls may be served to, or kept alive across, the recording: cached operand casts live in the eager
        # allocator pool (a recording handed one would bake its address in without recording the cast); gradient tensors of the
        # eager step would be accumulated into instead of being allocated from the graph's pool; the stack kernel's workspace of
        # the eager calls' stream gets a twin for the recording's stream now (allocated under capture it would be zero-filled by
        # a recorded launch on every replay)
        _ops.cast_cache_reset()
        _ops.stack_ws_prepare(cur.cuda_stream, self._stream.cuda_stream)
        for mod in (m, self.disc):
            for p in mod.parameters():
                p.grad = None
        gc.collect()
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(self._stream), _ops.no_timer():
            _ops.STATUS_SINK, _steps._CAPTURE, _ops.SEED_DEV = cap.status, cap, self._seed_word
            _ops.CUT_HOOK = cap.cut_leaf if (self.exchange and self.overlap) else None
            if self._has_proto:
                m._proto_S = self._S
                self._S.grad = None
            try:
                cap.begin()
                try:
                    self._prototype_segment(cap, amp_factory())
                    with _AdamMultiStep(self.g_opt, self._adam_plans):
                        _steps.train_llm(self.args, epoch, *static, m, self.disc, self.g_opt, self.d_opt, cap)
                    # tail of the last segment: the backward launches' status words, read by the NEXT replay's fetch
                    if cap.status:
                        self._bwd_status.copy_(torch.stack([w.float().reshape(()) for w in cap.status]).sum())
                        cap.status.clear()
                except BaseException as exc:
                    self._recording_failed(cap, exc)
                    raise
                else:
                    cap.end()
            finally:
                _ops.STATUS_SINK, _steps._CAPTURE, _ops.SEED_DEV, _ops.CUT_HOOK = prev_sink, prev_cap, prev_seed, prev_cut
                m._proto_S = None
                _ops.cast_cache_reset()                          # (copies made while recording live in the graph's pool)
        cur.wait_stream(self._stream)
        if cap.fetch is None:
            raise RuntimeError("hopmi GraphedTrainStep: the recorded step never fetched its losses")
        return dict(cap=cap, static=static, terms=cap.fetch.terms, n_vals=len(cap.fetch.terms) + 1)

    def _recording_failed(self, cap, exc):
        """An exception interrupted a recording: `self._stream` is still inside a stream capture.  Measured on this runtime
        (tools/probes/capture_failure_probe.py, profiles/r04_capture_probe.txt): a capture LEFT OPEN aborts the process at
        interpreter exit (SIGABRT out of the graph object's destructor); a healthy capture (a Python-level error, and most illegal
        calls, which this runtime refuses without invalidating the capture) ends the ordinary way; an INVALIDATED one (e.g. a
        device-wide synchronize under capture) makes hipStreamEndCapture answer with the invalidation error and the stream keeps
        REPORTING that state -- but the process goes on working (fresh allocations, GEMMs, syncs, a second capture on a new
        stream).  So: look first (hopmi_stream_capture_status); healthy -> capture_end(), graph dropped; otherwise the capture is
        ended by hopmi_stream_capture_abandon whatever it answers, the allocator is taken off the graph's pool by hand, torch's
        graph object -- which still believes it is capturing -- is parked for the life of the process, and the stream is replaced.
        Either way this object stops recording (`enabled = False`: every later call is the eager step, `broken` says why) and the
        caller gets the original exception.  Last, a round trip of ordinary device work proves the process can still train; if
        THAT fails it says so in one line and exits with EXIT_CAPTURE_FAILED (never a re-exec)."""
        import ctypes
        import os
        import sys
        import traceback
        from . import _lib
        print("hopmi GraphedTrainStep: exception while recording the step:", file=sys.stderr)
        traceback.print_exception(type(exc), exc, exc.__traceback__, file=sys.stderr)
        self.enabled = False
        self.records.clear()
        _ops.reset_all_caches("recording")          # (whatever the interrupted recording cached lives in a pool that is being given up)
        self.broken = f"{type(exc).__name__}: {str(exc).splitlines()[0] if str(exc) else ''}"
        L = _lib.lib()
        st = ctypes.c_int(-1)
        handle = self._stream.cuda_stream
        rc = L.hopmi_stream_capture_status(handle, ctypes.byref(st))
        how = {0: "not capturing", 1: "active", 2: "invalidated"}.get(st.value, f"unknown ({rc})")
        ended = rc == 0 and st.value == 0
        if rc == 0 and st.value == 1 and cap.graph is not None:
            try:
                cap.graph.capture_end()                      # healthy: the ordinary end; the graph is dropped with `cap`
                ended = True
            except BaseException as e2:                      # noqa: BLE001
                print(f"hopmi GraphedTrainStep: capture_end after the failure raised {type(e2).__name__}: {e2}", file=sys.stderr)
        if not ended:
            _GRAVEYARD.append((cap.graph, self._stream))     # (their destructors must not run against a capture they do not own any more)
            if L.hopmi_stream_capture_abandon(handle) != 0:
                print(f"hopmi GraphedTrainStep: {L.hopmi_last_error().decode()}", file=sys.stderr)
            dev = torch.cuda.current_device()
            for fn in (torch._C._cuda_endAllocateToPool, torch._C._cuda_releasePool):
                try:
                    fn(dev, self._pool)
                except Exception as e3:                       # noqa: BLE001  (the pool was never begun / already handed back)
                    print(f"hopmi GraphedTrainStep: {fn.__name__}: {e3}", file=sys.stderr)
            self._stream = torch.cuda.Stream(device=dev)
        cap.graph = None
        self._pool = torch.cuda.graph_pool_handle()
        try:                                                  # can this process still issue device work?  (on the caller's stream)
            with torch.cuda.stream(torch.cuda.default_stream()):
                probe = torch.empty(1 << 20, dtype=torch.float32, device="cuda").fill_(1.0)
                alive = float(probe.sum().item()) == float(1 << 20)
        except Exception as e4:                               # noqa: BLE001
            alive = False
            print(f"hopmi GraphedTrainStep: device work after the failed recording raised {type(e4).__name__}: {e4}", file=sys.stderr)
        if not alive:
            print(f"hopmi GraphedTrainStep: FATAL: no device work is possible after the failed recording (capture was {how}; "
                  f"{self.broken}) -- exiting with code {EXIT_CAPTURE_FAILED}", file=sys.stderr, flush=True)
            os._exit(EXIT_CAPTURE_FAILED)
        print(f"hopmi GraphedTrainStep: recording abandoned (capture was {how}); this object runs the eager step from now on: {self.broken}",
              file=sys.stderr, flush=True)

    # -- call --------------------------------------------------------------------------------------------------------
    def __call__(self, epoch, in_audio, log_melspec, text_token_padded, target_dir_vec, vid_indices):
        batch = (in_audio, log_melspec, text_token_padded, target_dir_vec, vid_indices)
        if not (self.enabled and in_audio.is_cuda):
            return self._eager(epoch, batch)
        gan = epoch > 10 and self.args.loss_gan_weight > 0.0
        left = self.eager_left.setdefault(gan, self.eager_calls)
        if left > 0:
            self.eager_left[gan] = left - 1
            return self._eager(epoch, batch)
        key = (gan, torch.is_autocast_enabled(), _steps._MIXED, getattr(self.args, "mixed_precision", None)) + tuple(
            (tuple(t.shape), t.dtype) for t in batch)
        rec = self.records.get(key)
        if rec is not None and rec["frozen"] != self._frozen_versions():
            # a frozen parameter was modified behind the recordings: they replay tensors derived from the old values
            self.records.clear()
            self.eager_left.clear()
            self._pool = torch.cuda.graph_pool_handle()          # (the old pool went with its last graph)
            return self(epoch, *batch)
        if rec is None:
            if any(k[0] == gan for k in self.records):           # another batch shape of a recorded phase: eager
                return self._eager(epoch, batch)
            if not self._built:
                self._build(in_audio.device)
            rec = self.records[key] = self._capture(epoch, batch)
            rec["frozen"] = self._frozen_versions()
        for dst, src in zip(rec["static"], batch):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.n_replay += 1
        for kind, x in rec["cap"].plan:
            if kind == "graph":
                x.replay()
            else:
                x()
        self._mark_written()
        # every replay of an N > 1 recording updates only this rank's rows of the mapping layer (and their Adam moments): the
        # copies are sharded again after ANY replay, also one that follows an unshard() (no new capture happens then)
        self.sharded = self.world > 1 and self._has_proto
        self._event.synchronize()
        return _steps._LossFetch.decode(rec["terms"], self._host[:rec["n_vals"]].tolist(), True)

    def node_census(self):
        """Node types of every recorded graph segment (needs `debug=True`): a list, one dict {type name: count} per segment,
        read from the hipGraph_t objects with hipGraphGetNodes / hipGraphNodeGetType.  tests/test_gpu_graph.py uses it to
        check that a recording holds no memset node."""
        import ctypes
        if not self.debug:
            raise RuntimeError("hopmi GraphedTrainStep.node_census: construct with debug=True")
        hip = ctypes.CDLL(_loaded_hip_runtime())          # the runtime torch's graphs belong to, not whatever a bare soname finds
        hip.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        hip.hipGraphNodeGetType.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        names = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record",
                 8: "ext_semaphore_signal", 9: "ext_semaphore_wait", 10: "mem_alloc", 11: "mem_free", 12: "memcpy_from_symbol",
                 13: "memcpy_to_symbol"}
        out = []
        for rec in self.records.values():
            for kind, x in rec["cap"].plan:
                if kind != "graph":
                    continue
                g = ctypes.c_void_p(x.raw_cuda_graph())
                n = ctypes.c_size_t(0)
                if hip.hipGraphGetNodes(g, None, ctypes.byref(n)) != 0:
                    raise RuntimeError("hipGraphGetNodes failed")
                nodes = (ctypes.c_void_p * max(n.value, 1))()
                if n.value and hip.hipGraphGetNodes(g, nodes, ctypes.byref(n)) != 0:
                    raise RuntimeError("hipGraphGetNodes failed")
                counts = {}
                for k in range(n.value):
                    t = ctypes.c_int(-1)
                    if hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[k]), ctypes.byref(t)) != 0:
                        raise RuntimeError("hipGraphNodeGetType failed")
                    name = names.get(t.value, f"type{t.value}")
                    counts[name] = counts.get(name, 0) + 1
                out.append(counts)
        return out

    def dump_graphs(self, directory):
        """Write every recorded graph segment as a DOT file (hipGraphDebugDotPrint; needs `debug=True`) and return the paths
        (diagnostic: which launch a node belongs to)."""
        import os
        if not self.debug:
            raise RuntimeError("hopmi GraphedTrainStep.dump_graphs: construct with debug=True")
        os.makedirs(directory, exist_ok=True)
        paths = []
        for r, rec in enumerate(self.records.values()):
            for k, (kind, x) in enumerate(rec["cap"].plan):
                if kind == "graph":
                    path = os.path.join(directory, f"rec{r}_seg{k:02d}.dot")
                    x.debug_dump(path)
                    paths.append(path)
        return paths

    def _eager(self, epoch, batch):
        self.n_eager += 1
        if self.sharded:
            self.unshard()
        return _steps.train_llm(self.args, epoch, *batch, self.model, self.disc, self.g_opt, self.d_opt, self.accel)

    # -- prototype shards -> full copies ---------------------------------------------------------------------------------
    @torch.no_grad()
    def unshard(self):
        """All-gather the mapping layer's rows (and their Adam moments) from their owners, so that every rank holds the
        current full tensors (call before state_dict() / evaluation when training with more than one rank)."""
        if not (self.sharded and self.world > 1):
            return
        m = self.model
        n = m.mapping_layer.weight.shape[0]
        tensors = [m.mapping_layer.weight, m.mapping_layer.bias]
        for p in (m.mapping_layer.weight, m.mapping_layer.bias):
            st = self.g_opt.state.get(p, {})
            tensors += [st[k] for k in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq") if k in st]
        for t in tensors:
            pad = torch.zeros((self.per * self.world,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            pad[:n].copy_(t)
            all_gather_rows(pad, self.per, self.group)
            t.copy_(pad[:n])
        self.sharded = False


def _loaded_hip_runtime():
    """Path of the libamdhip64 this process has mapped (torch ships its own copy; a second runtime must never be loaded)."""
    with open("/proc/self/maps") as f:
        for line in f:
            path = line.rsplit(" ", 1)[-1].strip()
            if "libamdhip64" in path.rsplit("/", 1)[-1]:
                return path
    raise RuntimeError("hopmi: no libamdhip64 is mapped in this process")


class _PlainBackward:
    def backward(self, loss):
        loss.backward()


def _make_capturable(opt):
    """A captured optimizer step needs its step counters on the device (torch.optim `capturable=True`) and -- the point
    of doing this BEFORE anything is recorded -- its state to exist: torch creates `step`, `exp_avg` and `exp_avg_sq`
    lazily in the first `step()` that sees a gradient (`Adam._init_group`), with `torch.zeros` / `zeros_like`.  Inside a
    capture those zero fills would be recorded and run again on every replay (moments and step count reset every step:
    the update degenerates to lr * sign(g)).  That is what would happen to the discriminator's optimizer, which never
    steps before epoch 11 (train_llm.py:15-36), when the first GAN-phase call is the one that records; and to both
    optimizers with `eager_calls=0`.  Parameters that never receive a gradient get state too; `_init_group` skips a
    parameter whose `.grad` is None regardless of its state, so they stay untouched."""
    if not isinstance(opt, (torch.optim.Adam, torch.optim.AdamW)):
        raise TypeError("hopmi GraphedTrainStep: only torch.optim.Adam / AdamW steps are recorded "
                        f"(got {type(opt).__name__}); use train_llm for other optimizers")
    for g in opt.param_groups:
        g["capturable"] = True
        for p in g["params"]:
            if not p.requires_grad:
                continue
            st = opt.state[p]
            if "step" not in st:
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if g.get("amsgrad", False):
                    st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif torch.is_tensor(st["step"]):
                if st["step"].device != p.device:
                    st["step"] = st["step"].to(p.device)
            else:
                st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32, device=p.device)
