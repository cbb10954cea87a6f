"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce over RCCL/xGMI.

The reference gets its data parallelism implicitly from HF accelerate (DDP with
`find_unused_parameters=True`, run_ted.py:110-112,363-364); there is no explicit collective
in its source.  Replicas are arithmetically independent (per-replica BatchNorm statistics, no
SyncBN), so the only exchange per step is the gradient mean of the parameters that receive a
gradient (65.6 M fp32 for TED; 34 tensors never do: audio_encoder.*, gwnet.residual_convs.*,
gwnet.bn.7.*, gwnet.gconv.7.* -- they are left out of the buckets, which is what
find_unused_parameters achieves in the reference).

Design for xGMI (7 point-to-point links per GPU, ring collectives are per-link bound):
few, large buckets (default 64 MB) launched from autograd hooks as soon as their last gradient
is produced, on RCCL's own stream, so the 183 MB `mapping_layer.weight` all-reduce overlaps with
the rest of backward (the model issues its gwnet branch first in forward, which puts the
mapping-layer branch early in backward).  `GradSync.backward(loss)` has the
`accelerator.backward` shape that `train_llm` expects.
"""
from typing import List, Optional

import torch
import torch.distributed as dist

from . import ops as _ops


class _Bucket:
    """One flat all-reduce buffer and its per-parameter views.  After the exchange the parameters' `.grad` ARE these
    views (no copy back); the copy in is one multi-tensor launch."""

    def __init__(self, params: List[torch.nn.Parameter], dtype):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat = torch.empty(self.numel, dtype=dtype or params[0].dtype, device=params[0].device)
        self.views, off = [], 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        self.pending = len(params)
        self.handle = None
        self.had_grad = [True] * len(params)     # which parameters had a gradient when the bucket was launched


class _Group:
    """The trainable parameters of one module (generator / discriminator); buckets never span
    groups, so a backward that only reaches the discriminator only exchanges its 1 MB."""

    def __init__(self, params, module=None):
        self.params = params
        self.module = module
        self.buckets: Optional[List[_Bucket]] = None
        self.order: List[torch.nn.Parameter] = []     # gradient production order seen in the planning backward


class GradSync:
    """Bucketed, overlapped gradient all-reduce (mean) for a set of modules.

    Usage:  sync = GradSync([model, discriminator]);  train_llm(..., accelerator=sync)
    With world_size == 1 (or torch.distributed not initialised) it is a plain backward.
    """

    takes_only = True            # backward(loss, only=...) is understood (steps._backward)

    def __init__(self, modules, bucket_mb: float = 64.0, grad_dtype: Optional[torch.dtype] = None, group=None,
                 force: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # `force` runs the bucketed path even with one rank (a 1-rank RCCL group): used to exercise hooks, streams and
        # collectives on a single-GPU box
        self.active = self.world > 1 or (force and dist.is_available() and dist.is_initialized())
        # RCCL averages in the collective; gloo (CPU tests) has no AVG: sum, then one in-place scale per bucket
        self._avg = self.active and dist.get_backend(group) == "nccl"
        self.grad_dtype = grad_dtype                 # e.g. torch.bfloat16 halves the xGMI bytes
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.groups: List[_Group] = []
        seen = set()
        for m in modules:
            ps = []
            for p in m.parameters():
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    ps.append(p)
            if ps:
                self.groups.append(_Group(ps, m))
        self._bucket_of = {}
        self._group_of = {}
        self._in_backward = False
        self._skip = set()
        self.bytes_reduced = 0                       # for tests / reporting
        if self.active:
            for g in self.groups:
                for p in g.params:
                    self._group_of[id(p)] = g
                    p.register_post_accumulate_grad_hook(self._on_grad)

    # -- bucket plan of a group: built from the first backward that reaches it -----------------------
    def _plan(self, g: _Group):
        live = [p for p in g.params if p.grad is not None]
        if self.active:                                           # same plan on every rank, or hang
            n = torch.tensor([len(live), sum(p.numel() for p in live)], device=g.params[0].device)
            lo, hi = n.clone(), n.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            if not torch.equal(lo, hi):
                raise RuntimeError("GradSync: ranks disagree on which parameters receive gradients")
        # bucket in the order the planning backward PRODUCED the gradients (identical on every rank: same
        # graph), so each bucket's all-reduce starts as early as its last gradient exists -- for HOP that puts
        # the 183 MB mapping-layer gradient mid-backward, hidden under the gwnet / audio-MLP backward
        seen = {id(p) for p in g.order}
        ordered = [p for p in g.order if p.grad is not None] + [p for p in reversed(live) if id(p) not in seen]
        g.buckets = []
        cur, cur_bytes = [], 0
        for p in ordered:
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > self.bucket_bytes:
                g.buckets.append(_Bucket(cur, self.grad_dtype))
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            g.buckets.append(_Bucket(cur, self.grad_dtype))
        for b in g.buckets:
            for p in b.params:
                self._bucket_of[id(p)] = b

    def _launch(self, b: _Bucket):
        dst, src = [], []
        for i, (p, v) in enumerate(zip(b.params, b.views)):
            b.had_grad[i] = p.grad is not None
            if p.grad is None:                               # same on every rank (same graph): exchanged as zeros,
                v.zero_()                                    # left None afterwards (_collect)
            elif p.grad.data_ptr() != v.data_ptr():          # (already the view: accumulated in place by autograd)
                dst.append(v)
                src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)                   # one multi-tensor launch, converts to grad_dtype if set
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        b.handle = dist.all_reduce(b.flat, op=op, group=self.group, async_op=True)
        self.bytes_reduced += b.flat.numel() * b.flat.element_size()

    def _on_grad(self, p):
        if not self._in_backward:
            return
        g = self._group_of[id(p)]
        if id(g) in self._skip:
            return
        b = self._bucket_of.get(id(p))
        if b is None:                            # group not planned yet: just record the production order
            if g.buckets is None:
                g.order.append(p)
            return
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _collect(self, b: _Bucket):
        b.handle.wait()
        if not self._avg:
            b.flat.mul_(1.0 / self.world)
        # a parameter without a gradient in this backward keeps `.grad is None`, so the optimizer skips it exactly as a
        # single-GPU run does (a zero gradient would still move it through Adam's momentum)
        if b.flat.dtype == b.params[0].dtype:
            for p, v, had in zip(b.params, b.views, b.had_grad):
                if had:
                    p.grad = v                               # gradient lives in the bucket until the next zero_grad
        else:                                                # compressed exchange: widen back into fp32 gradients
            live = [(p, v) for p, v, had in zip(b.params, b.views, b.had_grad) if had]
            torch._foreach_copy_([p.grad for p, _ in live], [v for _, v in live])
        b.handle = None

    # -- accelerator.backward shape (train_llm.py:34,85) -----------------------------------------------
    def backward(self, loss, only=None):
        """`only`: modules whose gradients this backward is FOR (train_llm passes the generator for the generator loss:
        in the GAN phase that backward also reaches the discriminator's parameters, whose gradients nobody uses --
        the next discriminator step zeroes them -- so their buckets are not exchanged)."""
        if not self.active:
            loss.backward()
            return
        wanted = None if only is None else {id(m) for m in only}
        self._skip = set() if wanted is None else {id(g) for g in self.groups if id(g.module) not in wanted}
        for g in self.groups:
            for b in g.buckets or ():
                b.pending = len(b.params)
            if g.buckets is None:
                g.order = []
        self._in_backward = True
        try:
            # RCCL kernels run beside the compute stream from here on: the persistent GRU kernels' co-residency
            # assumption does not hold, use the per-time-step launches for this backward
            with _ops.no_persistent_gru():
                loss.backward()
        finally:
            self._in_backward = False
        for g in self.groups:
            if id(g) in self._skip:
                continue
            if g.buckets is None:                # not planned yet: did this backward reach the group?
                if any(p.grad is not None for p in g.params):
                    self._plan(g)
                    for b in g.buckets:
                        self._launch(b)
                        self._collect(b)
                continue
            for b in g.buckets:
                if b.pending == len(b.params):   # untouched by this backward (e.g. G buckets in the D step)
                    continue
                if b.handle is None:             # partially produced: reduce what is there
                    self._launch(b)
                self._collect(b)
