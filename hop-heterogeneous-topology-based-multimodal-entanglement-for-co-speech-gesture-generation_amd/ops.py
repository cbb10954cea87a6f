"""torch.autograd bindings of the hand-written gfx950 kernels (through the C ABI).

Every op requires ROCm device tensors and the built libhopmi.so; there is deliberately
no eager / CPU implementation here (the CPU oracle lives in oracle/, test-only).
"""
import weakref

import torch

from . import _lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """HIP events around every launch of the named kernels, recorded on the stream the kernel is
    launched on (torch's current stream), so bench.py can report algorithmic bytes / kernel time
    live.  Enable with `ops.TIMER = KernelTimer()`; read with `.summary()` after a synchronize.

    An event pair costs several microseconds of its own (the two marker packets), which matters for 10-20 us
    kernels: every 8th timed launch is followed by an EMPTY pair (two records, nothing between) and the SMALLEST empty
    interval is reported as `event_overhead_ms` (a lower bound of the pair's cost, so the corrected kernel time is an
    upper bound); `total_ms` is the raw sum, `kernel_ms` has launches x overhead taken off (this is the figure that
    agrees with rocprofv3's kernel durations)."""

    def __init__(self):
        self.spans = {}
        self.empty = []
        self.exact = set()
        self._n = 0

    def launch_exact(self, name, nbytes, flops, fn, extra=0):
        """For entry points that support hopmi_time_next_launch: the events are recorded by the dispatch itself
        (kernel begin / end), so there is no pair overhead to take off."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()                     # materialise the hipEvent_t handles
        _lib.check(_lib.lib().hopmi_time_next_launch(e0.cuda_event, e1.cuda_event), "hopmi_time_next_launch")
        rc = fn()
        self.spans.setdefault(name, []).append((e0, e1, nbytes, flops, extra))
        self.exact.add(name)
        return rc

    def launch(self, name, nbytes, flops, fn, extra=0):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn()
        e1.record()
        self.spans.setdefault(name, []).append((e0, e1, nbytes, flops, extra))
        self._n += 1
        if self._n % 8 == 0:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            b.record()
            self.empty.append((a, b))
        return rc

    def summary(self):
        gaps = sorted(a.elapsed_time(b) for a, b in self.empty)
        over = gaps[0] if gaps else 0.0
        out = {}
        for name, spans in self.spans.items():
            ms = sum(s[0].elapsed_time(s[1]) for s in spans)
            o = 0.0 if name in self.exact else over
            out[name] = dict(launches=len(spans), total_ms=ms, kernel_ms=max(ms - len(spans) * o, 0.0),
                             event_overhead_ms=o, bytes=sum(s[2] for s in spans), flops=sum(s[3] for s in spans),
                             extra_bytes=sum(s[4] for s in spans))
        return out


TIMER = None


class no_timer:
    """Within the block no launch is timed (event records do not belong inside a stream capture)."""

    def __enter__(self):
        global TIMER
        self.prev, TIMER = TIMER, None

    def __exit__(self, *exc):
        global TIMER
        TIMER = self.prev


def _timed(name, nbytes, flops, fn, exact=False, extra=0):
    """`nbytes` = algorithmic bytes of the launch; `extra` = bytes it moves on top of those by the
    build's own choice (tensors saved for the backward), reported separately."""
    if TIMER is None:
        return fn()
    return TIMER.launch_exact(name, nbytes, flops, fn, extra) if exact else TIMER.launch(name, nbytes, flops, fn, extra)


def time_noop_launch():
    """Launch an empty kernel through the timer's exact path (when a KernelTimer is active): its reported duration is
    the floor of the dispatch-event timing method, recorded next to the kernels it is compared with."""
    L, st = _lib.lib(), _stream()
    _lib.check(_timed("noop", 0, 0, lambda: L.hopmi_noop_launch(st), exact=True), "hopmi_noop_launch")


# ------------------------------------------------------- Adam over a parameter list in one launch (csrc/adam.hip)
FUSED_ADAM = __import__("os").environ.get("HOPMI_FUSED_ADAM", "1") != "0"


def adam_multi_supported(opt) -> bool:
    """Can hopmi_adam_multi stand in for `opt.step()`?  Plain torch.optim.Adam (not AdamW), no weight decay, amsgrad, maximize or
    differentiable mode, scalar hyper-parameters, fp32 parameters on one ROCm device with device-side step counters."""
    if not FUSED_ADAM or type(opt) is not torch.optim.Adam:
        return False
    for g in opt.param_groups:
        if (g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False) or g.get("differentiable", False)
                or g.get("decoupled_weight_decay", False) or torch.is_tensor(g["lr"]) or any(torch.is_tensor(b) for b in g["betas"])):
            return False
        for p in g["params"]:
            if p.requires_grad and not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                return False
    return True


class AdamPlan:
    """The device tables of one hopmi_adam_multi launch over `params`: (p, g, m, v, n) per tensor and the (tensor, chunk) work
    items.  Made BEFORE a recording (pinned host buffers and device tables are allocated here, outside any capture); `bind(grads)`
    is called where the gradients' addresses are final -- inside the recording, at the optimizer's step -- and enqueues one small
    host -> device copy of the tensor table (a copy node of the recording; the pinned buffer lives as long as the plan).  A
    parameter without a gradient in that step gets n = 0 (its work items do nothing)."""

    def __init__(self, params, exp_avgs, exp_avg_sqs):
        import numpy as np
        L = _lib.lib()
        chunk = L.hopmi_adam_chunk()
        self.params = list(params)
        n = len(self.params)
        self._tab_host = torch.zeros(n, 5, dtype=torch.int64).pin_memory()
        items = []
        for i, (p, m, v) in enumerate(zip(self.params, exp_avgs, exp_avg_sqs)):
            for t, nm in ((p, "parameter"), (m, "exp_avg"), (v, "exp_avg_sq")):
                if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()):
                    raise _lib.HopmiError(f"hopmi AdamPlan: {nm} {i} is not a contiguous fp32 device tensor of the parameter's size")
            self._tab_host[i, 0], self._tab_host[i, 2], self._tab_host[i, 3] = p.data_ptr(), m.data_ptr(), v.data_ptr()
            items += [(i, c) for c in range((p.numel() + chunk - 1) // chunk)]
        dev = self.params[0].device
        self.tensors = torch.zeros(n, 5, dtype=torch.int64, device=dev)
        self.items = torch.from_numpy(np.asarray(items, dtype=np.int32).reshape(-1, 2)).to(dev)
        self.n_items = int(self.items.shape[0])
        self.keep = (self.params, list(exp_avgs), list(exp_avg_sqs))
        self._grads = None

    def bind(self, grads, exp_avgs, exp_avg_sqs):
        """grads[i]: the gradient of params[i] (contiguous fp32, its size) or None; the moments as the optimizer holds them NOW
        (every address is written again: a parameter re-packed or a state loaded since the plan was made is followed)."""
        for i, (p, g, m, v) in enumerate(zip(self.params, grads, exp_avgs, exp_avg_sqs)):
            if g is None:
                self._tab_host[i, 1], self._tab_host[i, 4] = 0, 0
                continue
            for t, nm in ((g, "gradient"), (m, "exp_avg"), (v, "exp_avg_sq")):
                if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()):
                    raise _lib.HopmiError(f"hopmi AdamPlan: {nm} {i} is not a contiguous fp32 device tensor of the parameter's size")
            self._tab_host[i, 0], self._tab_host[i, 1], self._tab_host[i, 2], self._tab_host[i, 3] = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
            self._tab_host[i, 4] = p.numel()
        self.keep = (self.params, list(grads), list(exp_avgs), list(exp_avg_sqs))      # (the table holds their addresses)
        self.tensors.copy_(self._tab_host, non_blocking=True)

    def step(self, lr, beta1, beta2, eps, step_tensor):
        _lib.check(_lib.lib().hopmi_adam_multi(self.tensors.data_ptr(), self.items.data_ptr(), self.n_items, float(lr), float(beta1),
                                               float(beta2), float(eps), step_tensor.data_ptr(), _stream()), "hopmi_adam_multi")


# ------------------------------------------------------- diagnostic build: the fp16-split status word (csrc/common.h, split_check)
_SPLIT_FILES = {1: "gemm.hip", 2: "gemm_tn.hip", 3: "elementwise.hip", 4: "attn.hip", 5: "bert_attn.hip", 6: "gru.hip", 7: "wavenet.hip",
                8: "wavenet_stack.hip"}
_SPLIT_STATUS = None


def split_status(reset=False):
    """Diagnostic library only (`make dbg`, HOPMI_LIB=.../libhopmi_dbg.so): what the fp16 hi/lo splits of every kernel reported since
    the last reset -- {"overflow": (site, count, value) or None, "nonfinite_in": (site, count, value) or None}, `site` = "file:line" of
    the FIRST conversion in stream order whose hi part was infinity / NaN ("overflow": on a finite input, i.e. an operand scale that
    does not hold for the data; "nonfinite_in": the input already was).  The first call registers the buffer with every translation
    unit.  Returns None with the production library (which compiles the check away).  Synchronises the device."""
    global _SPLIT_STATUS
    import ctypes
    import struct
    L = _lib.lib()
    if not hasattr(L, "hopmi_debug_set_split_status_gemm"):
        return None
    if _SPLIT_STATUS is None:
        buf = torch.zeros(16, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for name in ("gemm", "gemm_tn", "elementwise", "attn", "bert_attn", "gru", "wavenet", "wavenet_stack"):
            fn = getattr(L, "hopmi_debug_set_split_status_" + name)
            fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
            if fn(buf.data_ptr()) != 0:
                raise _lib.HopmiError(f"hopmi: hopmi_debug_set_split_status_{name} failed")
        _SPLIT_STATUS = buf
    torch.cuda.synchronize()
    w = [x & 0xFFFFFFFF for x in _SPLIT_STATUS.tolist()]

    def dec(o):
        if w[o + 1] == 0:
            return None
        return (f"{_SPLIT_FILES.get(w[o] >> 16, w[o] >> 16)}:{w[o] & 0xFFFF}", w[o + 1], struct.unpack("f", struct.pack("I", w[o + 2]))[0])

    out = {"overflow": dec(0), "nonfinite_in": dec(4)}
    if w[8]:      # the image-emitting GEMM epilogue's first bound violation: where, and what the bound was made of
        f = lambda u: struct.unpack("f", struct.pack("I", u))[0]
        out["ab_img"] = dict(row=w[9], col=w[10], row_norm=f(w[11]), mul=f(w[12]), add=f(w[13]), out=f(w[14]), epilogue=w[15])
    if reset:
        _SPLIT_STATUS.zero_()
    return out


def gcn_algorithmic_bytes(n_slabs: int, V: int) -> int:
    """read x' + write h = 2*64*V*4 B per slab, plus the per-launch constants
    (A: V*V*4 B twice, Wm+bm: 49 408 B)."""
    return n_slabs * 2 * 64 * V * 4 + 2 * V * V * 4 + 64 * 192 * 4 + 64 * 4


def gcn_flops(n_slabs: int, V: int) -> int:
    """per slab V*(2*192*64) + 2*(2*64*V*V)."""
    return n_slabs * (V * 2 * 192 * 64 + 4 * 64 * V * V)


# Mixed precision (steps.mixed_precision / torch.autocast): the library GEMMs run in bf16, the HIP kernels stay
# fp32 -- every Function below casts its floating inputs to fp32 and runs with autocast disabled.
_fwd32 = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd32 = torch.amp.custom_bwd(device_type="cuda")


def _dev_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.HopmiError(f"hopmi: `{name}` is on {t.device}; the hot path only runs on a ROCm device "
                              "(no CPU fallback)")
    if t.dtype != torch.float32:
        raise _lib.HopmiError(f"hopmi: `{name}` must be float32, got {t.dtype}")
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


_fwd_any = torch.amp.custom_fwd(device_type="cuda")      # no input cast: the graph-wavenet kernels take fp32 OR bf16 activations


def _dev_act(t: torch.Tensor, name: str):
    """An activation tensor of the graph-wavenet kernels: contiguous fp32 or bf16 on the device -> (tensor, dtype code of the
    `_dt` entry points: 0 = fp32, 1 = bf16).  The kernels read / write it as it is (bf16: the storage form of BASELINE.json
    configs 2 / 4, arithmetic stays fp32)."""
    if not t.is_cuda:
        raise _lib.HopmiError(f"hopmi: `{name}` is on {t.device}; the hot path only runs on a ROCm device (no CPU fallback)")
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.HopmiError(f"hopmi: `{name}` must be float32 or bfloat16, got {t.dtype}")
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t, (1 if t.dtype == torch.bfloat16 else 0)


def _f32_param(t, name):
    return _dev_f32(t if t.dtype == torch.float32 else t.float(), name)


def gcn_prepare(A1: torch.Tensor, A2: torch.Tensor) -> torch.Tensor:
    """Padded on-chip images of the node-mix matrices (hopmi_gcn_prepare); build once per forward
    pass and hand to every `gcn(..., prep=...)` call of that pass.  Not differentiable: gradients
    w.r.t. A1 / A2 come out of the gcn backward kernel."""
    A1, A2 = _dev_f32(A1.detach(), "A1"), _dev_f32(A2.detach(), "A2")
    V = A1.shape[0]
    if A1.shape != (V, V) or A2.shape != (V, V):
        raise _lib.HopmiError(f"hopmi gcn_prepare: A1{tuple(A1.shape)} A2{tuple(A2.shape)} must both be (V,V)")
    L = _lib.lib()
    n = L.hopmi_gcn_prep_floats(V)
    if n == 0:
        raise _lib.HopmiError(f"hopmi gcn_prepare: V={V} unsupported (1..48)")
    prep = torch.empty(n, dtype=torch.float32, device=A1.device)
    _lib.check(L.hopmi_gcn_prepare(A1.data_ptr(), A2.data_ptr(), prep.data_ptr(), V, _stream()), "hopmi_gcn_prepare")
    return prep


class _GcnFn(torch.autograd.Function):
    """h = Wm.[x ; xA1 ; xA2] + bm on channels-last slabs (gwnet.py:24-46)."""

    @staticmethod
    @_fwd_any
    def forward(ctx, x, A1, A2, Wm, bm, prep):
        x, dt = _dev_act(x, "x")                         # fp32, or bf16 as a bf16 GEMM left it (hopmi_gcn_fwd_dt)
        Wm, bm = _f32_param(Wm, "Wm"), _f32_param(bm, "bm")
        V = A1.shape[0]
        if x.shape[-1] != 64 or x.shape[-2] != V or A1.shape != (V, V) or A2.shape != (V, V):
            raise _lib.HopmiError(f"hopmi gcn: bad shapes x{tuple(x.shape)} A1{tuple(A1.shape)} A2{tuple(A2.shape)}")
        if Wm.numel() != 64 * 192 or bm.numel() != 64:
            raise _lib.HopmiError(f"hopmi gcn: Wm must have 64*192 elements, bm 64 (got {Wm.numel()}, {bm.numel()})")
        L, st = _lib.lib(), _stream()
        if prep.numel() != L.hopmi_gcn_prep_floats(V) or not prep.is_cuda:
            raise _lib.HopmiError("hopmi gcn: `prep` does not come from gcn_prepare for this V")
        n_slabs = x.numel() // (V * 64)
        h = torch.empty_like(x)
        _lib.check(_timed("gcn_fwd", gcn_algorithmic_bytes(n_slabs, V) // (2 if dt else 1), gcn_flops(n_slabs, V),
                          lambda: L.hopmi_gcn_fwd_dt(x.data_ptr(), prep.data_ptr(), Wm.data_ptr(), bm.data_ptr(),
                                                     h.data_ptr(), n_slabs, V, dt, st)), "hopmi_gcn_fwd")
        ctx.save_for_backward(x, prep, Wm)
        ctx.wm_shape, ctx.V, ctx.dt = Wm.shape, V, dt
        return h

    @staticmethod
    @_bwd32
    def backward(ctx, dh):
        x, prep, Wm = ctx.saved_tensors
        dh, _ = _dev_act(dh if dh.dtype == x.dtype else dh.to(x.dtype), "dh")
        V, dt = ctx.V, ctx.dt
        n_slabs = x.numel() // (V * 64)
        L = _lib.lib()
        ws = torch.empty(L.hopmi_gcn_bwd_ws_floats(n_slabs, V), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        dA1 = torch.empty(V, V, dtype=torch.float32, device=x.device)
        dA2 = torch.empty_like(dA1)
        dWm = torch.empty(ctx.wm_shape, dtype=torch.float32, device=x.device)
        dbm = torch.empty(64, dtype=torch.float32, device=x.device)
        st = _stream()
        # backward: read x', dh, write dx' (+ tiny dA, dWm, dbm) = 1.5x forward bytes, ~2x forward FLOPs
        _lib.check(_timed("gcn_bwd", n_slabs * 3 * 64 * V * (2 if dt else 4), 2 * gcn_flops(n_slabs, V),
                          lambda: L.hopmi_gcn_bwd_dt(x.data_ptr(), dh.data_ptr(), prep.data_ptr(), Wm.data_ptr(),
                                                     dx.data_ptr(), dA1.data_ptr(), dA2.data_ptr(),
                                                     dWm.data_ptr(), dbm.data_ptr(), ws.data_ptr(), n_slabs, V, dt, st)),
                   "hopmi_gcn_bwd")
        return dx, dA1, dA2, dWm, dbm, None


def gcn(x: torch.Tensor, A1: torch.Tensor, A2: torch.Tensor, Wm: torch.Tensor, bm: torch.Tensor,
        prep: torch.Tensor = None) -> torch.Tensor:
    """x (..., V, 64) channels-last; A1 = adp, A2 = adp @ adp (V,V); Wm (64,192[,1,1]); bm (64,).
    `prep` = gcn_prepare(A1, A2), shared by all calls of one forward pass (built here if absent)."""
    if prep is None:
        prep = gcn_prepare(A1, A2)
    return _GcnFn.apply(x, A1, A2, Wm, bm, prep)


# ------------------------------------------------------- BatchNorm1d on channels-last rows (the discriminator's pre_conv)
class _BnClFn(torch.autograd.Function):
    """Training-mode BatchNorm1d over the rows of x (..., C), C <= 64 (hopmi_bn_cl_fwd / _bwd: one launch each); the running
    statistics are advanced in the forward, as torch does."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, gamma, beta, rmean, rvar, eps, momentum):
        x = _dev_f32(x, "x")
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        save = torch.empty(2, C, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().hopmi_bn_cl_fwd(x.data_ptr(), _dev_f32(gamma.detach(), "weight").data_ptr(), _dev_f32(beta.detach(), "bias").data_ptr(),
                                              _ptr(rmean), _ptr(rvar), y.data_ptr(), save.data_ptr(), M, C, eps, momentum, 1, _stream()),
                   "hopmi_bn_cl_fwd")
        ctx.save_for_backward(x, gamma, save)
        return y

    @staticmethod
    @_bwd32
    def backward(ctx, dy):
        x, gamma, save = ctx.saved_tensors
        dy = _dev_f32(dy, "dy")
        C = x.shape[-1]
        M = x.numel() // C
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dg, db = torch.empty_like(gamma), torch.empty_like(gamma)
        _lib.check(_lib.lib().hopmi_bn_cl_bwd(x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), save.data_ptr(), _ptr(dx), dg.data_ptr(),
                                              db.data_ptr(), M, C, _stream()), "hopmi_bn_cl_bwd")
        return dx, dg, db, None, None, None, None


def batch_norm_cl(x, bn, training):
    """torch.nn.BatchNorm1d `bn` applied to channels-last x (B, T, C) (statistics over B and T): training-mode forward (running
    statistics advanced, num_batches_tracked counted) or the running-statistics map, fp32."""
    x = x.float()
    C = x.shape[-1]
    if not (bn.affine and bn.track_running_stats and bn.momentum is not None and bn.running_mean is not None and C <= 64 and x.is_cuda):
        # what the kernels do not cover (no affine parameters, no running statistics, cumulative-average momentum, > 64 channels):
        # the module itself on the (B, C, T) view
        if bool(training) == bool(bn.training):
            return bn(x.transpose(1, 2)).transpose(1, 2)
        # the caller's mode, not the module's own flag: the functional form with the module's buffers (batch statistics also when the
        # module tracks none; the cumulative average when momentum is None)
        use_batch = bool(training) or bn.running_mean is None
        mom = bn.momentum
        if training and bn.track_running_stats and bn.num_batches_tracked is not None:
            with torch.no_grad():
                bn.num_batches_tracked += 1
            if mom is None:
                mom = 1.0 / float(bn.num_batches_tracked)
        return torch.nn.functional.batch_norm(x.transpose(1, 2), bn.running_mean if (not training or bn.track_running_stats) else None,
                                              bn.running_var if (not training or bn.track_running_stats) else None, bn.weight, bn.bias,
                                              use_batch, 0.0 if mom is None else float(mom), float(bn.eps)).transpose(1, 2)
    if training:
        with torch.no_grad():
            bn.num_batches_tracked += 1
        return _BnClFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps), float(bn.momentum))
    if torch.is_grad_enabled() and (x.requires_grad or bn.weight.requires_grad or bn.bias.requires_grad):
        # the running-statistics map with a gradient (an eval-mode discriminator in front of a generator loss): plain tensor
        # operations, autograd gives dx = dy * scale and the affine parameters' gradients
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        return x * scale + (bn.bias - bn.running_mean * scale)
    y = torch.empty_like(x)
    _lib.check(_lib.lib().hopmi_bn_cl_fwd(_dev_f32(x, "x").data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                          bn.running_var.data_ptr(), y.data_ptr(), None, x.numel() // C, C, float(bn.eps), 0.0, 0, _stream()),
               "hopmi_bn_cl_fwd")
    return y


def batch_norm_cl_statistics(x, bn):
    """Only the lasting effect of a training-mode BatchNorm1d forward whose output is discarded: the running-statistics update."""
    x = _dev_f32(x.detach().float(), "x")
    C = x.shape[-1]
    with torch.no_grad():
        bn.num_batches_tracked += 1
    save = torch.empty(2, C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().hopmi_bn_cl_fwd(x.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                          bn.running_var.data_ptr(), None, save.data_ptr(), x.numel() // C, C, float(bn.eps), float(bn.momentum), 1,
                                          _stream()), "hopmi_bn_cl_fwd")
    return save


# ------------------------------------------------------- leaf cut of the recorded backward (graph.GraphedTrainStep)
CUT_HOOK = None     # while a step with an overlapped gradient exchange is being recorded: tensor -> detached leaf


def cut_point(t):
    """Identity -- except while hopmi.GraphedTrainStep records a step whose gradient exchange is overlapped with the backward:
    there the tensor is replaced by a detached leaf, so that the backward of everything behind this point (decoder GRU, head,
    losses) ends here, its gradients can be exchanged, and the backward of everything in front of it continues from the leaf's
    gradient while that exchange is on the wire (graph._Capture.backward)."""
    h = CUT_HOOK
    if h is None or t is None or not torch.is_grad_enabled() or not t.requires_grad:
        return t
    return h(t)


# ------------------------------------------------------- frozen-weight linears on split-bf16 MFMA (hopmi_gemm_split)
CAST_CACHE_ENABLED = __import__("os").environ.get("HOPMI_CAST_CACHE", "1") != "0"
# ---- strict fp32 (test oracle) ------------------------------------------------------------------------------------------------
# Rounds 2-4 took the contractions INSIDE three kernel families (fused WaveNet forward, reprogramming attention, persistent GRU
# recurrences) as three-term split-bf16 products (~2^-16 per product), and this switch routed them to fp32-exact compositions the
# repository also carries to say what the default owed to that.  Since round 5 the default kernels carry scaled fp16 hi/lo operands
# (csrc/f16_dev.h: fp32-equivalent, tests/test_gpu_parity.py::test_*_vs_float64 hold them to 4 x plain fp32's error), so the switch
# no longer changes the accuracy class.  It stays as an independently written second evaluation of the same operators for the tests:
# the WaveNet block as the composition of the exact-fp32 graph-conv kernel with library GEMMs and torch's BatchNorm
# (gwnet.forward_cl's composed branch), the GRU recurrences as per-time-step launches of the exact-fp32 kernel, the reprogramming
# attention as fp32 tensor operations with materialised scores.
STRICT_FP32 = False
_GRU_SMALL_ENV = None          # the user's HOPMI_GRU_SMALL while the switch is on


def strict_fp32(on=None):
    """Route the WaveNet forward, reprogramming attention and GRU recurrences to their composed fp32-exact forms (see above).
    Returns the previous setting."""
    global STRICT_FP32, _GRU_SMALL_ENV
    prev = STRICT_FP32
    if on is not None and bool(on) != STRICT_FP32:
        import os
        STRICT_FP32 = bool(on)
        if STRICT_FP32:                                    # (the small-hidden-size GRU kernels are the fused form too)
            _GRU_SMALL_ENV = os.environ.get("HOPMI_GRU_SMALL")
            os.environ["HOPMI_GRU_SMALL"] = "0"
        elif _GRU_SMALL_ENV is None:
            os.environ.pop("HOPMI_GRU_SMALL", None)
        else:
            os.environ["HOPMI_GRU_SMALL"] = _GRU_SMALL_ENV
        _lib.lib().hopmi_reload_env()
    return prev


F16_PARTS = 16      # the `parts` code of the fp16 hi/lo form (two scaled fp16 parts per operand, three MFMA terms, fp32-equivalent)
# 16: fp16 hi/lo, three terms (fp32-equivalent, default since round 4);  3: six-term bf16 split (fp32-equivalent);
# 2: three-term bf16 split (2^-16 class);  0: library fp32 GEMM (hipBLASLt)
def _env_gemm_parts():
    raw = __import__("os").environ.get("HOPMI_GEMM_PARTS", "16")
    if raw not in ("0", "2", "3", "16"):
        raise ValueError(f"HOPMI_GEMM_PARTS={raw!r}: 0 (library fp32), 2, 3 (bf16 parts) or 16 (fp16 hi/lo)")
    return int(raw)


GEMM_PARTS = _env_gemm_parts()


def gemm_parts(parts=None):
    """Select how the frozen BERT's linears are computed (bert_fast): 0 = the library's fp32 GEMM, 16 = hopmi_gemm_f16x2 (two
    scaled fp16 parts per operand, three MFMA terms, fp32-equivalent), 3 = hopmi_gemm_split with three bf16 parts per operand
    (six MFMA terms, fp32-equivalent), 2 = two bf16 parts (three terms, 2^-16-class products).  Returns the previous setting;
    `None` only reads it."""
    global GEMM_PARTS
    prev = GEMM_PARTS
    if parts is not None:
        if parts not in (0, 2, 3, F16_PARTS):
            raise ValueError("hopmi gemm_parts: 0 (library fp32), 2, 3 or 16 (fp16 hi/lo)")
        GEMM_PARTS = parts
    return prev


def split_weight_image(w: torch.Tensor, parts: int) -> torch.Tensor:
    """bf16 part images of a frozen (N, K) fp32 weight for hopmi_gemm_split (as its Bt operand)."""
    w = _dev_f32(w.detach(), "weight")
    N, K = w.shape
    L = _lib.lib()
    nbytes = L.hopmi_gemm_split_image_bytes(N, K, parts)
    if nbytes == 0 or K % 2:
        raise _lib.HopmiError(f"hopmi split_weight_image: unsupported shape {tuple(w.shape)} / parts {parts}")
    img = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    _lib.check(L.hopmi_gemm_split_prepare(w.data_ptr(), N, K, parts, img.data_ptr(), _stream()), "hopmi_gemm_split_prepare")
    return img


def split_gemm_supported(N: int, K: int) -> bool:
    if GEMM_PARTS == F16_PARTS:                      # (the fp16 form pads its weight image to whole tiles)
        return K % 4 == 0 and N % 2 == 0
    return N % 128 == 0 and K % 32 == 0


def row_scales(a2d):
    """Per-row power-of-two scales [2][M] of a2d (hopmi_row_scales): the A operand's scales of hopmi_gemm_f16x2."""
    M, K = a2d.shape
    sc = torch.empty(2, M, dtype=torch.float32, device=a2d.device)
    _lib.check(_lib.lib().hopmi_row_scales(a2d.data_ptr(), M, K, sc.data_ptr(), _stream()), "hopmi_row_scales")
    return sc


# ------------------------------------------------------- derived-operand caches: one invalidation point, one paranoid mode
# Everything this package derives from a tensor and keeps beside it -- the row scales / fp16 operand images producers attach to
# their outputs (`_hopmi_rs`, `_hopmi_img`), bf16 casts of parameters (_CAST_CACHE), fp16 hi/lo images of trainable and frozen
# weights (_F16_IMG, _F16_IMG_FROZEN), constant scale tables (_UNIT_RS), and the frozen BERT's fused QKV weight, images, FFN bounds
# and bf16 copies (bert_fast.FrozenBertEncoder) -- is valid for ONE (owner identity, version counter, CACHE_EPOCH) triple.
# `reset_all_caches()` is the single invalidation point: scope "recording" forgets what must not cross a recording's boundary
# (graph.GraphedTrainStep, both ends and after a failed recording); scope "all" (invalidate_weight_images: a write through `.data`
# that no version counter saw) additionally moves CACHE_EPOCH, which every table and the encoder's private caches carry in their key.
# HOPMI_CACHE_CHECK=1 (`cache_check(True)`): every HIT of those caches outside a stream capture is verified against a fresh
# computation -- bitwise for scales, weight images and casts; for operand images by decoding them (a producer may have chosen
# another power-of-two scale than the row maximum's: hi + lo must reproduce the tensor to 2^-21 of the row's scaled maximum and be
# finite) -- and raises HopmiError naming the cache.  The GPU suite runs the recorded-step and soak tests once under it.
CACHE_EPOCH = 0
CACHE_CHECK = __import__("os").environ.get("HOPMI_CACHE_CHECK", "0") == "1"
_CACHE_CHECKS_DONE = 0          # hits verified so far (tests assert the mode did something)
_EXTRA_RESETTERS = []           # callables(scope) of caches that live outside this module


def cache_check(on=None):
    """Switch the paranoid mode of the derived-operand caches; returns the previous setting (`None` only reads it)."""
    global CACHE_CHECK
    prev = CACHE_CHECK
    if on is not None:
        CACHE_CHECK = bool(on)
    return prev


def cache_checks_done():
    return _CACHE_CHECKS_DONE


def register_cache_resetter(fn):
    """`fn(scope)` is called by reset_all_caches (scope "recording" or "all")."""
    _EXTRA_RESETTERS.append(fn)


def reset_all_caches(scope="all"):
    """THE invalidation point of every derived-operand cache (see above).  scope "recording": what an eager call made must not be
    served to a recording and vice versa (casts and trainable-weight images: dropped; frozen-weight images first BUILT under capture:
    dropped -- their prepare launch was only recorded and their memory is the graph pool's).  scope "all": everything, and
    CACHE_EPOCH moves so that caches held elsewhere (the frozen BERT encoder's, tensor-attached scales / images) stop matching."""
    global CACHE_EPOCH
    if scope not in ("recording", "all"):
        raise ValueError(f"hopmi reset_all_caches: scope {scope!r}")
    _CAST_CACHE.clear()
    _F16_IMG.clear()
    if scope == "all":
        # (graph.GraphedTrainStep keeps CACHE_EPOCH beside the frozen parameters' version counters: a recording made before this
        # call holds the addresses of the images dropped here and is re-made on its next call.  _UNIT_RS stays: constants.)
        CACHE_EPOCH += 1
        _F16_IMG_FROZEN.clear()
    else:
        for key in [k for k, v in _F16_IMG_FROZEN.items() if v[3]]:
            del _F16_IMG_FROZEN[key]
    for fn in list(_EXTRA_RESETTERS):
        fn(scope)


def _checking():
    return CACHE_CHECK and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())


def _check_fail(what, detail):
    raise _lib.HopmiError(f"hopmi HOPMI_CACHE_CHECK: stale or wrong cached operand -- {what}: {detail}")


def _check_count():
    global _CACHE_CHECKS_DONE
    _CACHE_CHECKS_DONE += 1


def _check_equal(what, cached, fresh):
    _check_count()
    if cached.shape != fresh.shape or cached.dtype != fresh.dtype or not torch.equal(cached, fresh):
        _check_fail(what, f"cached {tuple(cached.shape)} {cached.dtype} differs from a fresh computation")


def _check_scales(what, t2d, sc):
    """Scales a producer attached to `t2d` still describe it: (power of two, inverse) pairs, equal to the row-maximum scale of a fresh
    pass or -- a producer that scaled from an a-priori bound -- up to 2^12 below it (never above: that row would overflow fp16)."""
    _check_count()
    fresh = row_scales(t2d)
    if not bool(((sc[0].double() * sc[1].double()) == 1.0).all()):
        _check_fail(what, "not (scale, inverse) pairs")
    ratio = fresh[0].double() / sc[0].double()
    ok = ((ratio >= 1.0) & (ratio <= 4096.0)) | (t2d.abs().amax(dim=1) == 0)      # (an all-zero row: any scale describes it)
    if not bool(ok.all()):
        r = int(torch.nonzero(~ok)[0])
        _check_fail(what, f"row {r}: attached scale {float(sc[0][r]):g} against {float(fresh[0][r]):g} from the values")


def _decode_rows_image(img, M, K):
    """(hi + lo) of a tile-blocked fp16 rows image (csrc/gemm.hip f16_blk) as an fp32 (M, Kp) matrix + the hi part (for finiteness)."""
    Kp = (K + 31) // 32 * 32
    Mp = (M + 127) // 128 * 128
    h = img.view(torch.float16).view(2, Mp // 128, Kp // 32, 128, 32).permute(0, 1, 3, 2, 4).reshape(2, Mp, Kp)
    return (h[0, :M].float() + h[1, :M].float()), h[0, :M]


def _check_image(what, t2d, img, sc):
    """The image + scales a producer attached to `t2d` (M, K) describe it: hi finite, (hi + lo) / s == t to 2^-21 of the row's
    scaled maximum (+ the fp16 subnormal floor), s a power of two with its inverse beside it."""
    _check_count()
    M, K = t2d.shape
    val, hi = _decode_rows_image(img, M, K)
    s, inv = sc[0].double(), sc[1].double()
    if not bool(torch.isfinite(hi).all()):
        _check_fail(what, "non-finite hi part in the operand image")
    if not bool(((s * inv) == 1.0).all()) or not bool((torch.frexp(sc[0])[0] == 0.5).all()):
        _check_fail(what, "row scales are not (power of two, inverse) pairs")
    want = t2d.double() * s.unsqueeze(1)
    tol = want.abs().amax(dim=1, keepdim=True) * 2.0 ** -21 + 2.0 ** -23
    bad = (val[:, :K].double() - want).abs() > tol
    if K < val.shape[1] and bool((val[:, K:] != 0).any()):
        _check_fail(what, "non-zero pad columns in the operand image")
    if bool(bad.any()):
        r = int(torch.nonzero(bad.any(dim=1))[0])
        _check_fail(what, f"image row {r} does not reproduce the tensor ({int(bad.sum())} elements off)")


def _attach_rs(t, sc):
    """Remember the fp16-form row scales `sc` ([2][M]) of tensor `t` ON the tensor object, with what identifies the values they
    were taken from (storage address, version counter): the kernel that produced `t` had its rows in registers, and the GEMM that
    consumes `t` (possibly on the other side of an autograd edge: the Python object travels) asks `_take_rs`."""
    t._hopmi_rs = (sc, t.data_ptr(), t._version, CACHE_EPOCH)


def _take_rs(t, M):
    """The row scales attached to `t` if they still describe it, else None (the consumer then runs hopmi_row_scales itself)."""
    hit = getattr(t, "_hopmi_rs", None)
    if hit is None:
        return None
    sc, ptr, ver, epoch = hit
    if ptr != t.data_ptr() or ver != t._version or epoch != CACHE_EPOCH or tuple(sc.shape) != (2, M) or sc.device != t.device:
        return None
    if _checking() and t.dtype == torch.float32 and t.numel() % M == 0 and (t.numel() // M) % 4 == 0 and t.is_contiguous() and t.data_ptr() % 16 == 0:
        _check_scales("row scales attached to a tensor (_take_rs)", t.detach().reshape(M, -1), sc)
    return sc


def _attach_img(t, img, sc, norms=None):
    """Remember the fp16 hi / lo operand image `img` (tile-blocked, hopmi_rows_image_f16_bytes) and its row scales `sc` of tensor `t` ON
    the tensor object (as _attach_rs does for the scales alone): the LayerNorm kernels write it beside `t`, the GEMM that consumes `t`
    asks `_take_img` and runs the LDS-DMA form (hopmi_gemm_f16x2_ab_ep).  `norms`: the rows' 2-norms ([M], `_take_norms`)."""
    t._hopmi_img = (img, sc, t.data_ptr(), t._version, norms, CACHE_EPOCH)
    _attach_rs(t, sc)


def _take_norms(t, M):
    hit = getattr(t, "_hopmi_img", None)
    if hit is None or hit[4] is None or hit[2] != t.data_ptr() or hit[3] != t._version or hit[5] != CACHE_EPOCH or tuple(hit[4].shape) != (M,):
        return None
    if _checking() and t.dtype == torch.float32 and t.is_contiguous():
        # the norms feed an a-priori BOUND (gemm.hip, image epilogue): never below the true norm, never far above it
        _check_count()
        true = t.detach().reshape(M, -1).double().norm(dim=1)
        got = hit[4].double()
        big = true > 1e-37
        if not bool((got[big] >= true[big]).all()) or not bool((got <= true * (1 + 1e-5) + 1e-44).all()):
            _check_fail("row norms attached to a tensor (_take_norms)", "they do not bound the rows' 2-norms")
    return hit[4]


def _take_img(t, M, K):
    hit = getattr(t, "_hopmi_img", None)
    if hit is None or not IMG_FUSED:
        return None
    img, sc, ptr, ver, _, epoch = hit
    if (ptr != t.data_ptr() or ver != t._version or epoch != CACHE_EPOCH or img.numel() != _lib.lib().hopmi_rows_image_f16_bytes(M, K)
            or tuple(sc.shape) != (2, M) or img.device != t.device):
        return None
    if _checking() and t.dtype == torch.float32 and t.is_contiguous():
        _check_image("operand image attached to a tensor (_take_img)", t.detach().reshape(M, K), img, sc)
    return img, sc


RS_FUSED = __import__("os").environ.get("HOPMI_RS_FUSED", "1") != "0"
# the LayerNorm operators also write the fp16 hi/lo image of what they hand to the next GEMM (A/B: HOPMI_IMG_FUSED=0)
IMG_FUSED = __import__("os").environ.get("HOPMI_IMG_FUSED", "1") != "0"
# BertIntermediate's product hands its output to BertOutput.dense as an operand image (and the backward likewise): A/B HOPMI_FFN_IMG=0
FFN_IMG = __import__("os").environ.get("HOPMI_FFN_IMG", "1") != "0"
# Operand IMAGES (LayerNorm / GEMM-epilogue / rows_image producers + the LDS-DMA GEMM form) pay from about 3 000 rows on: at M = 4352
# (TED, batch 128) the step gains 1.1 ms over row scales + the split form, at M = 2176 (TED-Expressive, batch 64) it LOSES 0.3 ms
# (13.24 vs 12.94 ms, A/B on one box: with half the row tiles the LDS-DMA form's workgroups no longer cover one another's staging
# latency, and the image stores are a fixed cost).  Below the threshold: row scales from the producing kernels and the split form.
IMG_MIN_ROWS = int(__import__("os").environ.get("HOPMI_IMG_MIN_ROWS", "3072"))
# An operand nobody wrote an image of (the attention's output and gradient, the embedding LayerNorm's output): ONE pass takes the row
# scales and writes the fp16 images (hopmi_rows_image_f16: 6.6 us at K = 768, 19 us at K = 2304, M = 4352) and the LDS-DMA form
# multiplies -- bit-identical to row scales + the split form.  Round 4 measured no gain in the step; with the tile-blocked images and
# 64-row tiles of round 5 it is 50-90 us per configs[1] step (kernel time 14.41 -> 14.36 ms, the step 14.81 -> 14.74 ms, A/B on one
# box), so it is on, for N up to the QKV width.  HOPMI_GEMM_AB=0: row scales + split form.
GEMM_AB = __import__("os").environ.get("HOPMI_GEMM_AB", "1") == "1"
GEMM_AB_MAX_N = int(__import__("os").environ.get("HOPMI_GEMM_AB_MAX_N", "2304"))


# the trainable linears (GRU input projections, align layer, beat MLP, reprogramming projections): one pass writes the operand's image
# AND its row scales (which the weight-gradient GEMM needs anyway), the LDS-DMA form multiplies -- also where K is not a multiple of
# 32 (GRU: K = 700, 2100): A/B HOPMI_F16_LINEAR_IMG=0 (row scales + split form)
F16_LINEAR_IMG = __import__("os").environ.get("HOPMI_F16_LINEAR_IMG", "1") != "0"
# (the generator's linears have larger N K per row than the BERT products the 3072-row threshold was measured on: from 2048 rows on,
# 14.34 -> 14.30 ms at configs[1] -- the beat MLP -- and 12.73 -> 12.70 at configs[3] -- the GRU input projections at M = 2176)
LINEAR_IMG_MIN_ROWS = int(__import__("os").environ.get("HOPMI_LINEAR_IMG_MIN_ROWS", "2048"))


def rows_image(a2d):
    """(image, scales) of a contiguous fp32 (M, K) operand, K even: hopmi_rows_image_f16 (tile-blocked fp16 hi / lo images with the
    columns padded to a multiple of 32, [2][M] row-scale pairs) -- the `a_img` of _split_gemm."""
    M, K = a2d.shape
    L = _lib.lib()
    img_a = torch.empty(L.hopmi_rows_image_f16_bytes(M, K), dtype=torch.uint8, device=a2d.device)
    sc = torch.empty(2, M, dtype=torch.float32, device=a2d.device)
    _lib.check(L.hopmi_rows_image_f16(a2d.data_ptr(), M, K, img_a.data_ptr(), sc.data_ptr(), _stream()), "hopmi_rows_image_f16")
    return img_a, sc


def _linear_operand(t, t2):
    """The fp16-form operand of a trainable linear: (row scales, image or None) of the 2-D view t2 of tensor t -- from the producer
    when it left them, else one pass."""
    M, K = t2.shape
    rs = _take_rs(t, M)
    im = _take_img(t, M, K) if K % 32 == 0 else None
    if im is not None:
        return im[1], im
    if rs is None and F16_LINEAR_IMG and M >= LINEAR_IMG_MIN_ROWS and K % 2 == 0 and K >= 128 and t2.is_contiguous() and t2.data_ptr() % 8 == 0:
        im = rows_image(t2)
        return im[1], im
    return (rs if rs is not None else row_scales(t2)), None


def _split_gemm(a2d, img, bias, N, K, parts, a_part=None, out=None, a_img=None):
    """`a_part` (fp16 form): the A operand's [2][M] row-scale pairs, or a ([P][M] partial row maxima, P) pair as a producing GEMM's
    `rowmax` left them; None: a hopmi_row_scales pass.  `a_img`: (image, scales) of a2d as `_take_img` returns them."""
    if parts == F16_PARTS:
        return _split_gemm_ep(a2d, img, bias, N, K, parts, 0, a_part=a_part, out=out, a_img=a_img)[0]
    if out is not None:
        raise _lib.HopmiError("hopmi _split_gemm: `out` is for the fp16 form")
    M = a2d.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=a2d.device)
    L = _lib.lib()
    _lib.check(_timed("gemm_split", 4 * (M * K + M * N) + 2 * parts * N * K, 2 * M * N * K,
                      lambda: L.hopmi_gemm_split(a2d.data_ptr(), img.data_ptr(), _ptr(bias), out.data_ptr(), M, N, K, parts, _stream())),
               "hopmi_gemm_split")
    return out


def split_rows_image(a2d, parts):
    """The part images [parts][M][K] (bf16) of a row-major fp32 matrix (hopmi_gemm_split_prepare): the A operand of
    hopmi_gemm_split_ab."""
    M, K = a2d.shape
    L = _lib.lib()
    img = torch.empty(L.hopmi_gemm_split_image_bytes(M, K, parts), dtype=torch.uint8, device=a2d.device)
    _lib.check(L.hopmi_gemm_split_prepare(a2d.data_ptr(), M, K, parts, img.data_ptr(), _stream()), "hopmi_gemm_split_prepare")
    return img


def _split_gemm_ab(a_img, M, img, bias, N, K, parts):
    out = torch.empty(M, N, dtype=torch.float32, device=img.device)
    L = _lib.lib()
    _lib.check(_timed("gemm_split", 2 * parts * (M + N) * K + 4 * M * N, 2 * M * N * K,
                      lambda: L.hopmi_gemm_split_ab(a_img.data_ptr(), img.data_ptr(), _ptr(bias), out.data_ptr(), M, N, K, parts, _stream())),
               "hopmi_gemm_split_ab")
    return out


class _SplitLinearFn(torch.autograd.Function):
    """y = x W^T (+ b) against a FROZEN weight: forward and activation gradient on hopmi_gemm_split; no weight gradient."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, img_w, img_wt, bias, N, K, parts, rowmax=None):
        x = _dev_f32(x, "x")
        rs = _take_rs(x, x.numel() // K) if parts == F16_PARTS else None
        ai = _take_img(x, x.numel() // K, K) if parts == F16_PARTS else None
        bd = None if bias is None else _dev_f32(bias.detach(), "bias")
        if rowmax is not None and parts == F16_PARTS:   # (the product's partial row maxima for its consumer: a list that receives them)
            y = _split_gemm_ep(x.reshape(-1, K), img_w, bd, N, K, parts, 0, a_part=rs, a_img=ai, rowmax=rowmax)[0]
        else:
            y = _split_gemm(x.reshape(-1, K), img_w, bd, N, K, parts, a_part=rs, a_img=ai)
        ctx.img_wt, ctx.N, ctx.K, ctx.parts = img_wt, N, K, parts
        return y.view(*x.shape[:-1], N)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        dy = _dev_f32(dy, "dy")
        rs = _take_rs(dy, dy.numel() // ctx.N) if ctx.parts == F16_PARTS else None
        ai = _take_img(dy, dy.numel() // ctx.N, ctx.N) if ctx.parts == F16_PARTS else None
        dx = _split_gemm(dy.reshape(-1, ctx.N), ctx.img_wt, None, ctx.K, ctx.N, ctx.parts, a_part=rs, a_img=ai)      # dX = dY . W = dY . (W^T)^T
        return dx.view(*dy.shape[:-1], ctx.K), None, None, None, None, None, None, None


def split_linear(x, img_w, img_wt, bias, N, K, parts, rowmax=None):
    return _SplitLinearFn.apply(x, img_w, img_wt, bias, N, K, parts, rowmax)


def _split_gemm_ep(a2d, img, bias, N, K, parts, epilogue, keep=False, aux=None, a_part=None, out=None, rowmax=None, a_img=None, rows=None):
    """hopmi_gemm_split_ep / hopmi_gemm_f16x2: epilogue 0 -> a2d W^T + bias; 1 -> (gelu(h), h if keep else None) with
    h = a2d W^T + bias; 2 -> (a2d W^T) * gelu'(aux).  `a_part`: the fp16 form's per-row scales of a2d (row_scales) when the caller has them."""
    M = a2d.shape[0] if rows is None else rows          # (`rows`: the operand only exists as its image, a2d is a shape-less stand-in)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a2d.device)
    elif tuple(out.shape) != (M, N) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != a2d.device:
        raise _lib.HopmiError(f"hopmi _split_gemm: `out` must be a contiguous float32 ({M}, {N}) tensor on {a2d.device}")
    h = torch.empty_like(out) if (epilogue == 1 and keep) else None
    L = _lib.lib()
    if parts == F16_PARTS and a_img is not None and K % 2 == 0:
        # the producer (a LayerNorm kernel, rows_image) wrote the operand's image: both operands by LDS-DMA, nothing split in the
        # k-loop.  (K % 32 != 0: both images carry zero columns up to the next multiple of 32 -- the kernel is told that K)
        img_a, sc = a_img
        K_log, K = K, (K + 31) // 32 * 32
        cm = torch.empty(L.hopmi_gemm_f16x2_tiles_n(N), M, dtype=torch.float32, device=a2d.device) if rowmax is not None else None
        _lib.check(_timed("gemm_split", 4 * (M * K_log + (3 if (h is not None or aux is not None) else 2) * M * N) + 4 * N * K_log, 2 * M * N * K_log,
                          lambda: L.hopmi_gemm_f16x2_ab_ep(img_a.data_ptr(), sc.data_ptr(), img.data_ptr(), _ptr(bias), out.data_ptr(), _ptr(h),
                                                           _ptr(aux), _ptr(cm), M, N, K, epilogue, _stream())),
                   "hopmi_gemm_f16x2_ab_ep")
        if rowmax is not None:
            rowmax.append((cm, cm.shape[0]))
        return out, h
    if parts == F16_PARTS:
        if (a_part is None and GEMM_AB and epilogue == 0 and N <= GEMM_AB_MAX_N and K % 32 == 0
                and M >= max(F16_LINEAR_MIN_ROWS, IMG_MIN_ROWS)):
            # no scales at hand: the pass that would take the row scales writes the operand's fp16 images as well, and the LDS-DMA form
            # multiplies (bit-identical to row scales + the split form)
            im = rows_image(a2d)
            _attach_rs(a2d, im[1])                     # (a later consumer of the same tensor object -- the weight-gradient GEMM -- takes the scales)
            return _split_gemm_ep(a2d, img, bias, N, K, parts, epilogue, keep=keep, aux=aux, out=out, rowmax=rowmax, a_img=im)
        if a_part is None:
            a_part = row_scales(a2d)
            _attach_rs(a2d, a_part)                    # (a later consumer of the same tensor object -- the weight-gradient GEMM -- takes them)
        a_t, a_p = a_part if isinstance(a_part, tuple) else (a_part, 0)
        if tuple(a_t.shape) != ((a_p, M) if a_p else (2, M)) or a_t.dtype != torch.float32 or not a_t.is_contiguous():
            raise _lib.HopmiError(f"hopmi _split_gemm: bad operand-scale tensor {tuple(a_t.shape)} for M = {M}, parts = {a_p}")
        # `rowmax` (a list): receives ([tiles_n][M] partial row maxima of the output, tiles_n) -- the `a_part` of the GEMM that consumes it
        cm = torch.empty(L.hopmi_gemm_f16x2_tiles_n(N), M, dtype=torch.float32, device=a2d.device) if rowmax is not None else None
        _lib.check(_timed("gemm_split", 4 * (M * K + (3 if (h is not None or aux is not None) else 2) * M * N) + 4 * N * K, 2 * M * N * K,
                          lambda: L.hopmi_gemm_f16x2(a2d.data_ptr(), a_t.data_ptr(), a_p, img.data_ptr(), _ptr(bias), out.data_ptr(), _ptr(h),
                                                     _ptr(aux), _ptr(cm), M, N, K, epilogue, _stream())),
                   "hopmi_gemm_f16x2")
        if rowmax is not None:
            rowmax.append((cm, cm.shape[0]))
        return out, h
    _lib.check(_timed("gemm_split", 4 * (M * K + (3 if (h is not None or aux is not None) else 2) * M * N) + 2 * parts * N * K, 2 * M * N * K,
                      lambda: L.hopmi_gemm_split_ep(a2d.data_ptr(), img.data_ptr(), _ptr(bias), out.data_ptr(), _ptr(h), _ptr(aux), M, N, K,
                                                    parts, epilogue, _stream())),
               "hopmi_gemm_split_ep")
    return out, h


def _gemm_ab_img(a_img, M, img, bias, N, K, epilogue, norms, mul, add, keep=False, aux=None):
    """hopmi_gemm_f16x2_ab_img: (image, scales) of the product (no fp32 output) and, epilogue 1 with `keep`, the pre-activation."""
    L = _lib.lib()
    img_a, sc = a_img
    dev = img_a.device
    h = torch.empty(M, N, dtype=torch.float32, device=dev) if (epilogue == 1 and keep) else None
    oi = torch.empty(L.hopmi_rows_image_f16_bytes(M, N), dtype=torch.uint8, device=dev)
    osc = torch.empty(2, M, dtype=torch.float32, device=dev)
    _lib.check(_timed("gemm_split", 4 * (M * K + (2 if (h is not None or aux is not None) else 1) * M * N) + 4 * N * K, 2 * M * N * K,
                      lambda: L.hopmi_gemm_f16x2_ab_img(img_a.data_ptr(), sc.data_ptr(), img.data_ptr(), _ptr(bias), None, _ptr(h), _ptr(aux), M, N, K,
                                                        epilogue, oi.data_ptr(), osc.data_ptr(), norms.data_ptr(), float(mul), float(add), _stream())),
               "hopmi_gemm_f16x2_ab_img")
    return (oi, osc), h


class _SplitFfnFn(torch.autograd.Function):
    """BertIntermediate + the dense of BertOutput against FROZEN weights: o = gelu(x W1^T + b1) W2^T (the bias of the second
    linear is added by the LayerNorm operator behind it).  The activation is the epilogue of the first product and its gradient
    the epilogue of the backward's first product (hopmi_gemm_split_ep): the values of split_linear -> bias_gelu -> split_linear
    bit for bit, without the two launches that re-read and re-write the M x 3072 tensor.
    Round 5, `bounds` = (largest row 2-norm of W1, largest |b1|, largest column 2-norm of W2) and an input that arrives with its
    operand image and row norms (a LayerNorm output / LayerNorm-backward dx): the first product hands its result to the second as an
    operand IMAGE (hopmi_gemm_f16x2_ab_img: no fp32 gelu output at all, its row scales from the Cauchy-Schwarz bound), and both run
    the LDS-DMA form; the backward likewise.  Same three-term arithmetic; the results differ from the split form's by the choice of
    the intermediate's row scale (a power of two) only where lo parts are subnormal."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, img1, img1t, b1, img2, img2t, N1, K, parts, bounds=None):
        x = _dev_f32(x, "x")
        M = x.numel() // K
        rs = _take_rs(x, M) if parts == F16_PARTS else None
        ai = _take_img(x, M, K) if parts == F16_PARTS else None
        nr = _take_norms(x, M) if (ai is not None and bounds is not None and FFN_IMG and N1 % 32 == 0) else None
        ctx.imgs, ctx.dims, ctx.bounds = (img1t, img2t), (N1, K, parts), bounds
        b1d = _dev_f32(b1.detach(), "bias")
        if nr is not None:
            fi, h = _gemm_ab_img(ai, M, img1, b1d, N1, K, 1, nr, bounds[0], bounds[1], keep=ctx.needs_input_grad[0])
            o = _split_gemm_ep(x.reshape(-1, K)[:, :0], img2, None, K, N1, parts, 0, a_img=fi, rows=M)[0]
        else:
            fmax = [] if (parts == F16_PARTS and RS_FUSED) else None           # the GELU epilogue leaves its output's row maxima
            f, h = _split_gemm_ep(x.reshape(-1, K), img1, b1d, N1, K, parts, 1, keep=ctx.needs_input_grad[0], a_part=rs, rowmax=fmax, a_img=ai)
            o = _split_gemm(f, img2, None, K, N1, parts, a_part=fmax[0] if fmax else None)
        ctx.save_for_backward(h)
        return o.view(*x.shape[:-1], K)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, do):
        (h,) = ctx.saved_tensors
        img1t, img2t = ctx.imgs
        N1, K, parts = ctx.dims
        do = _dev_f32(do, "do")
        M = do.numel() // K
        rs = _take_rs(do, M) if parts == F16_PARTS else None
        ai = _take_img(do, M, K) if parts == F16_PARTS else None
        nr = _take_norms(do, M) if (ai is not None and ctx.bounds is not None and FFN_IMG and N1 % 32 == 0) else None
        if nr is not None:
            # |(dO W2) gelu'(h)| <= ||dO_row|| max_n ||W2[:, n]|| max |gelu'| (1.1290)
            di, _ = _gemm_ab_img(ai, M, img2t, None, N1, K, 2, nr, 1.13 * ctx.bounds[2], 0.0, aux=h)
            dx = _split_gemm_ep(do.reshape(-1, K)[:, :0], img1t, None, K, N1, parts, 0, a_img=di, rows=M)[0]
        else:
            dmax = [] if (parts == F16_PARTS and RS_FUSED) else None
            dh, _ = _split_gemm_ep(do.reshape(-1, K), img2t, None, N1, K, parts, 2, aux=h, a_part=rs, rowmax=dmax, a_img=ai)   # (dO W2) * gelu'(h)
            dx = _split_gemm(dh, img1t, None, K, N1, parts, a_part=dmax[0] if dmax else None)                         # dH W1
        return dx.view(*do.shape), None, None, None, None, None, None, None, None, None


def split_ffn(x, img1, img1t, b1, img2, img2t, N1, K, parts, bounds=None):
    return _SplitFfnFn.apply(x, img1, img1t, b1, img2, img2t, N1, K, parts, bounds)


# ------------------------------------------------------- fused BERT epilogues (frozen LLM: no parameter grads)
class _BiasGeluFn(torch.autograd.Function):
    @staticmethod
    @_fwd32
    def forward(ctx, x, bias):
        x, bias = _dev_f32(x, "x"), _dev_f32(bias.detach(), "bias")
        N = x.shape[-1]
        M = x.numel() // N
        out = torch.empty_like(x)
        L, st = _lib.lib(), _stream()
        _lib.check(_timed("bias_gelu_fwd", 8 * x.numel(), 0,
                          lambda: L.hopmi_bias_gelu_fwd(x.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, st)),
                   "hopmi_bias_gelu_fwd")
        ctx.save_for_backward(x, bias)
        return out

    @staticmethod
    @_bwd32
    def backward(ctx, dy):
        x, bias = ctx.saved_tensors
        dy = _dev_f32(dy, "dy")
        N = x.shape[-1]
        M = x.numel() // N
        dx = torch.empty_like(x)
        L, st = _lib.lib(), _stream()
        _lib.check(_timed("bias_gelu_bwd", 12 * x.numel(), 0,
                          lambda: L.hopmi_bias_gelu_bwd(x.data_ptr(), bias.data_ptr(), dy.data_ptr(), dx.data_ptr(), M, N, st)),
                   "hopmi_bias_gelu_bwd")
        return dx, None


def bias_gelu(x, bias):
    """gelu_erf(x + bias); gradient w.r.t. x only (frozen-LLM epilogue)."""
    return _BiasGeluFn.apply(x, bias)


class _BiasDropResLnFn(torch.autograd.Function):
    @staticmethod
    @_fwd32
    def forward(ctx, x, bias, res, gamma, beta, eps, p_drop, seed):
        x, res = _dev_f32(x, "x"), _dev_f32(res, "res")
        bias, gamma, beta = (_dev_f32(t.detach(), n) for t, n in ((bias, "bias"), (gamma, "gamma"), (beta, "beta")))
        D = x.shape[-1]
        M = x.numel() // D
        res_rows = res.numel() // D
        need = x.requires_grad or res.requires_grad
        out = torch.empty_like(x)
        xhat = torch.empty_like(x) if need else None
        rstd = torch.empty(M, dtype=torch.float32, device=x.device) if need else None
        L, st, sp = _lib.lib(), _stream(), _seed_ptr()
        _lib.check(_timed("bias_drop_res_ln_fwd", 4 * x.numel() * (4 if need else 3), 0,
                          lambda: L.hopmi_bias_dropout_residual_layernorm_fwd(
                              x.data_ptr(), bias.data_ptr(), res.data_ptr(), res_rows, gamma.data_ptr(), beta.data_ptr(),
                              out.data_ptr(), _ptr(xhat), _ptr(rstd), M, D, float(eps), float(p_drop), int(seed) & _M32, sp, st)),
                   "hopmi_bias_dropout_residual_layernorm_fwd")
        if need:
            ctx.save_for_backward(xhat, rstd, gamma)
        ctx.p_drop, ctx.seed, ctx.res_shape, ctx.x_rows, ctx.sp = float(p_drop), int(seed) & _M32, res.shape, M, sp
        return out

    @staticmethod
    @_bwd32
    def backward(ctx, dout):
        xhat, rstd, gamma = ctx.saved_tensors
        dout = _dev_f32(dout, "dout")
        D = xhat.shape[-1]
        M = ctx.x_rows
        dx, dres = torch.empty_like(xhat), torch.empty_like(xhat)
        L, st = _lib.lib(), _stream()
        _lib.check(_timed("bias_drop_res_ln_bwd", 16 * xhat.numel(), 0,
                          lambda: L.hopmi_bias_dropout_residual_layernorm_bwd(
                              dout.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dx.data_ptr(),
                              dres.data_ptr(), M, D, ctx.p_drop, ctx.seed, ctx.sp, st)),
                   "hopmi_bias_dropout_residual_layernorm_bwd")
        if tuple(ctx.res_shape) != tuple(dres.shape):            # broadcast residual (e.g. position embeddings)
            dres = dres.view(-1, *ctx.res_shape).sum(0)
        return dx, None, dres, None, None, None, None, None


class _BiasDropResLn2Fn(torch.autograd.Function):
    """_BiasDropResLnFn handing its output out TWICE (two tensors on one storage): one for the next GEMM, one for the next
    residual add.  Autograd then delivers the two gradients separately and the backward kernel adds them while loading
    (`dout` + `dout_t` of hopmi_bias_dropout_residual_layernorm_bwd_dt, dtype 0) instead of a separate add launch over the
    activation for every residual junction (12 per step in a 6-layer encoder)."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, bias, res, gamma, beta, eps, p_drop, seed):
        x, res = _dev_f32(x, "x"), _dev_f32(res, "res")
        bias, gamma, beta = (_dev_f32(t.detach(), n) for t, n in ((bias, "bias"), (gamma, "gamma"), (beta, "beta")))
        D = x.shape[-1]
        M = x.numel() // D
        res_rows = res.numel() // D
        need = x.requires_grad or res.requires_grad
        out = torch.empty_like(x)
        xhat = torch.empty_like(x) if need else None
        rstd = torch.empty(M, dtype=torch.float32, device=x.device) if need else None
        L, st, sp = _lib.lib(), _stream(), _seed_ptr()
        # the fp16-form GEMMs behind this operator take their A operand's row scales from here (the rows are in registers)
        ctx.rs = RS_FUSED and GEMM_PARTS == F16_PARTS
        sc = torch.empty(2, M, dtype=torch.float32, device=x.device) if ctx.rs else None
        # ... and, where the LDS-DMA form of that GEMM pays (enough rows, whole 32-wide k-steps), the operand's image itself
        ctx.im = bool(ctx.rs and IMG_FUSED and D % 32 == 0 and M >= IMG_MIN_ROWS)
        im = torch.empty(L.hopmi_rows_image_f16_bytes(M, D), dtype=torch.uint8, device=x.device) if ctx.im else None
        nr = torch.empty(M, dtype=torch.float32, device=x.device) if ctx.im else None
        _lib.check(_timed("bias_drop_res_ln_fwd", 4 * x.numel() * (4 if need else 3), 0,
                          lambda: L.hopmi_bias_dropout_residual_layernorm_fwd_im(
                              x.data_ptr(), bias.data_ptr(), res.data_ptr(), res_rows, gamma.data_ptr(), beta.data_ptr(),
                              out.data_ptr(), None, _ptr(xhat), _ptr(rstd), _ptr(sc), _ptr(im), _ptr(nr), M, D, float(eps), float(p_drop),
                              int(seed) & _M32, sp, 0, st)),
                   "hopmi_bias_dropout_residual_layernorm_fwd")
        if need:
            ctx.save_for_backward(xhat, rstd, gamma)
        ctx.p_drop, ctx.seed, ctx.res_shape, ctx.x_rows, ctx.sp = float(p_drop), int(seed) & _M32, res.shape, M, sp
        ctx.set_materialize_grads(False)
        if sc is not None:
            ctx.mark_non_differentiable(sc)
        if im is not None:
            ctx.mark_non_differentiable(im, nr)
        return out, out.detach(), sc, im, nr

    @staticmethod
    @_bwd32
    def backward(ctx, dout, dout2, _dsc=None, _dim=None, _dnr=None):
        xhat, rstd, gamma = ctx.saved_tensors
        if dout is None:
            dout, dout2 = dout2, None
        if dout is None:
            return (None,) * 8
        dout = _dev_f32(dout, "dout")
        d2 = None if dout2 is None else _dev_f32(dout2, "dout2")
        D = xhat.shape[-1]
        M = ctx.x_rows
        dx, dres = torch.empty_like(xhat), torch.empty_like(xhat)
        L, st = _lib.lib(), _stream()
        sc = torch.empty(2, M, dtype=torch.float32, device=dx.device) if ctx.rs else None      # dx feeds the backward's next GEMM
        im = torch.empty(L.hopmi_rows_image_f16_bytes(M, D), dtype=torch.uint8, device=dx.device) if ctx.im else None
        nr = torch.empty(M, dtype=torch.float32, device=dx.device) if ctx.im else None
        _lib.check(_timed("bias_drop_res_ln_bwd", (16 if d2 is None else 20) * xhat.numel(), 0,
                          lambda: L.hopmi_bias_dropout_residual_layernorm_bwd_im(
                              dout.data_ptr(), _ptr(d2), xhat.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dx.data_ptr(),
                              dres.data_ptr(), _ptr(sc), _ptr(im), _ptr(nr), M, D, ctx.p_drop, ctx.seed, ctx.sp, 0, st)),
                   "hopmi_bias_dropout_residual_layernorm_bwd_dt")
        if im is not None:
            _attach_img(dx, im, sc, nr)
        elif sc is not None:
            _attach_rs(dx, sc)
        if tuple(ctx.res_shape) != tuple(dres.shape):            # broadcast residual (e.g. position embeddings)
            dres = dres.view(-1, *ctx.res_shape).sum(0)
        return dx, None, dres, None, None, None, None, None


def bias_dropout_residual_layernorm2(x, bias, res, gamma, beta, eps, p_drop=0.0, seed=0):
    """(out, out again on the same storage): see _BiasDropResLn2Fn."""
    out, out2, sc, im, nr = _BiasDropResLn2Fn.apply(x, bias, res, gamma, beta, eps, p_drop, seed)
    if im is not None:
        _attach_img(out, im, sc, nr)
    elif sc is not None:
        _attach_rs(out, sc)
    return out, out2


# ---- bf16-storage forms of the frozen BERT's epilogue / attention operators (dtype argument of the `_dt` entry points) ------
# Under bf16 autocast the library GEMMs produce and consume bf16; these operators read the GEMM output and write the next
# GEMM's input in bf16 themselves (fp32 arithmetic inside), so no cast kernel sits on either side of them.  The values are
# the ones the fp32-storage operators + autocast's casts produce: the rounding to bf16 just moves into the store.
_BF16 = 1


def _dev_bf16(t, name):
    if not t.is_cuda:
        raise _lib.HopmiError(f"hopmi: `{name}` is on {t.device}; the hot path only runs on a ROCm device (no CPU fallback)")
    if t.dtype != torch.bfloat16:
        raise _lib.HopmiError(f"hopmi: `{name}` must be bfloat16, got {t.dtype}")
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def colsum(x: torch.Tensor) -> torch.Tensor:
    """Sum over every dimension but the last: x (..., N), fp32 or bf16 -> (N,) fp32 (hopmi_colsum: the bias gradient of a
    linear layer at HBM speed, fixed summation order)."""
    N = x.shape[-1]
    x2 = x.reshape(-1, N)
    typed = x2.dtype == torch.bfloat16
    x2 = _dev_bf16(x2, "x") if typed else _dev_f32(x2.float(), "x")
    M = x2.shape[0]
    L, st = _lib.lib(), _stream()
    out = torch.empty(N, dtype=torch.float32, device=x2.device)
    nws = L.hopmi_colsum_ws_floats(M, N)
    ws = torch.empty(nws, dtype=torch.float32, device=x2.device) if nws else None
    _lib.check(_timed("colsum", x2.numel() * x2.element_size(), x2.numel(),
                      lambda: L.hopmi_colsum(x2.data_ptr(), 1 if typed else 0, M, N, out.data_ptr(), _ptr(ws), st)), "hopmi_colsum")
    return out


_MM_F32_OUT = None            # torch.mm(bf16, bf16, out_dtype=float32) available on this device?


def _mm_f32(a, b):
    """a @ b for low-precision a, b with the product accumulated AND returned in fp32 (a weight gradient goes into an fp32
    .grad: no rounding to bf16 in between, no cast launch); falls back to a cast where the library has no such GEMM."""
    global _MM_F32_OUT
    if _MM_F32_OUT is None and __import__("os").environ.get("HOPMI_MM_F32", "0") != "1":
        _MM_F32_OUT = False       # default: bf16-out GEMM + cast (measured 0.9 % faster on the bf16 step: the library's fp32-out
                                  # kernels for these shapes are not in the tuned table); HOPMI_MM_F32=1 for the unrounded gradient
    if _MM_F32_OUT is None:
        try:
            torch.mm(a[:1], b[:, :1], out_dtype=torch.float32)
            _MM_F32_OUT = True
        except (RuntimeError, TypeError):
            _MM_F32_OUT = False
    return torch.mm(a, b, out_dtype=torch.float32) if _MM_F32_OUT else (a @ b).float()


_CAST_CACHE = {}    # id(parameter) -> (weak reference, version, dtype, cast copy)


def cast_cache_reset():
    """What graph.GraphedTrainStep calls right before it starts recording a step and right after (and after a failed recording): a
    copy made by an eager call lives in the eager allocator pool, and a recording that was served that copy would bake its address
    in without recording the cast (stale or freed memory on every replay); a copy made while recording lives in the graph's pool.
    = reset_all_caches("recording")."""
    reset_all_caches("recording")


def invalidate_weight_images():
    """Forget every cached operand derived from a weight or attached to a tensor: bf16 casts, fp16 hi/lo images of trainable AND
    frozen owners, the frozen BERT encoder's fused QKV weight / images / FFN norm bounds / bf16 copies, row scales and images
    producers attached to their outputs.  The caches are keyed on identity + version counter (+ CACHE_EPOCH), so an update that
    bypasses the counter (`p.data.copy_`, `p.data.mul_`, an EMA written through `.data`) leaves stale operands behind -- for the
    FFN bounds a too-small bound, i.e. overflowing fp16 images -- until the next optimizer step: call this after any such write.
    (`load_state_dict`, optimizer steps and in-place ops on the parameter itself move the counter and need nothing.)
    = reset_all_caches("all")."""
    reset_all_caches("all")


def _cast_param(w, dt):
    """`w.to(dt)` for an operand of `_LinearFn` under autocast.  A step runs every trainable linear in two forwards (the graded one
    and the no-grad one of the diversity regulariser) between two optimizer steps: for an nn.Parameter -- an object that outlives
    the call, so its identity + version counter say whether the copy is current -- the second forward reuses the first one's
    copy.  Anything else (a temporary such as a packed weight: ids and versions of temporaries repeat) is cast on the spot."""
    if not CAST_CACHE_ENABLED or not isinstance(w, torch.nn.Parameter):
        return w.to(dt)
    cap = w.is_cuda and torch.cuda.is_current_stream_capturing()
    hit = _CAST_CACHE.get(id(w))
    # (a hit must come from the same side of a recording as the lookup: entry[4] says whether the copy was made under capture)
    if hit is not None and hit[0]() is w and hit[1] == w._version and hit[2] == dt and hit[4] == cap:
        if _checking():
            _check_equal("cast of a parameter (_cast_param)", hit[3], w.detach().to(dt))
        return hit[3]
    c = w.detach().to(dt)
    _CAST_CACHE[id(w)] = (weakref.ref(w), w._version, dt, c, cap)
    return c


class _LinearFn(torch.autograd.Function):
    """torch.nn.functional.linear whose backward takes the bias gradient with hopmi_colsum (the library's column reduction
    is the slowest piece of a trainable linear layer's backward here).  Under autocast the operands are cast as
    F.linear's autocast rule does and both gradient GEMMs run in that type; the results are handed back in the
    parameters' types (HOPMI_MM_F32=1: the weight gradient straight from the GEMM's fp32 accumulators, no rounding to bf16 and
    no cast launch -- 0.9 % slower on the bf16 step with the library's default kernel selection, so not the default).
    (Round 3: a cast of an nn.Parameter operand is reused by the step's second forward, `_cast_param` below: -0.04 ms on the bf16
    step.  Tried and dropped: persistent bf16 shadows of all parameters, re-cast in one multi-tensor copy per step -- no measurable
    gain, and a cache keyed on storage is wrong for temporaries such as gwnet's packed skip weights, whose address the allocator
    recycles.)"""

    @staticmethod
    def forward(ctx, x, w, b):
        amp = x.is_cuda and torch.is_autocast_enabled("cuda")
        xc, wc, bc = x, w, b
        if amp:
            dt = torch.get_autocast_dtype("cuda")
            xc = x if x.dtype == dt else x.to(dt)
            wc = w if w.dtype == dt else _cast_param(w, dt)
            bc = b if (b is None or b.dtype == dt) else _cast_param(b, dt)
        with torch.autocast("cuda", enabled=False):
            y = torch.nn.functional.linear(xc.reshape(-1, xc.shape[-1]), wc, bc)
        ctx.save_for_backward(xc, wc)
        ctx.types = (x.dtype, w.dtype, b.dtype if b is not None else None)
        # (not a tracked view: an in-place activation may follow, e.g. HOP.py:131's LeakyReLU(inplace=True))
        return torch.ops.aten._unsafe_view(y, list(x.shape[:-1]) + [w.shape[0]])

    @staticmethod
    def backward(ctx, dy):
        xc, wc = ctx.saved_tensors
        tx, tw, tb = ctx.types
        with torch.autocast("cuda", enabled=False):
            dy2 = dy.reshape(-1, dy.shape[-1])
            if dy2.dtype != wc.dtype:
                dy2 = dy2.to(wc.dtype)
            dx = dw = db = None
            if ctx.needs_input_grad[0]:
                dx = (dy2 @ wc).view(xc.shape)
                dx = dx if dx.dtype == tx else dx.to(tx)
            if ctx.needs_input_grad[1]:
                x2 = xc.reshape(-1, xc.shape[-1])
                if tw == torch.float32 and dy2.dtype != torch.float32:
                    dw = _mm_f32(dy2.t(), x2)
                else:
                    dw = dy2.t() @ x2
                    dw = dw if dw.dtype == tw else dw.to(tw)
            if tb is not None and ctx.needs_input_grad[2]:
                db = colsum(dy2)
                db = db if db.dtype == tb else db.to(tb)
        return dx, dw, db


# ---- trainable linears on the fp16 hi/lo GEMM form (fp32 mode) --------------------------------------------------------------
# The generator's own nn.Linear layers and GRU input projections (HOP.py:118,130-134,166-167,259-265) are fp32 library GEMMs at
# 100-130 TFLOP/s; hopmi_gemm_f16x2 computes the same products fp32-equivalently at ~2 x that.  Forward y = x W^T + b and the
# activation gradient dX = dY W go through it; the weight gradient dW = dY^T x (both operands activations, contraction over the
# rows) stays with the library.  The weights are trainable: their images (W for the forward, W^T for dX, one launch each) are
# rebuilt when the weight's version counter moves, i.e. once per optimizer step, and shared by the step's forwards.
F16_LINEAR = __import__("os").environ.get("HOPMI_F16_LINEAR", "1") != "0"
# From M N K = 5e9 on (tools/bench_linear.py, against the TUNED library kernels: GRU input projections 120 -> 80 us + 9 us of row
# scales, align layer 96 -> 60, beat MLP 214 -> 134; below, the row-scales pass and the weight images eat the gain)
F16_LINEAR_MIN_MNK = float(__import__("os").environ.get("HOPMI_F16_LINEAR_MIN_MNK", "3.0e9"))
_F16_IMG = {}                       # (ids of the owner parameters, N, K, transpose) -> (weak references, versions, image, made under capture)
_F16_IMG_FROZEN = {}                # the same for owners that take no gradient (never reset by a recording)


def f16_weight_image(w, transpose=False, owners=None):
    """The hopmi_gemm_f16x2 image of the row-major (N, K) weight `w` (transpose: of w^T, for dX = dY w).  `owners`: the
    nn.Parameter objects `w` is (a view or packed alias of): the image is cached under their identities and version counters (a
    hit must also come from the same side of a recording as the lookup, as in `_cast_param`); without owners nothing is cached
    (addresses and version counters of temporaries repeat)."""
    N, K = w.shape
    L = _lib.lib()

    def build():
        wd = _dev_f32(w.detach().t().contiguous() if transpose else w.detach(), "weight")
        n, k = wd.shape
        img = torch.empty(L.hopmi_gemm_f16x2_image_bytes(n, k), dtype=torch.uint8, device=w.device)
        _lib.check(L.hopmi_gemm_f16x2_prepare(wd.data_ptr(), n, k, img.data_ptr(), _stream()), "hopmi_gemm_f16x2_prepare")
        return img

    if not owners:
        return build()
    key = tuple(id(o) for o in owners) + (N, K, bool(transpose))
    vers = tuple(o._version for o in owners) + (w.data_ptr(), CACHE_EPOCH)
    # FROZEN owners (no gradient): the image is static -- one table that no recording resets (a recording may hold the address of an
    # image made by an eager call: the table keeps it alive until the weight itself changes, and GraphedTrainStep drops its
    # recordings when a frozen parameter's version moves)
    frozen = not any(o.requires_grad for o in owners)
    table = _F16_IMG_FROZEN if frozen else _F16_IMG
    cap = w.is_cuda and torch.cuda.is_current_stream_capturing()
    hit = table.get(key)
    # (a frozen image made by an EAGER call serves both sides of a recording; one made under capture only the recording)
    if hit is not None and all(r() is o for r, o in zip(hit[0], owners)) and hit[1] == vers and (hit[3] == cap or (frozen and not hit[3])):
        if _checking():
            _check_equal("fp16 image of a weight (f16_weight_image)", hit[2], build())
        return hit[2]
    img = build()
    if not frozen and len(table) > 256:
        table.clear()
    table[key] = (tuple(weakref.ref(o) for o in owners), vers, img, cap)
    return img


def f16_linear_ok(x, w, b, owners=None) -> bool:
    """Does `linear` send this product to hopmi_gemm_f16x2?  A weight with known owner parameters (its images are cached per optimizer
    step), fp32 operands on the device, no autocast, K % 4 == 0, and enough work for the form to beat the tuned library GEMM."""
    return bool(F16_LINEAR and owners and GEMM_PARTS == F16_PARTS and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32
                and w.dim() == 2 and (b is None or b.dtype == torch.float32) and not torch.is_autocast_enabled("cuda")
                and w.shape[1] % 4 == 0 and w.shape[0] >= 128 and w.is_contiguous() and w.data_ptr() % 16 == 0
                and float(x.numel()) * w.shape[0] >= F16_LINEAR_MIN_MNK)


def f16_mm_nt_ok(a2d, w, owners) -> bool:
    return bool(F16_LINEAR and owners and GEMM_PARTS == F16_PARTS and a2d.is_cuda and a2d.dtype == torch.float32 and w.dtype == torch.float32
                and a2d.dim() == 2 and w.dim() == 2 and not torch.is_autocast_enabled("cuda") and w.shape[1] % 4 == 0 and w.is_contiguous()
                and w.data_ptr() % 16 == 0 and a2d.shape[0] >= F16_LINEAR_MIN_ROWS and float(a2d.numel()) * w.shape[0] >= F16_LINEAR_MIN_MNK)


def f16_mm_nt(a2d, w, owners, out=None):
    """a2d (M, K) @ w (N, K)^T on hopmi_gemm_f16x2 (no autograd), the image of `w` cached under `owners` -- e.g. the mapping
    layer's weight gradient dW = dS E^T against the FROZEN word embeddings E (30522 x 768: E itself is the row-major Bt operand)."""
    N, K = w.shape
    a2d = _dev_f32(a2d.detach(), "a")
    # (the split form: with N = 30522 column tiles the LDS-DMA form's 64-row tiles re-read the weight image once more per row tile --
    # image pass + LDS-DMA form measured 0.16 ms per step slower here, A/B twice on one box)
    return _split_gemm(a2d, f16_weight_image(w, owners=tuple(owners)), None, N, K, F16_PARTS, a_part=_take_rs(a2d, a2d.shape[0]), out=out)


F16_LINEAR_MIN_ROWS = 1024          # below: the 128-row tiles leave the chip empty


# The mapping layer's forward S = W_map E + b[:, None] (HOP.py:200; 1500 x 768 outputs, K = vocab = 30522) on the LDS-DMA form with
# the contraction split into slabs (hopmi_gemm_f16x2_ab_splitk) instead of the library's strided-batched fp32 GEMM.  Built, parity-
# tested (test_mapping_forward_split_k_vs_float64) and OFF by default: back to back 523 us against the library path's 605
# (tools/probes/bench_mapfwd.py: image of W 189, product + slab sum 228), kernel time per step 14.36 -> 14.24 ms under the profiler --
# but the RECORDED step is slower with it, reproducibly (14.84 / 14.87 / 14.93 -> 15.47 / 15.39 / 15.39 ms, A/B on two boxes; the
# eager step does not move): W's image is 183 MB written and read back inside every step, and the idle time between the step's
# kernels grows by more than the kernels shrink.  Not understood further; HOPMI_F16_SPLITK=1 enables it.
F16_SPLITK = __import__("os").environ.get("HOPMI_F16_SPLITK", "0") == "1"


def f16_affine_splitk_ok(W, E) -> bool:
    """W (M, K) trainable, E (K, N) frozen: does W @ E + b[:, None] go to hopmi_gemm_f16x2_ab_splitk?"""
    return bool(F16_SPLITK and F16_LINEAR and GEMM_PARTS == F16_PARTS and W.is_cuda and W.dtype == torch.float32 and E.dtype == torch.float32
                and W.dim() == 2 and E.dim() == 2 and not E.requires_grad and not torch.is_autocast_enabled("cuda")
                and W.shape[1] % 2 == 0 and E.shape[1] % 4 == 0 and W.stride(1) == 1 and W.stride(0) == W.shape[1]
                and W.data_ptr() % 8 == 0 and float(W.numel()) * E.shape[1] >= 4 * F16_LINEAR_MIN_MNK and W.shape[1] >= 4096)


def f16_affine_splitk(W, E, b):
    """W (M, K) @ E (K, N) + b[:, None] (no autograd): one pass writes the fp16 hi/lo image of W's rows (W changes every optimizer
    step), E^T's image is cached under the frozen E, the split-K LDS-DMA form multiplies."""
    M, K = W.shape
    N = E.shape[1]
    L = _lib.lib()
    Wd = _dev_f32(W.detach(), "W")
    img_b = f16_weight_image(E, transpose=True, owners=(E,))
    img_a = torch.empty(L.hopmi_rows_image_f16_bytes(M, K), dtype=torch.uint8, device=W.device)
    sc = torch.empty(2, M, dtype=torch.float32, device=W.device)
    _lib.check(L.hopmi_rows_image_f16(Wd.data_ptr(), M, K, img_a.data_ptr(), sc.data_ptr(), _stream()), "hopmi_rows_image_f16")
    ws = torch.empty(L.hopmi_gemm_f16x2_ab_splitk_ws_floats(M, N, K), dtype=torch.float32, device=W.device)
    out = torch.empty(M, N, dtype=torch.float32, device=W.device)
    bd = None if b is None else _dev_f32(b.detach(), "b")
    _lib.check(_timed("gemm_split", 4 * (M * K + N * K + M * N), 2 * M * N * K,
                      lambda: L.hopmi_gemm_f16x2_ab_splitk(img_a.data_ptr(), sc.data_ptr(), img_b.data_ptr(), _ptr(bd), out.data_ptr(), M, N, K,
                                                           ws.data_ptr(), _stream())), "hopmi_gemm_f16x2_ab_splitk")
    return out

# dW = dY^T X on hopmi_gemm_f16x2_tn (csrc/gemm_tn.hip) instead of the library's fp32 GEMM: from M N K = 1e9 on (below, the row-scale
# passes and the slab sum eat the gain; A/B at configs[1]: 2e9 14.38, 1e9 14.34, 5e8 14.35 ms per step)
F16_TN = __import__("os").environ.get("HOPMI_F16_TN", "1") != "0"
TN_COLSUM = __import__("os").environ.get("HOPMI_TN_COLSUM", "1") != "0"      # bias gradients as a by-product of the TN weight gradient
F16_TN_MIN_MNK = float(__import__("os").environ.get("HOPMI_F16_TN_MIN_MNK", "1.0e9"))
_UNIT_RS = {}                       # (device, M) -> [2][M] row-scale pairs {2^14, 2^-14}: operands bounded by 1 (GRU states)


def unit_row_scales(M, device):
    """The row-scale pairs of an operand whose magnitudes are bounded by 1 (no pass over it): s = 2^14 for every row."""
    key = (str(device), M)
    t = _UNIT_RS.get(key)
    if t is None:
        t = torch.empty(2, M, dtype=torch.float32, device=device)
        t[0].fill_(16384.0)
        t[1].fill_(1.0 / 16384.0)
        # (a constant made by an eager call serves recordings too: this table keeps it alive; one first made under capture lives
        # in the graph's pool and is filled by recorded launches -- not kept)
        if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):
            _UNIT_RS[key] = t
    return t


def f16_mm_tn_ok(a, b) -> bool:
    """Does the weight gradient a^T b (a: (.., M, N), b: (.., M, K), rows strided or not) go to hopmi_gemm_f16x2_tn?"""
    return bool(F16_TN and GEMM_PARTS == F16_PARTS and a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32
                and not torch.is_autocast_enabled("cuda") and a.stride(-1) == 1 and b.stride(-1) == 1
                and float(a.shape[-2]) * a.shape[-1] * b.shape[-1] * (a.shape[0] if a.dim() == 3 else 1)
                >= F16_TN_MIN_MNK * (1.0 if a.shape[-2] >= IMG_MIN_ROWS else 2.0))       # (short contractions, e.g. M = 2176: from 2e9 on)


def f16_mm_tn(a, b, a_rs, b_rs, out=None, accumulate=False, colsum=False):
    """a^T b on hopmi_gemm_f16x2_tn (no autograd): a (M, N) and b (M, K) fp32, unit stride along the last axis, any row stride; or
    (batch, M, N) / (batch, M, K) with one batch stride each.  a_rs / b_rs: the operands' [2][M] row-scale pairs (row_scales,
    _take_rs, unit_row_scales; shared by the batch members).  Returns (N, K) or (batch, N, K); with `colsum` (unbatched) the pair
    (a^T b, column sums of a) -- with a = dY the weight AND the bias gradient of a linear from one pass over dY."""
    batched = a.dim() == 3
    bt = a.shape[0] if batched else 1
    M, N = a.shape[-2:]
    K = b.shape[-1]
    if b.shape[-2] != M or (batched and b.shape[0] != bt) or a.stride(-1) != 1 or b.stride(-1) != 1:
        raise _lib.HopmiError(f"hopmi f16_mm_tn: bad operands {tuple(a.shape)} / {tuple(b.shape)}")
    for t, nm in ((a_rs, "a_rs"), (b_rs, "b_rs")):
        if tuple(t.shape) != (2, M) or t.dtype != torch.float32 or not t.is_contiguous():
            raise _lib.HopmiError(f"hopmi f16_mm_tn: {nm} must be a contiguous float32 (2, {M}) tensor")
    if out is None:
        out = torch.empty((bt, N, K) if batched else (N, K), dtype=torch.float32, device=a.device)
        accumulate = False
    elif out.dtype != torch.float32 or out.stride(-1) != 1 or tuple(out.shape) != ((bt, N, K) if batched else (N, K)):
        raise _lib.HopmiError("hopmi f16_mm_tn: bad `out`")
    L = _lib.lib()
    nws = L.hopmi_gemm_f16x2_tn_ws_floats(M, N, K, bt)
    ws = torch.empty(nws, dtype=torch.float32, device=a.device) if nws else None
    if colsum and batched:
        raise _lib.HopmiError("hopmi f16_mm_tn: `colsum` is for unbatched operands")
    cs = torch.empty(N, dtype=torch.float32, device=a.device) if colsum else None
    _lib.check(_timed("gemm_tn", 4 * bt * (M * N + M * K + N * K), 2 * bt * M * N * K,
                      lambda: L.hopmi_gemm_f16x2_tn_cs(a.data_ptr(), a.stride(-2), a.stride(0) if batched else 0, a_rs.data_ptr(),
                                                       b.data_ptr(), b.stride(-2), b.stride(0) if batched else 0, b_rs.data_ptr(),
                                                       out.data_ptr(), out.stride(-2), out.stride(0) if batched else 0, _ptr(ws), M, N, K, bt,
                                                       1 if accumulate else 0, _ptr(cs), _stream())),
               "hopmi_gemm_f16x2_tn")
    return (out, cs) if colsum else out



class _F16LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, owners):
        N, K = w.shape
        x2 = _dev_f32(x.detach(), "x").reshape(-1, K)
        xs, xi = _linear_operand(x, x2)                # (the scales are kept: the weight gradient takes its operand scale from them)
        y = _split_gemm(x2, f16_weight_image(w, owners=owners), None if b is None else _dev_f32(b.detach(), "bias"), N, K, F16_PARTS,
                        a_part=xs, a_img=xi)
        ctx.save_for_backward(x2, w, xs)
        ctx.x_shape, ctx.has_b, ctx.owners = x.shape, b is not None, owners
        return torch.ops.aten._unsafe_view(y, list(x.shape[:-1]) + [N])

    @staticmethod
    def backward(ctx, dy):
        x2, w, xs = ctx.saved_tensors
        N, K = w.shape
        dy2 = _dev_f32(dy, "dy").reshape(-1, N)
        dx = dw = db = None
        tn = ctx.needs_input_grad[1] and f16_mm_tn_ok(dy2, x2)
        ds, di = _take_rs(dy, dy2.shape[0]), None
        dx_f16 = ctx.needs_input_grad[0] and N % 4 == 0 and K >= 128       # (dX contracts over N: the kernel's K % 4 == 0 rule applies to it here)
        if dx_f16:
            ds, di = _linear_operand(dy, dy2)          # one pass (scales + image) serves both gradient products
        elif tn and ds is None and N % 4 == 0:
            # only the weight gradient wants dY (a first layer, or dX with K < 128 on the library): its row scales, no image
            ds = row_scales(dy2)
            _attach_rs(dy2, ds)
        if ctx.needs_input_grad[0]:
            if dx_f16:
                dx = _split_gemm(dy2, f16_weight_image(w, transpose=True, owners=ctx.owners), None, K, N, F16_PARTS, a_part=ds, a_img=di)
            else:
                dx = dy2 @ w
            dx = dx.view(ctx.x_shape)
        want_db = ctx.has_b and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if tn and ds is not None and want_db and TN_COLSUM:
                dw, db = f16_mm_tn(dy2, x2, ds, xs, colsum=True)      # (the bias gradient rides on the weight gradient's pass over dY)
            else:
                dw = f16_mm_tn(dy2, x2, ds, xs) if (tn and ds is not None) else dy2.t() @ x2
        if want_db and db is None:
            db = colsum(dy2)
        return dx, dw, db, None


def linear(x, w, b=None, owners=None):
    """F.linear(x, w, b).  `owners` = the nn.Parameter objects behind `w` (the weight itself for an nn.Linear; both directions'
    parameters for a packed GRU weight): with them, in fp32 mode and with enough work, the product runs on hopmi_gemm_f16x2
    (`_F16LinearFn`, fp32-equivalent); otherwise the library, with a bias that needs a gradient through _LinearFn (column-sum
    kernel)."""
    if f16_linear_ok(x, w, b, owners):
        owners = tuple(owners)
        if not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (b is not None and b.requires_grad))):
            N, K = w.shape
            x2 = _dev_f32(x.detach(), "x").reshape(-1, K)
            xs, xi = _linear_operand(x, x2)
            y = _split_gemm(x2, f16_weight_image(w, owners=owners), None if b is None else _dev_f32(b.detach(), "bias"), N, K, F16_PARTS,
                            a_part=xs, a_img=xi)
            return y.view(*x.shape[:-1], N)
        return _F16LinearFn.apply(x, w, b, owners)
    if b is None or not (torch.is_grad_enabled() and b.requires_grad):
        return torch.nn.functional.linear(x, w, b)
    return _LinearFn.apply(x, w, b)


class Linear(torch.nn.Linear):
    """torch.nn.Linear (same parameters, same state_dict keys) whose forward is `linear` above."""

    def forward(self, x):
        return linear(x, self.weight, self.bias, owners=(self.weight,))


class _BiasGeluBf16Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias):
        x, bias = _dev_bf16(x, "x"), _dev_f32(bias.detach().float(), "bias")
        N = x.shape[-1]
        M = x.numel() // N
        out = torch.empty_like(x)
        L, st = _lib.lib(), _stream()
        _lib.check(_timed("bias_gelu_fwd", 4 * x.numel(), 0,
                          lambda: L.hopmi_bias_gelu_fwd_dt(x.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, _BF16, st)),
                   "hopmi_bias_gelu_fwd_dt")
        ctx.save_for_backward(x, bias)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, bias = ctx.saved_tensors
        dy = _dev_bf16(dy, "dy")
        N = x.shape[-1]
        M = x.numel() // N
        dx = torch.empty_like(x)
        L, st = _lib.lib(), _stream()
        _lib.check(_timed("bias_gelu_bwd", 6 * x.numel(), 0,
                          lambda: L.hopmi_bias_gelu_bwd_dt(x.data_ptr(), bias.data_ptr(), dy.data_ptr(), dx.data_ptr(), M, N, _BF16, st)),
                   "hopmi_bias_gelu_bwd_dt")
        return dx, None


def bias_gelu_bf16(x, bias):
    """gelu_erf(x + bias) with bf16 storage on both sides (x: a bf16 GEMM output; result: the next GEMM's bf16 input)."""
    return _BiasGeluBf16Fn.apply(x, bias)


class _BiasDropResLnBf16Fn(torch.autograd.Function):
    """(out fp32, out16 bf16) = LayerNorm(dropout(x + bias) + res): x is a bf16 GEMM output, res / out the fp32 residual
    stream, out16 the same values in bf16 for the next GEMM.  Backward takes both incoming gradients."""

    @staticmethod
    def forward(ctx, x, bias, res, gamma, beta, eps, p_drop, seed):
        x, res = _dev_bf16(x, "x"), _dev_f32(res, "res")
        bias, gamma, beta = (_dev_f32(t.detach().float(), n) for t, n in ((bias, "bias"), (gamma, "gamma"), (beta, "beta")))
        D = x.shape[-1]
        M = x.numel() // D
        res_rows = res.numel() // D
        need = x.requires_grad or res.requires_grad
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        out16 = torch.empty_like(x)
        xhat = torch.empty_like(out) if need else None
        rstd = torch.empty(M, dtype=torch.float32, device=x.device) if need else None
        L, st, sp = _lib.lib(), _stream(), _seed_ptr()
        _lib.check(_timed("bias_drop_res_ln_fwd", x.numel() * (2 + 4 + 4 + 2 + (4 if need else 0)), 0,
                          lambda: L.hopmi_bias_dropout_residual_layernorm_fwd_dt(
                              x.data_ptr(), bias.data_ptr(), res.data_ptr(), res_rows, gamma.data_ptr(), beta.data_ptr(),
                              out.data_ptr(), out16.data_ptr(), _ptr(xhat), _ptr(rstd), M, D, float(eps), float(p_drop),
                              int(seed) & _M32, sp, _BF16, st)),
                   "hopmi_bias_dropout_residual_layernorm_fwd_dt")
        if need:
            ctx.save_for_backward(xhat, rstd, gamma)
        ctx.p_drop, ctx.seed, ctx.res_shape, ctx.x_rows, ctx.sp = float(p_drop), int(seed) & _M32, res.shape, M, sp
        return out, out16

    @staticmethod
    def backward(ctx, dout, dout16):
        xhat, rstd, gamma = ctx.saved_tensors
        D = xhat.shape[-1]
        M = ctx.x_rows
        if dout is None:
            dout = torch.zeros_like(xhat)
        dout = _dev_f32(dout, "dout")
        d16 = None if dout16 is None else _dev_bf16(dout16, "dout16")
        dx = torch.empty(xhat.shape, dtype=torch.bfloat16, device=xhat.device)
        dres = torch.empty_like(xhat)
        L, st = _lib.lib(), _stream()
        _lib.check(_timed("bias_drop_res_ln_bwd", 14 * xhat.numel(), 0,
                          lambda: L.hopmi_bias_dropout_residual_layernorm_bwd_dt(
                              dout.data_ptr(), _ptr(d16), xhat.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dx.data_ptr(),
                              dres.data_ptr(), M, D, ctx.p_drop, ctx.seed, ctx.sp, _BF16, st)),
                   "hopmi_bias_dropout_residual_layernorm_bwd_dt")
        if tuple(ctx.res_shape) != tuple(dres.shape):
            dres = dres.view(-1, *ctx.res_shape).sum(0)
        return dx, None, dres, None, None, None, None, None


def bias_dropout_residual_layernorm_bf16(x, bias, res, gamma, beta, eps, p_drop=0.0, seed=0):
    return _BiasDropResLnBf16Fn.apply(x, bias, res, gamma, beta, eps, p_drop, seed)


class _BertAttnBf16Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, p_drop, seed):
        qkv = _dev_bf16(qkv, "qkv")
        B, L, three, H, dh = qkv.shape
        if three != 3 or dh != 64 or L > 64:
            raise _lib.HopmiError(f"hopmi bert_attn: unsupported qkv shape {tuple(qkv.shape)} (need (B, L<=64, 3, H, 64))")
        out = torch.empty(B, L, H * dh, dtype=torch.bfloat16, device=qkv.device)
        Lb, st, sp = _lib.lib(), _stream(), _seed_ptr()
        _lib.check(_timed("bert_attn_fwd", 2 * 4 * B * L * H * dh, 4 * B * H * L * L * dh,
                          lambda: Lb.hopmi_bert_attn_fwd_dt(qkv.data_ptr(), out.data_ptr(), B, L, H, float(p_drop), int(seed), sp, _BF16, st)),
                   "hopmi_bert_attn_fwd_dt")
        ctx.save_for_backward(qkv)
        ctx.p_drop, ctx.seed, ctx.sp = float(p_drop), int(seed), sp
        return out

    @staticmethod
    def backward(ctx, dout):
        (qkv,) = ctx.saved_tensors
        dout = _dev_bf16(dout, "dout")
        B, L, _, H, dh = qkv.shape
        dqkv = torch.empty_like(qkv)
        Lb, st = _lib.lib(), _stream()
        _lib.check(_timed("bert_attn_bwd", 2 * 7 * B * L * H * dh, 10 * B * H * L * L * dh,
                          lambda: Lb.hopmi_bert_attn_bwd_dt(qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), B, L, H,
                                                            ctx.p_drop, ctx.seed, ctx.sp, _BF16, st)), "hopmi_bert_attn_bwd_dt")
        return dqkv, None, None


def bert_attention_bf16(qkv, p_drop=0.0, seed=0):
    """bert_attention on a bf16 fused-QKV GEMM output, bf16 result."""
    return _BertAttnBf16Fn.apply(qkv, p_drop, seed)


def bias_dropout_residual_layernorm(x, bias, res, gamma, beta, eps, p_drop=0.0, seed=0):
    """LayerNorm(dropout(x + bias) + res) * gamma + beta; gradients w.r.t. x and res only."""
    return _BiasDropResLnFn.apply(x, bias, res, gamma, beta, eps, p_drop, seed)


# ------------------------------------------------------------------- reprogramming cross-attention
_M32 = 0xFFFFFFFF

# Dropout stream position in DEVICE memory (a 1-element int32 tensor) or None.  Every seeded kernel draws its mask
# from hash(seed + *SEED_DEV, ...): the host-side `seed` distinguishes the call sites of a step, the device word
# distinguishes the steps -- a captured hipGraph of the training step (graph.GraphedTrainStep) adds to it at the head
# of every replay, so replays draw fresh masks although the kernel arguments are frozen in the graph.
SEED_DEV = None


def base_seed() -> int:
    """torch's seed with the data-parallel rank folded in: replicas are seeded identically (same initial weights) but
    must draw different dropout masks (the reference's per-process generators diverge the same way)."""
    s = torch.initial_seed()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        s += 0x9E3779B97F4A7C15 * (torch.distributed.get_rank() + 1)
    return s & 0xFFFFFFFFFFFFFFFF


def _seed_ptr():
    return None if SEED_DEV is None else SEED_DEV.data_ptr()


def attn_keep_mask(seed: int, N: int, H: int, S: int, p_drop: float, device, pairs: bool = False) -> torch.Tensor:
    """The dropout keep-mask of the attention kernels as a (N,H,S) bool tensor: the same stateless hash (murmur3 finaliser of
    seed ^ row*c1 ^ key*c2 ^ head*c3) evaluated with integer tensor ops.  pairs=False: hopmi_bert_attn_* (one hash per key,
    compared with p_drop * 2^32); pairs=True: hopmi_reprog_attn_* (one hash per pair of adjacent keys 2p, 2p+1: its low and
    high 16 bits compared with p_drop * 2^16)."""
    row = torch.arange(N, device=device, dtype=torch.int64).view(N, 1, 1)
    head = torch.arange(H, device=device, dtype=torch.int64).view(1, H, 1)
    nk = (S + 1) // 2 if pairs else S
    key = torch.arange(nk, device=device, dtype=torch.int64).view(1, 1, nk)
    x = (seed & _M32) ^ ((row * 0x9E3779B1) & _M32) ^ ((key * 0x85EBCA77) & _M32) ^ ((head * 0xC2B2AE3D) & _M32)
    x = x ^ (x >> 16)
    x = (x * 0x85EBCA6B) & _M32
    x = x ^ (x >> 13)
    x = (x * 0xC2B2AE35) & _M32
    x = x ^ (x >> 16)
    # the kernels receive p_drop as a C float and form the threshold from that value: round the same way here (with
    # the double 0.1 the thresholds differ by 7, i.e. one mask bit in ~6e8 elements)
    p32 = float(torch.tensor(p_drop, dtype=torch.float32))
    if pairs:
        t16 = int(p32 * 65536.0)
        return torch.stack([(x & 0xFFFF) >= t16, (x >> 16) >= t16], dim=-1).reshape(N, H, 2 * nk)[:, :, :S]
    return x >= int(p32 * 4294967296.0)


class _BertAttnFn(torch.autograd.Function):
    """BERT self-attention on the fused projection output (hopmi_bert_attn_fwd / _bwd): qkv (B,L,3,H,64) ->
    (B,L,H*64); the backward recomputes the probabilities and returns the gradient in the qkv layout."""

    @staticmethod
    @_fwd32
    def forward(ctx, qkv, p_drop, seed, vmax=None):
        qkv = _dev_f32(qkv, "qkv")
        B, L, three, H, dh = qkv.shape
        if three != 3 or dh != 64 or L > 64:
            raise _lib.HopmiError(f"hopmi bert_attn: unsupported qkv shape {tuple(qkv.shape)} (need (B, L<=64, 3, H, 64))")
        out = torch.empty(B, L, H * dh, dtype=torch.float32, device=qkv.device)
        Lb, st, sp = _lib.lib(), _stream(), _seed_ptr()
        img = sc = None
        D, M = H * dh, B * L
        if vmax is not None and D % 128 == 0 and tuple(vmax[0].shape) == (3 * D // 128, M) and vmax[0].is_contiguous():
            # the output's operand image for the attention-output GEMM, scaled per clip from the QKV product's row maxima of V
            img = torch.empty(Lb.hopmi_rows_image_f16_bytes(M, D), dtype=torch.uint8, device=qkv.device)
            sc = torch.empty(2, M, dtype=torch.float32, device=qkv.device)
            _lib.check(_timed("bert_attn_fwd", 4 * 5 * B * L * H * dh, 4 * B * H * L * L * dh,
                              lambda: Lb.hopmi_bert_attn_fwd_im(qkv.data_ptr(), out.data_ptr(), vmax[0].data_ptr(), 2 * D // 128, 3 * D // 128,
                                                                img.data_ptr(), sc.data_ptr(), B, L, H, float(p_drop), int(seed), sp, st)),
                       "hopmi_bert_attn_fwd_im")
        else:
            _lib.check(_timed("bert_attn_fwd", 4 * 4 * B * L * H * dh, 4 * B * H * L * L * dh,
                              lambda: Lb.hopmi_bert_attn_fwd(qkv.data_ptr(), out.data_ptr(), B, L, H, float(p_drop), int(seed), sp, st)),
                       "hopmi_bert_attn_fwd")
        ctx.save_for_backward(qkv)
        ctx.p_drop, ctx.seed, ctx.sp = float(p_drop), int(seed), sp
        if img is not None:
            ctx.mark_non_differentiable(img, sc)
            ctx.set_materialize_grads(False)         # (no 13 MB zero-fill for the image's "gradient" in front of every backward)
        return out, img, sc

    @staticmethod
    @_bwd32
    def backward(ctx, dout, *_unused):
        (qkv,) = ctx.saved_tensors
        if dout is None:
            return None, None, None, None
        dout = _dev_f32(dout, "dout")
        B, L, _, H, dh = qkv.shape
        dqkv = torch.empty_like(qkv)
        Lb, st = _lib.lib(), _stream()
        _lib.check(_timed("bert_attn_bwd", 4 * 7 * B * L * H * dh, 10 * B * H * L * L * dh,
                          lambda: Lb.hopmi_bert_attn_bwd(qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), B, L, H,
                                                         ctx.p_drop, ctx.seed, ctx.sp, st)), "hopmi_bert_attn_bwd")
        return dqkv, None, None, None


ATTN_IMG = __import__("os").environ.get("HOPMI_ATTN_IMG", "1") != "0"     # the attention kernel writes the next GEMM's operand image


def bert_attention(qkv: torch.Tensor, p_drop: float = 0.0, seed: int = 0, v_rowmax=None) -> torch.Tensor:
    """dropout(softmax(q k^T / 8)) v for every (clip, head) of qkv (B,L,3,H,64) -> (B,L,H*64).  `v_rowmax`: the (c_rowmax, tiles)
    pair the QKV product left (`rowmax` of split_linear): the kernel then also writes the output's fp16 hi / lo operand image for
    the attention-output GEMM."""
    out, img, sc = _BertAttnFn.apply(qkv, p_drop, seed, v_rowmax)
    if img is not None:
        _attach_img(out, img, sc)
    return out


class _ReprogAttnFn(torch.autograd.Function):
    """softmax(q k^T * scale) (dropout) v over the S prototypes without materialising the scores
    (hopmi_reprog_attn_fwd_dt / _bwd_dt).  q (B,L,H,E); k, v (S,H,E).  All three bf16 (as they leave the projections under
    bf16 autocast): read as they are, o and dq leave in bf16 (the `dtype` ABI argument); otherwise fp32."""

    @staticmethod
    def forward(ctx, q, k, v, scale, p_drop, seed):
        typed = q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16
        if typed:
            q, k, v = _dev_bf16(q, "q"), _dev_bf16(k, "k"), _dev_bf16(v, "v")
        else:
            q, k, v = _dev_f32(q.float(), "q"), _dev_f32(k.float(), "k"), _dev_f32(v.float(), "v")
        B, Lq, H, E = q.shape
        S = k.shape[0]
        if k.shape != (S, H, E) or v.shape != (S, H, E):
            raise _lib.HopmiError(f"hopmi reprog_attn: bad shapes q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)}")
        o = torch.empty_like(q)
        lse = torch.empty(B, Lq, H, dtype=torch.float32, device=q.device)
        Lb, st, sp = _lib.lib(), _stream(), _seed_ptr()
        N = B * Lq
        eb = 2 if typed else 4
        ws = torch.empty(Lb.hopmi_reprog_attn_ws_bytes(S, H, E), dtype=torch.uint8, device=q.device)
        _lib.check(_timed("reprog_attn_fwd", eb * (2 * N * H * E + 2 * S * H * E), 4 * N * H * S * E,
                          lambda: Lb.hopmi_reprog_attn_fwd_dt(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), 1 if typed else 0,
                                                              lse.data_ptr(), ws.data_ptr(), N, S, H, E, float(scale), float(p_drop),
                                                              int(seed) & _M32, sp, st)), "hopmi_reprog_attn_fwd")
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.scale, ctx.p_drop, ctx.seed, ctx.sp, ctx.typed = float(scale), float(p_drop), int(seed) & _M32, sp, typed
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        B, Lq, H, E = q.shape
        S = k.shape[0]
        with torch.autocast("cuda", enabled=False):
            do = _dev_bf16(do.to(torch.bfloat16), "do") if ctx.typed else _dev_f32(do.float(), "do")
            delta = (do.float() * o.float()).sum(-1)                    # (B,L,H): the only reduction left to torch
            Lb, st = _lib.lib(), _stream()
            R = Lb.hopmi_reprog_attn_bwd_splits()
            dq = torch.empty_like(q)
            dk = torch.empty((R,) + tuple(k.shape), dtype=torch.float32, device=q.device)
            dv = torch.empty_like(dk)
            N = B * Lq
            eb = 2 if ctx.typed else 4
            ws = torch.empty(Lb.hopmi_reprog_attn_bwd_ws_bytes(N, S, H, E), dtype=torch.uint8, device=q.device)
            _lib.check(_timed("reprog_attn_bwd", eb * (4 * N * H * E + 2 * S * H * E) + 8 * S * H * E, 14 * N * H * S * E,
                              lambda: Lb.hopmi_reprog_attn_bwd_dt(q.data_ptr(), k.data_ptr(), v.data_ptr(), do.data_ptr(),
                                                                  1 if ctx.typed else 0, lse.data_ptr(), delta.data_ptr(), dq.data_ptr(),
                                                                  dk.data_ptr(), dv.data_ptr(), ws.data_ptr(), N, S, H, E, ctx.scale,
                                                                  ctx.p_drop, ctx.seed, ctx.sp, st)),
                       "hopmi_reprog_attn_bwd")
            dk, dv = dk.sum(0), dv.sum(0)
            if ctx.typed:
                dk, dv = dk.to(torch.bfloat16), dv.to(torch.bfloat16)
        return dq, dk, dv, None, None, None


class _HopLossesFn(torch.autograd.Function):
    """The generator losses of one train_llm step (train_llm.py:46-79) in two launches (hopmi_hop_losses_fwd) and their
    gradient in one (hopmi_hop_losses_bwd).  Returns (total, vals): total = w_reg huber + w_div div_reg + w_kld kld is the
    differentiable scalar, vals = (huber, div_reg, kld, total) for reporting."""

    @staticmethod
    def forward(ctx, out, target, out_rand, zc, zr, mu, lv, w_reg, w_div, w_kld):
        f = lambda t, n: None if t is None else _dev_f32(t, n)
        out, target = _dev_f32(out, "out"), _dev_f32(target, "target")
        out_rand, zc, zr, mu, lv = f(out_rand, "out_rand"), f(zc, "z_context"), f(zr, "z_rand"), f(mu, "mu"), f(lv, "logvar")
        B = out.shape[0]
        F_ = out.numel() // B
        zt = zc if zc is not None else mu
        Z = 0 if zt is None else zt.numel() // B
        if target.shape != out.shape or (out_rand is not None and out_rand.shape != out.shape):
            raise _lib.HopmiError(f"hopmi hop_losses: shapes out{tuple(out.shape)} target{tuple(target.shape)}")
        L, st = _lib.lib(), _stream()
        vals = torch.empty(4, dtype=torch.float32, device=out.device)
        ws = torch.empty(L.hopmi_hop_losses_ws_floats(B), dtype=torch.float32, device=out.device)
        _lib.check(L.hopmi_hop_losses_fwd(out.data_ptr(), target.data_ptr(), _ptr(out_rand), _ptr(zc), _ptr(zr), _ptr(mu), _ptr(lv),
                                          B, F_, Z, float(w_reg), float(w_div), float(w_kld), vals.data_ptr(), ws.data_ptr(), st),
                   "hopmi_hop_losses_fwd")
        ctx.save_for_backward(out, target, out_rand, mu, lv, ws)
        ctx.dims, ctx.w = (B, F_, Z), (float(w_reg), float(w_kld))
        total = vals[3].clone()
        ctx.mark_non_differentiable(vals)
        return total, vals

    @staticmethod
    def backward(ctx, g, _unused):
        out, target, out_rand, mu, lv, ws = ctx.saved_tensors
        B, F_, Z = ctx.dims
        g = _dev_f32(g, "g")
        d_out = torch.empty_like(out)
        d_mu = torch.empty_like(mu) if mu is not None else None
        d_lv = torch.empty_like(lv) if lv is not None else None
        _lib.check(_lib.lib().hopmi_hop_losses_bwd(out.data_ptr(), target.data_ptr(), _ptr(out_rand), _ptr(mu), _ptr(lv),
                                                   ws.data_ptr(), g.data_ptr(), B, F_, Z, ctx.w[0], ctx.w[1], d_out.data_ptr(),
                                                   _ptr(d_mu), _ptr(d_lv), _stream()), "hopmi_hop_losses_bwd")
        return d_out, None, None, None, None, d_mu, d_lv, None, None, None


def hop_losses(out, target, out_rand=None, z_context=None, z_rand=None, mu=None, logvar=None, w_reg=1.0, w_div=0.0, w_kld=0.0):
    """(total, vals): see _HopLossesFn.  out_rand / z_context / z_rand (all or none) enable the diversity regulariser,
    mu / logvar (both or none) the KLD term; out_rand, z_context, z_rand and target take no gradient."""
    return _HopLossesFn.apply(out, target, out_rand, z_context, z_rand, mu, logvar, w_reg, w_div, w_kld)


def reprog_attention(q, k, v, scale, p_drop=0.0, seed=0):
    if STRICT_FP32 and q.dtype == torch.float32:
        # HOP.py:289-299 as fp32 tensor operations (scores materialised; torch's dropout stream, not the kernel's hash)
        p = torch.softmax(scale * torch.einsum("blhe,she->bhls", q, k.float()), dim=-1)
        if p_drop > 0.0:
            p = torch.nn.functional.dropout(p, p_drop, True)
        return torch.einsum("bhls,she->blhe", p, v.float())
    return _ReprogAttnFn.apply(q, k, v, scale, p_drop, seed)


# ------------------------------------------------------------------------- fused WaveNet layer
def _ptr(t):
    return None if t is None else t.data_ptr()


def _conv_w(w):
    """Data pointer of a gated-TCN Conv2d weight (64, 64, 1, 2): the kernels read nn.Conv2d's own layout."""
    if w.shape != (64, 64, 1, 2) or not w.is_contiguous() or w.dtype != torch.float32:
        raise _lib.HopmiError(f"hopmi wn_layer: expected a contiguous float32 (64,64,1,2) Conv2d weight, got {tuple(w.shape)} {w.dtype}")
    return w.data_ptr()


def wn_prepare_weights(layers):
    """Split-bf16 MFMA weight images of the fused WaveNet layers (hopmi_wn_prepare_weights): `layers` = a sequence of
    (filter_conv.weight, gate_conv.weight, gconv.mlp.mlp.weight) per layer; one launch for all of them, once per forward
    pass.  Returns a (n_layers, bytes) uint8 tensor; row l is layer l's image (`wimg` of wn_layer_fwd / wn_layer_bwd)."""
    import ctypes
    n = len(layers)
    L = _lib.lib()
    per = L.hopmi_wn_weight_image_bytes(1)
    if n < 1 or L.hopmi_wn_weight_image_bytes(n) == 0:
        raise _lib.HopmiError(f"hopmi wn_prepare_weights: {n} layers unsupported")
    dev = layers[0][0].device
    img = torch.empty(n, per, dtype=torch.uint8, device=dev)
    tabs = []
    for k in range(3):
        ptrs = []
        for l in range(n):
            t = layers[l][k]
            if k < 2:
                ptrs.append(_conv_w(t))
            else:
                if t.numel() != 64 * 192 or not t.is_contiguous() or t.dtype != torch.float32:
                    raise _lib.HopmiError(f"hopmi wn_prepare_weights: Wm must be a contiguous float32 (64,192[,1,1]) tensor, got {tuple(t.shape)}")
                ptrs.append(t.data_ptr())
        tabs.append((ctypes.c_void_p * n)(*ptrs))
    _lib.check(L.hopmi_wn_prepare_weights(tabs[0], tabs[1], tabs[2], n, img.data_ptr(), _stream()), "hopmi_wn_prepare_weights")
    return img


def wn_layer_fwd(xin, scsh_in, wimg, bf, bg, prep, bm, utail, dilation, *, want_y=True, want_fs=False,
                 do_gcn=True, bn=None, stats_keep=None, timer_name="wn_layer_fwd"):
    """One fused WaveNet layer (hopmi_wn_layer_fwd).  No autograd here: the differentiable wrapper is the
    stack-level Function.  xin (B,T_in,V,64) contiguous; wimg: this layer's row of wn_prepare_weights(); utail: a
    (B,4,V,64) view whose last-dim stride is 1 (a channel slice of the (B,4,V,512) skip-tail buffer).  bn = (gamma, beta,
    running_mean, running_var, momentum, eps) for training-mode batch statistics -> returns scale/shift for the next
    layer and (mean, rstd).  `stats_keep`: a list that receives the layer's (mean, rstd, unbiased variance) row so
    that the same running-statistics update can be applied again (wn_bn_replay).  `want_fs` (diagnostic): also return the
    tanh / sigmoid gate values.  Returns (y, fs, scsh_out, mean_rstd)."""
    B, T_in, V, _ = xin.shape
    T_out = T_in - dilation
    dev = xin.device
    xin, dt = _dev_act(xin, "xin")
    esz = 2 if dt else 4
    y = torch.empty(B, T_out, V, 64, dtype=xin.dtype, device=dev) if want_y else None
    fs = torch.empty(B, T_out, V, 128, dtype=torch.float32, device=dev) if want_fs else None
    L = _lib.lib()
    scsh_out = mean_rstd = ws = None
    gamma = beta = rm = rv = None
    momentum, eps = 0.1, 1e-5
    if bn is not None:
        gamma, beta, rm, rv, momentum, eps = bn
        scsh_out = torch.empty(128, dtype=torch.float32, device=dev)
        mean_rstd = torch.empty(192, dtype=torch.float32, device=dev)       # mean, rstd, unbiased variance
        ws = torch.empty(L.hopmi_wn_layer_ws_floats(B, T_in, V, dilation), dtype=torch.float32, device=dev)
    if utail is not None and (utail.stride(-1) != 1 or utail.stride(2) % 4 or utail.shape != (B, 4, V, 64) or utail.dtype != xin.dtype):
        raise _lib.HopmiError(f"hopmi wn_layer: bad utail view {tuple(utail.shape)} strides {utail.stride()} {utail.dtype}")
    if wimg.dtype != torch.uint8 or wimg.numel() != L.hopmi_wn_weight_image_bytes(1) or not wimg.is_contiguous():
        raise _lib.HopmiError("hopmi wn_layer: `wimg` is not one layer's row of wn_prepare_weights()")
    st = _stream()
    n_out = B * T_out * V
    # SURVEY.md 8(d), fused layer: x in + x out + the last-4-frames skip tail (+ the per-launch weight image)
    nbytes = esz * (B * T_in * V * 64 + n_out * 64 * (1 if want_y else 0) + (B * 4 * V * 64 if utail is not None else 0))
    extra = 4 * n_out * 128 * (1 if want_fs else 0)
    flops = n_out * (2 * 2 * 2 * 64 * 64 + (2 * 192 * 64 + 4 * 64 * V if do_gcn else 0))
    _lib.check(_timed(timer_name, nbytes, flops,
                      lambda: L.hopmi_wn_layer_fwd_dt(xin.data_ptr(), scsh_in.data_ptr(), wimg.data_ptr(), bf.data_ptr(), bg.data_ptr(),
                                                      _ptr(prep), _ptr(bm), _ptr(y), _ptr(fs), _ptr(utail),
                                                      utail.stride(2) if utail is not None else 64, _ptr(ws), B, T_in, V, dilation,
                                                      1 if do_gcn else 0, dt, st), exact=True, extra=extra), "hopmi_wn_layer_fwd")
    if bn is not None:
        _lib.check(L.hopmi_wn_bn_finalize(ws.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(rm), _ptr(rv),
                                          float(momentum), float(eps), scsh_out.data_ptr(), mean_rstd.data_ptr(),
                                          B, T_in, V, dilation, st), "hopmi_wn_bn_finalize")
        if stats_keep is not None:
            stats_keep.append(mean_rstd)
    return y, fs, scsh_out, mean_rstd


# Under bf16 autocast the WaveNet stack keeps its activations (x0, the saved y_l, the skip tails and their gradient) in bf16, as
# the reference's convolutions and BatchNorm do under accelerate's bf16 mode.  False: fp32 storage in every mode (two cast
# launches at the block's boundary, twice the bytes; the BatchNorm backward then sees unrounded y).
WN_BF16_STORAGE = True

_STACK_WS = {}       # (device index, stream, geometry) -> workspace of the persistent stack kernel (counters zero between launches)
STACK_ENABLED = True  # False: the training forward runs as per-layer launches (A/B runs, HOPMI_WN_STACK=0)


def stack_ws_prepare(from_stream: int, to_stream: int):
    """Give `to_stream` (a raw stream handle) a zeroed workspace for every stack geometry `from_stream` has used: the eager
    warm-up calls of a training step run on the current stream, its recording on another one, and a workspace first seen under
    capture would be allocated and ZERO-FILLED by a recorded launch (2 MB of writes and a sequence-number reset on every replay)."""
    for key in [k for k in _STACK_WS if k[1] == from_stream]:
        new = (key[0], to_stream) + key[2:]
        if new not in _STACK_WS:
            _STACK_WS[new] = torch.zeros_like(_STACK_WS[key])


def stack_ws_reset():
    """After a launch raised its status word (sticky; the launch sequence number was not advanced): zero every stack workspace IN
    PLACE, each on the stream it belongs to.  The memory is never released -- a recorded step holds the workspace of its stream by
    ADDRESS (GraphedTrainStep keeps its recordings when the caller catches the time-out error and goes on stepping), so dropping
    the tensors would let the allocator hand the same bytes to somebody else under the next replay's hand-off counters.  All-zero
    tags match no launch (tag = sequence number x 16 + layer + 1 >= 1)."""
    for key, ws in _STACK_WS.items():
        if key[1]:
            with torch.cuda.stream(torch.cuda.ExternalStream(key[1], device=ws.device)):
                ws.zero_()
        else:
            with torch.cuda.stream(torch.cuda.default_stream(ws.device)):
                ws.zero_()


def wn_stack_supported(B: int, T_in: int, V: int, dilations) -> int:
    """Grid of the one-launch WaveNet stack (hopmi_wn_stack_fwd) for this geometry, or 0 when it cannot be used: unsupported
    geometry, or a context in which its workgroups are not guaranteed to be resident together (persistent kernels withheld
    beside RCCL kernels / on a shared device, see no_persistent_gru)."""
    import ctypes
    import os
    if not STACK_ENABLED or os.environ.get("HOPMI_WN_STACK", "1") == "0" or not gru_persistent_allowed():
        return 0
    n = len(dilations)
    return int(_lib.lib().hopmi_wn_stack_grid(B, T_in, V, (ctypes.c_int * n)(*dilations), n))


def wn_stack_fwd(x0, wimg, tcn_biases, prep, mlp_biases, bns, tails, dilations):
    """All WaveNet layers of a training-mode forward as ONE persistent launch (hopmi_wn_stack_fwd): x0 (B,T,V,64) start-conv
    output, wimg = wn_prepare_weights(...) of the n layers, tcn_biases = [(bf, bg)], mlp_biases = [bm], bns = the
    nn.BatchNorm2d modules (affine parameters, running statistics updated in place), tails (B,4,V,64 n) receives the skip
    tails.  Returns (ys, scsh, mean_rstd): ys[l] = layer l's pre-BatchNorm output for l < n - 1 (the last one is dead),
    scsh (n,128) the scale | shift rows, mean_rstd (n,192) mean | rstd | unbiased variance rows."""
    import ctypes
    n = len(dilations)
    B, T_in, V, _ = x0.shape
    dev = x0.device
    L = _lib.lib()
    dil = (ctypes.c_int * n)(*dilations)
    nbytes = L.hopmi_wn_stack_ws_bytes(B, T_in, V, dil, n)
    if nbytes == 0:
        raise _lib.HopmiError(f"hopmi wn_stack_fwd: unsupported geometry: {L.hopmi_last_error().decode()}")
    x0, dt = _dev_act(x0, "x0")
    esz = 2 if dt else 4
    st = _stream()
    key = (dev.index, st, B, T_in, V, tuple(dilations))
    ws = _STACK_WS.get(key)
    if ws is None:
        ws = _STACK_WS[key] = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device=dev)
    if tails.stride(-1) != 1 or tails.stride(2) % 4 or tails.shape != (B, 4, V, 64 * n) or tails.dtype != x0.dtype:
        raise _lib.HopmiError(f"hopmi wn_stack_fwd: bad tails tensor {tuple(tails.shape)} strides {tails.stride()}")
    if wimg.dtype != torch.uint8 or wimg.numel() != L.hopmi_wn_weight_image_bytes(n) or not wimg.is_contiguous():
        raise _lib.HopmiError("hopmi wn_stack_fwd: `wimg` is not the wn_prepare_weights() image of these layers")
    mom, eps = float(bns[0].momentum), float(bns[0].eps)
    if any(float(b.momentum) != mom or float(b.eps) != eps for b in bns):
        raise _lib.HopmiError("hopmi wn_stack_fwd: the BatchNorm layers must share momentum and eps")
    ys, T = [], T_in
    for l in range(n - 1):
        T -= dilations[l]
        ys.append(torch.empty(B, T, V, 64, dtype=x0.dtype, device=dev))
    scsh = torch.empty(n, 128, dtype=torch.float32, device=dev)
    mean_rstd = torch.empty(n, 192, dtype=torch.float32, device=dev)
    tab = lambda ts: (ctypes.c_void_p * len(ts))(*[None if t is None else _dev_f32(t, "stack parameter").data_ptr() for t in ts])
    for t in [b for pair in tcn_biases for b in pair] + list(mlp_biases) + [p for bn in bns for p in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]:
        if t is not None and (not t.is_contiguous() or t.dtype != torch.float32 or t.numel() != 64):
            raise _lib.HopmiError("hopmi wn_stack_fwd: per-layer vectors must be contiguous float32 [64]")
    ytab = (ctypes.c_void_p * max(n - 1, 1))(*[y.data_ptr() for y in ys]) if n > 1 else (ctypes.c_void_p * 1)(None)
    # SURVEY.md 8(d), fused layers: per layer x in + x out (not the dead last one) + the last-4-frames skip tail
    n_rows, nbytes_alg, flops, T = 0, 0, 0, T_in
    for l, d in enumerate(dilations):
        nbytes_alg += esz * 64 * V * (B * T + (B * (T - d) if l < n - 1 else 0) + 4 * B)
        flops += B * (T - d) * V * (2 * 2 * 2 * 64 * 64 + 2 * 192 * 64 + 4 * 64 * V)
        T -= d
    _lib.check(_timed("wn_stack_fwd", nbytes_alg, flops,
                      lambda: L.hopmi_wn_stack_fwd_dt(x0.data_ptr(), wimg.data_ptr(), tab([b[0] for b in tcn_biases]), tab([b[1] for b in tcn_biases]),
                                                   prep.data_ptr(), tab(mlp_biases), tab([bn.weight for bn in bns]), tab([bn.bias for bn in bns]),
                                                   tab([bn.running_mean for bn in bns]), tab([bn.running_var for bn in bns]), mom, eps, ytab,
                                                   tails.data_ptr(), tails.stride(2), scsh.data_ptr(), mean_rstd.data_ptr(), ws.data_ptr(),
                                                   B, T_in, V, dil, n, dt, st), exact=True), "hopmi_wn_stack_fwd")
    _track_status(ws[:48])                              # (status word = int 32 of the control block = index [-16] of this view)
    return ys, scsh, mean_rstd


def wn_fused_training_supported(V: int) -> bool:
    """The fused backward keeps five tile images and both mix-matrix images in LDS: that fits for V <= 42 (the
    reference's two skeletons are 9 and 42 nodes).  Larger graphs train through the composed path (gcn kernel + GEMMs)."""
    return _lib.lib().hopmi_wn_layer_bwd_ws_floats(1, 16, int(V), 1) > 0


def wn_bn_replay(kept, bn):
    """Advance bn's running statistics once more with the batch statistics of an earlier wn_layer_fwd call (`kept` = its
    mean / rstd / unbiased-variance row: the same floats the first update used, so this is bit-identical to recomputing)."""
    _lib.check(_lib.lib().hopmi_wn_bn_replay(kept.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                             float(bn.momentum), _stream()), "hopmi_wn_bn_replay")


def wn_layer_bwd(xin, scsh_in, fs, wf, wg, prep, Wm, P0n, P1n, d_next, y, bn_coef, dutail, gamma_prev, mean_rstd_prev,
                 dilation, do_gcn=True, dA=None):
    """Backward of one fused WaveNet layer (hopmi_wn_layer_bwd).  Returns a dict of the outputs the header
    documents; entries that do not exist for this call (do_gcn=False, first layer) are None.  `dA` = (dA1, dA2)
    buffers to ACCUMULATE the adjacency gradients into (one pair for the whole stack) instead of fresh outputs."""
    B, T_in, V, _ = xin.shape
    T_out = T_in - dilation
    dev = xin.device
    xin, dt = _dev_act(xin, "xin")                     # saved activations and the skip-tail gradient: fp32 or bf16 (one storage type)
    if (y is not None and y.dtype != xin.dtype) or dutail.dtype != xin.dtype:
        raise _lib.HopmiError(f"hopmi wn_layer_bwd: xin {xin.dtype}, y {None if y is None else y.dtype}, dutail {dutail.dtype} must share a dtype")
    new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
    out = dict(P0=new(B, T_out, V, 64), P1=new(B, T_out, V, 64), dwf=new(64, 64, 1, 2), dwg=new(64, 64, 1, 2), dbtcn=new(128),
               dWm=new(64, 192) if do_gcn else None, dbm=new(64) if do_gcn else None,
               dA1=(dA[0] if dA is not None else new(V, V)) if do_gcn else None,
               dA2=(dA[1] if dA is not None else new(V, V)) if do_gcn else None,
               dgamma_prev=new(64) if gamma_prev is not None else None,
               dbeta_prev=new(64) if gamma_prev is not None else None,
               coef_prev=new(3, 64) if gamma_prev is not None else None)
    L = _lib.lib()
    ws = new(L.hopmi_wn_layer_bwd_ws_floats(B, T_in, V, dilation))
    if dutail.stride(-1) != 1 or dutail.stride(2) % 4 or dutail.shape != (B, 4, V, 64):
        raise _lib.HopmiError(f"hopmi wn_layer_bwd: bad dutail view {tuple(dutail.shape)} strides {dutail.stride()}")
    st = _stream()
    n_out = B * T_out * V
    _lib.check(_timed("wn_layer_bwd", 4 * (2 * B * T_in * V * 64 + n_out * 64 * 5 + n_out * 128), 0,
                      lambda: L.hopmi_wn_layer_bwd_dt(xin.data_ptr(), scsh_in.data_ptr(), fs.data_ptr(), _conv_w(wf), _conv_w(wg),
                                                   _ptr(prep), _ptr(Wm), _ptr(P0n), _ptr(P1n), int(d_next), _ptr(y),
                                                   _ptr(bn_coef), dutail.data_ptr(), dutail.stride(2), _ptr(gamma_prev),
                                                   _ptr(mean_rstd_prev), out["P0"].data_ptr(), out["P1"].data_ptr(),
                                                   out["dwf"].data_ptr(), out["dwg"].data_ptr(), out["dbtcn"].data_ptr(),
                                                   _ptr(out["dWm"]), _ptr(out["dbm"]), _ptr(out["dA1"]), _ptr(out["dA2"]),
                                                   1 if dA is not None else 0, _ptr(out["dgamma_prev"]), _ptr(out["dbeta_prev"]), _ptr(out["coef_prev"]),
                                                   ws.data_ptr(), B, T_in, V, dilation, 1 if do_gcn else 0, dt, st)),
               "hopmi_wn_layer_bwd")
    return out


# ------------------------------------------------------------------------------------------- GRU
GRU_CHECK_STATUS = False      # tests set this: reads the persistent kernel's status word back (a host sync)
_PENDING_STATUS = []          # status words of persistent launches not yet checked (see deferred_status)
_PERSISTENT_HOLD = 0          # > 0: persistent GRU launches are withheld (see no_persistent_gru)


class no_persistent_gru:
    """Within the block the GRU recurrences run as per-time-step launches.  The persistent kernels need every one of their
    workgroups resident at once (they hand h_t over through counters); that only holds while nothing else competes for
    the CUs.  GradSync wraps a backward during which RCCL kernels run beside the compute stream in this; a shared
    device (HOPMI_REHEARSE_ONE_GPU) or HOPMI_GRU_PERSISTENT=0 withhold them for the whole process."""

    def __enter__(self):
        global _PERSISTENT_HOLD
        _PERSISTENT_HOLD += 1

    def __exit__(self, *exc):
        global _PERSISTENT_HOLD
        _PERSISTENT_HOLD -= 1


def gru_persistent_allowed() -> bool:
    import os
    return (_PERSISTENT_HOLD == 0 and not STRICT_FP32 and os.environ.get("HOPMI_GRU_PERSISTENT", "1") != "0"
            and os.environ.get("HOPMI_REHEARSE_ONE_GPU") != "1")


def deferred_status():
    """Sum of the status words of all persistent GRU launches since the last call, as a 0-dim float tensor (or
    None).  The training step folds it into its one device->host transfer and raises if it is non-zero, so a
    hand-off time-out can never pass silently and costs no extra sync."""
    global _PENDING_STATUS
    if not _PENDING_STATUS:
        return None
    st = torch.stack(_PENDING_STATUS).sum().float()
    _PENDING_STATUS = []
    return st


def check_status_now():
    """Inference entry points (no training step drains the list for them): read the pending status words back and
    raise if a persistent hand-off timed out.  One small device->host copy."""
    st = deferred_status()
    if st is not None and float(st.item()) != 0.0:
        raise _lib.HopmiError("hopmi gru: a persistent-kernel hand-off timed out (status word set); "
                              "set HOPMI_GRU_PERSISTENT=0 to use per-time-step launches")


STATUS_SINK = None            # a list while a training-step graph is being captured: receives the status words


def _track_status(ws):
    if ws is None:
        return
    if STATUS_SINK is not None:                 # captured launches: whoever replays the graph reads these words
        STATUS_SINK.append(ws[-16])
        return
    if torch.cuda.is_current_stream_capturing():
        return
    if len(_PENDING_STATUS) >= 1024:            # nobody drains (e.g. a long eval loop): fold what is there
        folded = torch.stack(_PENDING_STATUS).sum()
        _PENDING_STATUS.clear()
        _PENDING_STATUS.append(folded)
    _PENDING_STATUS.append(ws[-16])


class _GruLayerFn(torch.autograd.Function):
    """Recurrence of one bidirectional GRU layer (hopmi_gru_fwd_dt / hopmi_gru_bwd_dt).

    gi (B,T,2,3H) = input projections of both directions, fp32 or -- under bf16 autocast, straight from the library GEMM --
    bf16 (read as is: the `gi_dtype` ABI argument; its gradient goes back in the same type); whh (2,3H,H); bhh (2,3H) ->
    y (B,T,2H) fp32.  Everything inside runs with autocast off (fp32 weight-gradient GEMMs)."""

    @staticmethod
    def forward(ctx, gi, whh, bhh):
        typed = gi.dtype == torch.bfloat16
        gi = _dev_bf16(gi, "gi") if typed else _dev_f32(gi.float(), "gi")
        whh, bhh = _dev_f32(whh.float(), "whh"), _dev_f32(bhh.float(), "bhh")
        B, T, two, H3 = gi.shape
        H = H3 // 3
        if two != 2 or whh.shape != (2, 3 * H, H) or bhh.shape != (2, 3 * H):
            raise _lib.HopmiError(f"hopmi gru: bad shapes gi{tuple(gi.shape)} whh{tuple(whh.shape)} bhh{tuple(bhh.shape)}")
        y = torch.empty(B, T, 2 * H, dtype=torch.float32, device=gi.device)
        gates = torch.empty(B, T, 2, 4 * H, dtype=torch.float32, device=gi.device)
        L, st = _lib.lib(), _stream()
        ws = (torch.empty(L.hopmi_gru_ws_bytes(B, T, H) // 4, dtype=torch.int32, device=gi.device)
              if gru_persistent_allowed() else None)
        _lib.check(_timed("gru_fwd", 0, 2 * T * B * 2 * 3 * H * H,
                          lambda: L.hopmi_gru_fwd_dt(gi.data_ptr(), 1 if typed else 0, whh.data_ptr(), bhh.data_ptr(), y.data_ptr(),
                                                     gates.data_ptr(), _ptr(ws), B, T, H, st)), "hopmi_gru_fwd")
        _track_status(ws)
        if GRU_CHECK_STATUS and ws is not None and not torch.cuda.is_current_stream_capturing() and int(ws[-16].item()) != 0:
            raise _lib.HopmiError("hopmi gru: a persistent-kernel hand-off timed out (status word set)")
        ctx.save_for_backward(y, gates, whh)
        ctx.typed = typed
        return y

    @staticmethod
    def backward(ctx, dy):
        y, gates, whh = ctx.saved_tensors
        return _gru_layer_backward(y, gates, whh, dy, ctx.typed, ctx.needs_input_grad[1] or ctx.needs_input_grad[2])


def _gru_layer_backward(y, gates, whh, dy, typed, want_dw):
    """(dgi, dwhh, dbhh) of one bidirectional GRU layer's recurrence from its saved output and gates (hopmi_gru_bwd_dt): shared by
    _GruLayerFn and _GruLayerPairFn (whose y / gates are the graded batch's leading rows of the pair's tensors)."""
    with torch.autocast("cuda", enabled=False):
        dy = _dev_f32(dy.float(), "dy")
        B, T, H2 = y.shape
        H = H2 // 2
        L, st = _lib.lib(), _stream()
        # the recurrence's W_hh^T and the shifted states for dW_hh: one launch (was a transpose copy, a fill, two strided copies)
        whhT = torch.empty(2, H, 3 * H, dtype=torch.float32, device=y.device)
        hprev = torch.empty(B, T, 2, H, dtype=torch.float32, device=y.device)
        _lib.check(L.hopmi_gru_bwd_operands(y.data_ptr(), whh.data_ptr(), hprev.data_ptr(), whhT.data_ptr(), B, T, H, st),
                   "hopmi_gru_bwd_operands")
        dgi = torch.empty(B, T, 2, 3 * H, dtype=torch.bfloat16 if typed else torch.float32, device=y.device)
        dgh = torch.empty(B, T, 2, 3 * H, dtype=torch.float32, device=y.device)
        ws = torch.empty(L.hopmi_gru_bwd_ws_floats(B, H), dtype=torch.float32, device=y.device)
        ws2 = (torch.empty(L.hopmi_gru_ws_bytes(B, T, H) // 4, dtype=torch.int32, device=y.device)
               if gru_persistent_allowed() else None)
        _lib.check(_timed("gru_bwd", 0, 2 * T * B * 2 * 3 * H * H,
                          lambda: L.hopmi_gru_bwd_dt(dy.data_ptr(), y.data_ptr(), gates.data_ptr(), whhT.data_ptr(),
                                                     dgi.data_ptr(), 1 if typed else 0, dgh.data_ptr(), ws.data_ptr(), _ptr(ws2),
                                                     B, T, H, st)),
                   "hopmi_gru_bwd")
        _track_status(ws2)
        if GRU_CHECK_STATUS and ws2 is not None and not torch.cuda.is_current_stream_capturing() and int(ws2[-16].item()) != 0:
            raise _lib.HopmiError("hopmi gru: a persistent-kernel hand-off timed out in the backward (status word set)")
        # dW_hh[d] = sum_{b,t} dgh[b,t,d,:]^T h_prev[b,t,d,:]  with h_prev = y shifted one step along each
        # direction's processing order (zero at its first step); db_hh = sum dgh.  Plain GEMMs.
        if not want_dw:                    # (frozen recurrent weights, e.g. the discriminator inside the generator's step)
            return dgi, None, None
        M = B * T
        dgh2, hp2 = dgh.view(M, 2, 3 * H).transpose(0, 1), hprev.view(M, 2, H).transpose(0, 1)       # (2, M, 3H) / (2, M, H) views
        if f16_mm_tn_ok(dgh2, hp2) and (6 * H) % 4 == 0:
            # the states are bounded by 1: a constant scale; dgh: one pass over the (M, 6H) rows, shared by the two directions
            dwhh = f16_mm_tn(dgh2, hp2, row_scales(dgh.view(M, 6 * H)), unit_row_scales(M, y.device))
        else:
            dwhh = torch.einsum("btdg,btdh->dgh", dgh, hprev)
        dbhh = colsum(dgh.view(B * T, 6 * H)).view(2, 3 * H)
    return dgi, dwhh, dbhh


def gru_layer(gi: torch.Tensor, whh: torch.Tensor, bhh: torch.Tensor) -> torch.Tensor:
    return _GruLayerFn.apply(gi, whh, bhh)


class _GruLayerPairFn(torch.autograd.Function):
    """One recurrence launch for TWO batches on the same weights (hopmi_gru_fwd_pair_dt): gi1 (B1,T,2,3H) takes gradients, gi2
    (B2,T,2,3H) is the no-grad forward's (its output is marked non-differentiable).  The backward is _GruLayerFn's on the first B1
    rows: y and gates are allocated whole, the graded batch is their leading (contiguous) part."""

    @staticmethod
    def forward(ctx, gi1, gi2, whh, bhh):
        typed = gi1.dtype == torch.bfloat16
        if gi2.dtype != gi1.dtype:
            gi2 = gi2.to(gi1.dtype)
        gi1 = _dev_bf16(gi1, "gi1") if typed else _dev_f32(gi1.float(), "gi1")
        gi2 = _dev_bf16(gi2.detach(), "gi2") if typed else _dev_f32(gi2.detach().float(), "gi2")
        whh, bhh = _dev_f32(whh.float(), "whh"), _dev_f32(bhh.float(), "bhh")
        B1, T, two, H3 = gi1.shape
        B2 = gi2.shape[0]
        H = H3 // 3
        if two != 2 or tuple(gi2.shape[1:]) != (T, 2, H3) or whh.shape != (2, 3 * H, H) or bhh.shape != (2, 3 * H):
            raise _lib.HopmiError(f"hopmi gru pair: bad shapes gi1{tuple(gi1.shape)} gi2{tuple(gi2.shape)} whh{tuple(whh.shape)}")
        B = B1 + B2
        y = torch.empty(B, T, 2 * H, dtype=torch.float32, device=gi1.device)
        gates = torch.empty(B, T, 2, 4 * H, dtype=torch.float32, device=gi1.device)
        L, st = _lib.lib(), _stream()
        ws = (torch.empty(L.hopmi_gru_ws_bytes(B, T, H) // 4, dtype=torch.int32, device=gi1.device)
              if gru_persistent_allowed() else None)
        _lib.check(_timed("gru_fwd", 0, 2 * T * B * 2 * 3 * H * H,
                          lambda: L.hopmi_gru_fwd_pair_dt(gi1.data_ptr(), gi2.data_ptr(), B1, 1 if typed else 0, whh.data_ptr(), bhh.data_ptr(),
                                                          y.data_ptr(), gates.data_ptr(), _ptr(ws), B, T, H, st)), "hopmi_gru_fwd_pair")
        _track_status(ws)
        if GRU_CHECK_STATUS and ws is not None and not torch.cuda.is_current_stream_capturing() and int(ws[-16].item()) != 0:
            raise _lib.HopmiError("hopmi gru: a persistent-kernel hand-off timed out (status word set)")
        y1, y2 = y[:B1], y[B1:]
        ctx.save_for_backward(y1, gates[:B1], whh)
        ctx.typed = typed
        ctx.mark_non_differentiable(y2)
        return y1, y2

    @staticmethod
    def backward(ctx, dy, _dy2=None):
        y1, gates1, whh = ctx.saved_tensors
        dgi, dwhh, dbhh = _gru_layer_backward(y1, gates1, whh, dy, ctx.typed, ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        return dgi, None, dwhh, dbhh


GRU_PAIR = __import__("os").environ.get("HOPMI_GRU_PAIR", "1") != "0"


def gru_bidirectional_pair(x1: torch.Tensor, x2: torch.Tensor, gru: torch.nn.GRU):
    """gru_bidirectional of two batches on the same module with ONE recurrence launch per layer: x1 (B1,T,in) takes part in autograd,
    x2 (B2,T,in) is a no-grad forward's input -> (y1 (B1,T,2H), y2 (B2,T,2H), y2 without a graph).  The values are those of two
    gru_bidirectional calls (fp32 class; not bit for bit where the 32-row kernel takes over, see the kernel).  Inter-layer dropout
    is not supported here (the HOP decoder has none, HOP.py:166-167)."""
    if not (gru.bidirectional and gru.batch_first and gru.bias) or gru.dropout != 0:
        raise _lib.HopmiError("hopmi gru pair: batch_first, bidirectional, biased nn.GRU modules without inter-layer dropout")
    H = gru.hidden_size
    pack = gru.__dict__.get("_hopmi_pack")
    if pack is None:
        pack = gru.__dict__["_hopmi_pack"] = _GruPack()
    bufs = pack.ensure(gru)
    a, b = x1, x2.detach()
    for layer in range(gru.num_layers):
        al = lambda n: _PackedAlias.apply(bufs[layer][n], getattr(gru, f"{n}_l{layer}"), getattr(gru, f"{n}_l{layer}_reverse"))
        owners = (getattr(gru, f"weight_ih_l{layer}"), getattr(gru, f"weight_ih_l{layer}_reverse"))
        w_ih, b_ih = al("weight_ih"), al("bias_ih")
        gi1 = linear(a, w_ih.flatten(0, 1), b_ih.flatten(), owners=owners).view(a.shape[0], a.shape[1], 2, 3 * H)
        with torch.no_grad():
            gi2 = linear(b, w_ih.detach().flatten(0, 1), b_ih.detach().flatten(), owners=owners).view(b.shape[0], b.shape[1], 2, 3 * H)
        a, b = _GruLayerPairFn.apply(gi1, gi2, al("weight_hh"), al("bias_hh"))
    return a, b


class _PackedAlias(torch.autograd.Function):
    """`pack` is a buffer whose storage the parameters `parts` are views of (see _GruPack): hand it to the graph
    as a function of the parameters without copying anything; the gradient goes back as views of one tensor."""

    @staticmethod
    def forward(ctx, pack, *parts):
        return pack.view_as(pack)

    @staticmethod
    def backward(ctx, g):
        return (None, *g.contiguous().unbind(0))


class _GruPack:
    """cuDNN-style flattened weights for the HIP GRU: per layer, one buffer each for
    [weight_ih | weight_ih_reverse] (2,3H,in), [bias_ih | ..] (2,3H), weight_hh (2,3H,H), bias_hh (2,3H).
    The nn.GRU parameters are re-pointed (`param.data`) at slices of these buffers, so the optimizer's in-place
    updates, state_dict() and load_state_dict() keep working on the individual parameters while the kernels and
    GEMMs read both directions as one operand with no per-call concatenation.  `ensure()` re-packs when a
    parameter has been moved (`.to()`, `.float()`, deepcopy)."""

    NAMES = ("weight_ih", "bias_ih", "weight_hh", "bias_hh")

    def __init__(self):
        self.bufs = []          # [layer][name] -> tensor (2, ...)

    def ensure(self, gru):
        ok = len(self.bufs) == gru.num_layers
        if ok:
            for layer, lb in enumerate(self.bufs):
                for n in self.NAMES:
                    buf = lb[n]
                    f, r = getattr(gru, f"{n}_l{layer}"), getattr(gru, f"{n}_l{layer}_reverse")
                    if f.data_ptr() != buf.data_ptr() or r.data_ptr() != buf[1].data_ptr() or f.dtype != buf.dtype:
                        ok = False
        if ok:
            return self.bufs
        with torch.no_grad():
            self.bufs = []
            for layer in range(gru.num_layers):
                lb = {}
                for n in self.NAMES:
                    f, r = getattr(gru, f"{n}_l{layer}"), getattr(gru, f"{n}_l{layer}_reverse")
                    buf = torch.stack([f.detach(), r.detach()]).contiguous()
                    f.data, r.data = buf[0], buf[1]
                    lb[n] = buf
                self.bufs.append(lb)
        return self.bufs


def gru_bidirectional(x: torch.Tensor, gru: torch.nn.GRU, dropout_p: float = 0.0, training: bool = False) -> torch.Tensor:
    """torch.nn.GRU(batch_first=True, bidirectional=True) forward with h0 = 0 on the HIP recurrence.

    x (B,T,in) -> (B,T,2H).  The input projections of all time steps and both directions are one GEMM
    (hipBLASLt); inter-layer dropout follows nn.GRU (on every layer's output but the last)."""
    if not (gru.bidirectional and gru.batch_first and gru.bias):
        raise _lib.HopmiError("hopmi gru: only batch_first, bidirectional, biased nn.GRU modules are supported")
    H = gru.hidden_size
    pack = gru.__dict__.get("_hopmi_pack")
    if pack is None:
        pack = gru.__dict__["_hopmi_pack"] = _GruPack()
    bufs = pack.ensure(gru)
    inp = x
    for layer in range(gru.num_layers):
        al = lambda n: _PackedAlias.apply(bufs[layer][n], getattr(gru, f"{n}_l{layer}"), getattr(gru, f"{n}_l{layer}_reverse"))
        w_ih, b_ih = al("weight_ih"), al("bias_ih")
        gi = linear(inp, w_ih.flatten(0, 1), b_ih.flatten(), owners=(getattr(gru, f"weight_ih_l{layer}"), getattr(gru, f"weight_ih_l{layer}_reverse"))
                    ).view(inp.shape[0], inp.shape[1], 2, 3 * H)
        inp = gru_layer(gi, al("weight_hh"), al("bias_hh"))
        if dropout_p > 0 and training and layer < gru.num_layers - 1:
            inp = torch.nn.functional.dropout(inp, dropout_p, True)
    return inp
