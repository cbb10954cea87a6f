"""Spatio-temporal graph-wavenet block of HOP, MI355X-native.

Drop-in for the reference's `model/gwnet.py` (`nconv`, `linear`, `gcn`, `gwnet`): same
constructor arguments, same `state_dict` keys and shapes, same call signatures
(`gwnet.forward(input[B,173,V,16]) -> [B,173,V,4]`, gwnet.py:143-249).

Inside, activations are channels-last `[B][T][V][64]` (a slab = one (clip, frame) pair of
V x 64 floats, contiguous), the layout the HIP kernels stream; the NCHW tensors of the
reference exist only at the module boundary.  The graph convolution runs in
`libhopmi.so` (ops.gcn); nothing here falls back to an eager implementation of it.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

DILATIONS = (1, 2, 1, 2, 1, 2, 1, 2)            # gwnet.py:98-121 with blocks=4, layers=2


class _SkipPack(torch.autograd.Function):
    """The eight skip convs' weights side by side (256, 8 x 64) and their biases summed, as the one K = 512 GEMM of `_tail`
    reads them.  As tensor operations (cat / stack + sum) the backward hands every parameter a strided or expanded view of the
    packed gradient, which autograd then copies one launch at a time (16 launches); here the packed weight gradient is brought
    into layer-major order once and the bias gradient replicated once, and the parameters receive contiguous views."""

    @staticmethod
    def forward(ctx, *params):
        n = len(params) // 2
        ctx.n = n
        ws = torch.cat([w.flatten(1) for w in params[:n]], 1)
        bs = torch.stack(params[n:]).sum(0)
        return ws, bs

    @staticmethod
    def backward(ctx, dws, dbs):
        n = ctx.n
        O = dws.shape[0]
        dw = dws.reshape(O, n, -1).permute(1, 0, 2).contiguous()                      # (n, 256, 64): one copy
        db = dbs.unsqueeze(0).expand(n, O).contiguous()                               # (n, 256): one copy
        return (*[g.reshape(O, -1, 1, 1) for g in dw.unbind(0)], *db.unbind(0))


class _WaveNetStackFn(torch.autograd.Function):
    """The 8 fused WaveNet layers, training mode, differentiable: x0 (B,T,V,64) start-conv output ->
    (B,4,V,512) skip tails.  Forward = 8 hopmi_wn_layer_fwd calls (BatchNorm batch statistics, running
    statistics advanced in place); backward = 8 hopmi_wn_layer_bwd calls in reverse, each handing the next
    one its gradient as two tap tensors plus the BatchNorm-backward coefficients.

    apply(x0, A1, A2, prep, wimg, bns, keep, *params) with wimg = the layers' weight images (ops.wn_prepare_weights), params = 8 x (wf, bf, wg, bg), 8 x (Wm, bm), 8 x (gamma, beta);
    `bns` is the list of nn.BatchNorm2d modules (running-stat buffers, momentum, eps); `keep` a list that receives
    the per-layer statistics workspaces (gwnet.replay_bn_update)."""

    @staticmethod
    @ops._fwd_any
    def forward(ctx, x0, A1, A2, prep, wimg, bns, keep, *params):
        n = len(DILATIONS)
        tcn = [params[4 * i:4 * i + 4] for i in range(n)]
        mlp = [params[4 * n + 2 * i:4 * n + 2 * i + 2] for i in range(n)]
        aff = [params[6 * n + 2 * i:6 * n + 2 * i + 2] for i in range(n)]
        B, _, V, _ = x0.shape
        dev = x0.device
        # x0 is fp32, or bf16 as the start conv's bf16 GEMM left it: the stack keeps that storage type for its saved
        # activations and for the skip tails (arithmetic, statistics and the gradients between layers are fp32)
        tails = torch.empty(B, 4, V, 64 * n, dtype=x0.dtype, device=dev)
        scsh = _identity_scsh(dev)
        xin = x0.contiguous()
        saved_x, saved_y, saved_scsh, saved_mr = [xin], [], [], []
        if ops.wn_stack_supported(B, xin.shape[1], V, DILATIONS):
            # one persistent launch for the 8 layers (the BatchNorm statistics are exchanged inside it)
            ys, scsh_rows, mr_rows = ops.wn_stack_fwd(xin, wimg, [(t[1], t[3]) for t in tcn], prep, [m[1] for m in mlp], bns, tails, DILATIONS)
            saved_x += ys
            saved_y = list(ys)
            saved_scsh = [scsh] + [scsh_rows[i] for i in range(n - 1)]
            saved_mr = [mr_rows[i] for i in range(n - 1)]
            keep.items.extend(mr_rows[i] for i in range(n))
            if ops.TIMER is not None:
                ops.time_noop_launch()
        else:
            for i, d in enumerate(DILATIONS):
                wf, bf, wg, bg = tcn[i]
                bn = bns[i]
                last = i == n - 1
                y, fs, scsh_out, mean_rstd = ops.wn_layer_fwd(
                    xin, scsh, wimg[i], bf, bg, prep, mlp[i][1], tails[..., 64 * i:64 * (i + 1)], d,
                    want_y=not last, want_fs=False, do_gcn=True,
                    bn=(aff[i][0], aff[i][1], bn.running_mean, bn.running_var, bn.momentum, bn.eps), stats_keep=keep.items)
                saved_scsh.append(scsh)
                if ops.TIMER is not None and i == 0:
                    ops.time_noop_launch()               # the timing method's floor, measured in the same place (bench.py)
                if not last:
                    saved_y.append(y); saved_mr.append(mean_rstd); saved_x.append(y)
                    scsh, xin = scsh_out, y
        ctx.n_x, ctx.n_y = len(saved_x), len(saved_y)
        # the tanh / sigmoid gate values are not kept (2 x the layer output per layer): the backward regenerates them
        ctx.save_for_backward(prep, wimg, *saved_x, *saved_y, *saved_scsh, *saved_mr,
                              *[t[0] for t in tcn], *[t[2] for t in tcn], *[t[1] for t in tcn], *[t[3] for t in tcn],
                              *[m[0] for m in mlp], *[a[0] for a in aff])
        return tails

    @staticmethod
    @ops._bwd32
    def backward(ctx, dtails):
        n = len(DILATIONS)
        sv = list(ctx.saved_tensors)
        prep, wimg = sv.pop(0), sv.pop(0)
        take = lambda k: [sv.pop(0) for _ in range(k)]
        xs, ys, scshs, mrs, wfs, wgs, bfs, bgs, Wms, gammas = (take(n), take(n - 1), take(n), take(n - 1), take(n), take(n),
                                                               take(n), take(n), take(n), take(n))
        dtails = dtails.contiguous()
        if dtails.dtype != xs[0].dtype:
            dtails = dtails.to(xs[0].dtype)
        V = xs[0].shape[2]
        dA1 = torch.zeros(V, V, dtype=torch.float32, device=dtails.device)
        dA2 = torch.zeros_like(dA1)
        g_tcn, g_mlp, g_aff = [None] * n, [(None, None)] * n, [(None, None)] * n
        P0n = P1n = coef = None
        for i in range(n - 1, -1, -1):
            do_gcn = i < n - 1
            # gate values of layer i from its input: a gate-only call of the forward kernel (TCN + gate, no graph conv)
            _, fs_i, _, _ = ops.wn_layer_fwd(xs[i], scshs[i], wimg[i], bfs[i], bgs[i], None, None, None, DILATIONS[i],
                                             want_y=False, want_fs=True, do_gcn=False, timer_name="wn_layer_regate")
            r = ops.wn_layer_bwd(xs[i], scshs[i], fs_i, wfs[i], wgs[i], prep if do_gcn else None, Wms[i] if do_gcn else None,
                                 P0n, P1n, DILATIONS[i + 1] if do_gcn else 1, ys[i] if do_gcn else None, coef,
                                 dtails[..., 64 * i:64 * (i + 1)], gammas[i - 1] if i > 0 else None,
                                 mrs[i - 1] if i > 0 else None, DILATIONS[i], do_gcn=do_gcn, dA=(dA1, dA2))
            g_tcn[i] = (r["dwf"], r["dbtcn"][:64], r["dwg"], r["dbtcn"][64:])       # already in the Conv2d layouts
            if do_gcn:
                g_mlp[i] = (r["dWm"].view(64, 192, 1, 1), r["dbm"])
            if i > 0:
                g_aff[i - 1] = (r["dgamma_prev"], r["dbeta_prev"])
            P0n, P1n, coef = r["P0"], r["P1"], r["coef_prev"]
        # gradient w.r.t. x0: tap 0 lands on frame t, tap 1 (+ residual) on frame t + d of layer 0
        d0 = DILATIONS[0]
        if P0n.shape[1] >= d0:
            # (the same sums as the two padded tensors added, without the two fills and two copies of the padding: one add on
            # the frames both taps reach, one concatenation)
            dx0 = torch.cat([P0n[:, :d0], P0n[:, d0:] + P1n[:, :P1n.shape[1] - d0], P1n[:, P1n.shape[1] - d0:]], dim=1)
        else:
            dx0 = F.pad(P0n, (0, 0, 0, 0, 0, d0)) + F.pad(P1n, (0, 0, 0, 0, d0, 0))
        flat = [t for tup in g_tcn for t in tup] + [t for tup in g_mlp for t in tup] + [t for tup in g_aff for t in tup]
        return (dx0, dA1, dA2, None, None, None, None, *flat)


class _Keep:
    """Per-layer statistics workspaces of the latest training-mode forward (an object, not a list: the autocast
    input caster rebuilds containers)."""

    def __init__(self):
        self.items = []


_SCSH0 = {}


def _identity_scsh(dev):
    """scale = 1, shift = 0 for the first layer's normalise-on-load (a per-device constant)."""
    t = _SCSH0.get(dev)
    if t is None:
        t = _SCSH0[dev] = torch.cat([torch.ones(64, device=dev), torch.zeros(64, device=dev)])
    return t


class nconv(nn.Module):
    """gwnet.py:8-14 -- einsum('ncvl,vw->ncwl'); kept for API parity (tiny, torch)."""

    def forward(self, x, A):
        return torch.einsum("ncvl,vw->ncwl", (x, A)).contiguous()


class linear(nn.Module):
    """gwnet.py:16-22 -- 1x1 Conv2d container (`mlp`)."""

    def __init__(self, c_in, c_out):
        super().__init__()
        self.mlp = nn.Conv2d(c_in, c_out, kernel_size=(1, 1), padding=(0, 0), stride=(1, 1), bias=True)

    def forward(self, x):
        return self.mlp(x)


class gcn(nn.Module):
    """gwnet.py:24-46.  `forward(x[B,C,V,T], [adp])` runs the HIP kernel."""

    def __init__(self, c_in, c_out, dropout, support_len=3, order=2):
        super().__init__()
        self.nconv = nconv()
        self.c_in, self.c_out = c_in, c_out
        self.mlp = linear((order * support_len + 1) * c_in, c_out)
        self.dropout = dropout
        self.order = order
        self.support_len = support_len

    def _check(self):
        if not (self.c_in == 64 and self.c_out == 64 and self.order == 2 and self.support_len == 1):
            raise NotImplementedError("hopmi gcn kernel: only c_in=c_out=64, order=2, one support (the HOP.py:143 "
                                      "configuration) is built")

    def forward_cl(self, x_cl, A1, A2, prep=None):
        """x_cl (B,T,V,64) channels-last -> (B,T,V,64)."""
        self._check()
        h = ops.gcn(x_cl, A1, A2, self.mlp.mlp.weight, self.mlp.mlp.bias, prep)
        return F.dropout(h, self.dropout, training=self.training) if self.dropout > 0 else h

    def forward(self, x, support):
        if len(support) != 1:
            raise NotImplementedError("hopmi gcn kernel: exactly one support matrix (the adaptive adjacency)")
        A = support[0]
        h = self.forward_cl(x.permute(0, 3, 2, 1), A, A @ A)
        return h.permute(0, 3, 2, 1).contiguous()


class gwnet(nn.Module):
    """gwnet.py:49-249 with the arguments HOP.py:143 passes."""

    def __init__(self, device, num_nodes, dropout=0.3, supports=None, gcn_bool=True, addaptadj=True, aptinit=None,
                 in_dim=2, out_dim=12, residual_channels=32, dilation_channels=32, skip_channels=256,
                 end_channels=512, kernel_size=2, blocks=4, layers=2):
        super().__init__()
        if not (gcn_bool and addaptadj and supports is None and aptinit is None and kernel_size == 2
                and blocks * layers == len(DILATIONS) and residual_channels == 64 and dilation_channels == 64):
            raise NotImplementedError("hopmi gwnet: only the HOP.py:143 configuration (adaptive adjacency only, "
                                      "64 residual/dilation channels, 4x2 layers, kernel 2) is built")
        self.dropout, self.blocks, self.layers = dropout, blocks, layers
        self.gcn_bool, self.addaptadj = gcn_bool, addaptadj
        self.filter_convs, self.gate_convs = nn.ModuleList(), nn.ModuleList()
        self.residual_convs, self.skip_convs = nn.ModuleList(), nn.ModuleList()
        self.bn, self.gconv = nn.ModuleList(), nn.ModuleList()
        self.start_conv = nn.Conv2d(in_dim, residual_channels, kernel_size=(1, 1))
        self.supports = []
        self.supports_len = 1
        self.nodevec1 = nn.Parameter(torch.randn(num_nodes, 10))
        self.nodevec2 = nn.Parameter(torch.randn(10, num_nodes))
        receptive_field = 1
        for _b in range(blocks):
            additional_scope, new_dilation = kernel_size - 1, 1
            for _l in range(layers):
                self.filter_convs.append(nn.Conv2d(residual_channels, dilation_channels, (1, kernel_size), dilation=new_dilation))
                self.gate_convs.append(nn.Conv2d(residual_channels, dilation_channels, (1, kernel_size), dilation=new_dilation))
                self.residual_convs.append(nn.Conv2d(dilation_channels, residual_channels, (1, 1)))   # never used (gwnet.py:224-231)
                self.skip_convs.append(nn.Conv2d(dilation_channels, skip_channels, (1, 1)))
                self.bn.append(nn.BatchNorm2d(residual_channels))
                new_dilation *= 2
                receptive_field += additional_scope
                additional_scope *= 2
                self.gconv.append(gcn(dilation_channels, residual_channels, dropout, support_len=self.supports_len))
        self.end_conv_1 = nn.Conv2d(skip_channels, end_channels, (1, 1), bias=True)
        self.end_conv_2 = nn.Conv2d(end_channels, out_dim, (1, 1), bias=True)
        self.receptive_field = receptive_field
        self._bn_keep = _Keep()
        self.num_nodes, self.in_dim, self.out_dim = num_nodes, in_dim, out_dim
        self.skip_channels, self.end_channels = skip_channels, end_channels

    # -- pieces -----------------------------------------------------------------------------
    def adjacency(self):
        """gwnet.py:161-164; A2 = A @ A re-associates (xA)A for the kernel (V x V, torch)."""
        A1 = F.softmax(F.relu(torch.mm(self.nodevec1, self.nodevec2)), dim=1)
        return A1, A1 @ A1

    def _batchnorm(self, i, y):
        """bn[i] on channels-last y (B,T,V,64): gwnet.py:237 (training: batch statistics over
        (B,V,T), running stats updated with momentum 0.1 / unbiased variance)."""
        bn = self.bn[i]
        if self.training:
            var, mean = torch.var_mean(y, dim=(0, 1, 2), unbiased=False)
            with torch.no_grad():
                n = y.numel() // y.shape[-1]
                bn.running_mean.mul_(1 - bn.momentum).add_(mean, alpha=bn.momentum)
                bn.running_var.mul_(1 - bn.momentum).add_(var * (n / max(n - 1, 1)), alpha=bn.momentum)
                bn.num_batches_tracked += 1
                # (the row replay_bn_update() re-applies when a later forward of the step reuses this one: mean | rstd | unbiased variance)
                self._bn_keep.items.append(torch.cat([mean, torch.rsqrt(var + bn.eps), var * (n / max(n - 1, 1))]).float())
        else:
            mean, var = bn.running_mean, bn.running_var
        scale = bn.weight * torch.rsqrt(var + bn.eps)
        return y * scale + (bn.bias - mean * scale)

    def _weight_images(self):
        """Split-bf16 MFMA weight images of all 8 layers (one launch per forward pass)."""
        return ops.wn_prepare_weights([(self.filter_convs[i].weight, self.gate_convs[i].weight, self.gconv[i].mlp.mlp.weight)
                                       for i in range(len(DILATIONS))])

    def _skip_tails_fused(self, x, prep, wimg):
        """The 8 WaveNet layers as 8 fused kernels (no autograd graph): x (B,T,V,64) start-conv output ->
        (B,4,V,8*64) gated activations of every layer's last 4 frames (all the skip path needs).
        Training mode uses batch statistics and advances the running statistics like nn.BatchNorm2d."""
        B, _, V, _ = x.shape
        tails = torch.empty(B, 4, V, 64 * len(DILATIONS), dtype=x.dtype, device=x.device)
        scsh = _identity_scsh(x.device)
        xin = x.contiguous()
        last = len(DILATIONS) - 1
        keep = self._bn_keep = _Keep()
        if self.training and ops.wn_stack_supported(B, xin.shape[1], V, DILATIONS):
            # training-mode statistics: one persistent launch for the 8 layers
            _, _, mr_rows = ops.wn_stack_fwd(xin, wimg, [(self.filter_convs[i].bias, self.gate_convs[i].bias) for i in range(last + 1)], prep,
                                             [self.gconv[i].mlp.mlp.bias for i in range(last + 1)], list(self.bn), tails, DILATIONS)
            keep.items.extend(mr_rows[i] for i in range(last + 1))
            self._count_batches()
            return tails
        for i, d in enumerate(DILATIONS):
            bn = self.bn[i]
            fc, gc = self.filter_convs[i], self.gate_convs[i]
            mlp = self.gconv[i].mlp.mlp
            # the last layer's gcn/BN output is dead (gwnet.py:240); in training the reference still
            # advances bn[7]'s running statistics, which needs y_7's batch statistics.
            do_gcn = (i != last) or self.training
            bnargs = (bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps) if self.training else None
            y, _, scsh_out, _ = ops.wn_layer_fwd(xin, scsh, wimg[i], fc.bias, gc.bias, prep, mlp.bias,
                                                 tails[..., 64 * i:64 * (i + 1)], d, want_y=(i != last),
                                                 do_gcn=do_gcn, bn=bnargs, stats_keep=keep.items)
            if i == last:
                break
            if self.training:
                scsh = scsh_out
            else:
                sc = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
                scsh = torch.cat([sc, bn.bias - bn.running_mean * sc])
            xin = y
        if self.training:
            self._count_batches()
        return tails

    def _count_batches(self):
        torch._foreach_add_([bn.num_batches_tracked for bn in self.bn], 1)

    def replay_bn_update(self):
        """Apply the running-statistics update of the most recent training-mode fused forward once more: what a
        second forward over the same input with unchanged weights does to the BatchNorm buffers (the kernels
        are bitwise reproducible, so its batch statistics would be the same numbers)."""
        if len(self._bn_keep.items) != len(self.bn):
            raise RuntimeError("hopmi gwnet: no training-mode fused forward to replay")
        for kept, bn in zip(self._bn_keep.items, self.bn):
            ops.wn_bn_replay(kept, bn)
        self._count_batches()

    def _tail(self, tails):
        """skip 1x1 convs summed over layers (one K=512 GEMM), relu, end convs: gwnet.py:209-220,240-246."""
        ws, bs = _SkipPack.apply(*[c.weight for c in self.skip_convs], *[c.bias for c in self.skip_convs])   # (256, 8*64), (256,)
        s = F.relu(ops.linear(tails, ws, bs))
        s = F.relu(ops.linear(s, self.end_conv_1.weight.flatten(1), self.end_conv_1.bias))
        return ops.linear(s, self.end_conv_2.weight.flatten(1), self.end_conv_2.bias)

    def forward_cl(self, x):
        """x (B,T>=13,V,in_dim) channels-last -> (B,4,V,out_dim) channels-last."""
        if x.shape[1] < self.receptive_field:
            x = F.pad(x, (0, 0, 0, 0, self.receptive_field - x.shape[1], 0))          # gwnet.py:145-146
        x = ops.linear(x, self.start_conv.weight.flatten(1), self.start_conv.bias)            # gwnet.py:149
        if x.dtype != torch.float32 and not (x.dtype == torch.bfloat16 and ops.WN_BF16_STORAGE):
            x = x.float()                                  # (ops.WN_BF16_STORAGE = False: fp32 storage also under bf16 autocast)
        with torch.autocast("cuda", enabled=False):        # the kernels compute in fp32 and take fp32 or bf16 activations
            A1, A2 = self.adjacency()
            prep = ops.gcn_prepare(A1, A2)      # on-chip images of the mix matrices, shared by all layers
            wimg = self._weight_images() if self.dropout == 0 else None
            tails = self._skip_tails_fused(x, prep, wimg) if (not torch.is_grad_enabled() and self.dropout == 0 and not ops.STRICT_FP32) else None
        if tails is not None:
            return self._tail(tails)
        if self.training and self.dropout == 0 and ops.wn_fused_training_supported(x.shape[2]) and not ops.STRICT_FP32:
            # differentiable fused stack: one forward and one backward kernel per WaveNet layer
            params = []
            for i in range(len(DILATIONS)):
                params += [self.filter_convs[i].weight, self.filter_convs[i].bias, self.gate_convs[i].weight, self.gate_convs[i].bias]
            for i in range(len(DILATIONS)):
                params += [self.gconv[i].mlp.mlp.weight, self.gconv[i].mlp.mlp.bias]
            for i in range(len(DILATIONS)):
                params += [self.bn[i].weight, self.bn[i].bias]
            keep = self._bn_keep = _Keep()
            tails = _WaveNetStackFn.apply(x, A1, A2, prep, wimg, list(self.bn), keep, *params)
            self._count_batches()
            return self._tail(tails)
        # eval-mode BatchNorm with autograd (fine-tuning with frozen statistics), graphs beyond the fused kernels' size, and
        # ops.strict_fp32: composed from the gcn kernel (exact-fp32 MFMA) and library GEMMs
        if self.training:
            self._bn_keep = _Keep()
        T_out = x.shape[1] - sum(DILATIONS)
        tails = []
        last = len(DILATIONS) - 1
        for i, d in enumerate(DILATIONS):
            Tn = x.shape[1] - d
            lo, hi = x[:, :Tn], x[:, d:]                    # Conv2d (1,2) taps: t and t+d (gwnet.py:186-200)
            wf, wg = self.filter_convs[i].weight, self.gate_convs[i].weight
            w0 = torch.cat([wf[:, :, 0, 0], wg[:, :, 0, 0]], 0)
            w1 = torch.cat([wf[:, :, 0, 1], wg[:, :, 0, 1]], 0)
            fg = F.linear(lo, w0) + F.linear(hi, w1, torch.cat([self.filter_convs[i].bias, self.gate_convs[i].bias]))
            u = torch.tanh(fg[..., :64]) * torch.sigmoid(fg[..., 64:])
            # only the last T_out frames of every layer's skip reach the output (crop-adds, gwnet.py:213-220)
            tails.append(u[:, Tn - T_out:])
            if i == last:
                # gcn/bn of the last layer never reach the output (gwnet.py:240); the reference still
                # advances bn[7]'s running statistics in training mode, so do that (no autograd).
                if self.training:
                    with torch.no_grad():
                        self._batchnorm(i, self.gconv[i].forward_cl(u, A1, A2, prep) + hi)
                break
            x = self._batchnorm(i, self.gconv[i].forward_cl(u, A1, A2, prep) + hi)           # gwnet.py:226-237
        return self._tail(torch.cat(tails, -1))

    def forward(self, input):
        """input (B,in_dim,V,T) NCHW (any strides) -> (B,out_dim,V,T-12) NCHW contiguous."""
        return self.forward_cl(input.permute(0, 3, 2, 1)).permute(0, 3, 2, 1).contiguous()
