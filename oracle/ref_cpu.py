"""TEST INFRASTRUCTURE ONLY -- functional CPU restatement of the HOP generator hot path.

Every function takes a flat ``sd`` mapping (the reference's own ``state_dict`` key names
-> tensors) and plain tensors, uses only elementary torch ops (matmul / einsum / exp /
tanh ...), and cites the reference lines it restates (paths relative to the reference
checkout).  Tensors follow the *reference* layouts (NCHW for gwnet) so that the golden
vectors taken from the reference compare element for element.

Parity pin: ``tests/golden/*.npz`` made by ``tools/make_golden.py`` from the imported
reference.  This file is the checker for the HIP path and the CPU leg of
``bench.py``; it is never imported by the product package.
"""
import math
from typing import Dict, Optional, Tuple

import torch

SD = Dict[str, torch.Tensor]
GW_DILATIONS = (1, 2, 1, 2, 1, 2, 1, 2)          # gwnet.py:98-121 (blocks=4, layers=2)
BN_EPS = 1e-5                                     # nn.BatchNorm2d default, gwnet.py:120
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------- helpers
def linear(x, w, b=None):
    y = x.matmul(w.t())
    return y if b is None else y + b


def conv1x1_nchw(x, w, b):
    """1x1 Conv2d on (B,C,V,T): gwnet.py:16-22,65-67,117-119,129-137."""
    y = torch.einsum("bcvt,oc->bovt", x, w.reshape(w.shape[0], w.shape[1]))
    return y + b.view(1, -1, 1, 1)


def batchnorm_train(y, gamma, beta, dims, eps=BN_EPS):
    """Training-mode batch norm over `dims`; returns (out, batch_mean, unbiased_var)."""
    n = 1
    for d in dims:
        n *= y.shape[d]
    mean = y.mean(dim=dims, keepdim=True)
    var_b = ((y - mean) ** 2).mean(dim=dims, keepdim=True)           # biased: used to normalise
    shape = [1] * y.dim()
    shape[1] = -1
    out = (y - mean) / torch.sqrt(var_b + eps) * gamma.view(shape) + beta.view(shape)
    var_u = var_b * (n / max(n - 1, 1))                               # unbiased: feeds running_var
    return out, mean.flatten().detach(), var_u.flatten().detach()


def batchnorm_eval(y, gamma, beta, rmean, rvar, eps=BN_EPS):
    shape = [1] * y.dim()
    shape[1] = -1
    return (y - rmean.view(shape)) / torch.sqrt(rvar.view(shape) + eps) * gamma.view(shape) + beta.view(shape)


def layernorm(x, w, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


# ----------------------------------------------------------------------- graph wavenet
def adjacency(nodevec1, nodevec2):
    """gwnet.py:161-164: adp = softmax(relu(E1 @ E2), dim=1)."""
    return torch.softmax(torch.relu(nodevec1.matmul(nodevec2)), dim=1)


def nconv(x, A):
    """gwnet.py:12-14: einsum('ncvl,vw->ncwl')."""
    return torch.einsum("ncvl,vw->ncwl", x, A)


def gcn(x, A, w_mlp, b_mlp):
    """gwnet.py:33-46 with one support, order 2, dropout 0: h = Wm . [x; xA; xAA] + bm."""
    x1 = nconv(x, A)
    x2 = nconv(x1, A)
    return conv1x1_nchw(torch.cat([x, x1, x2], dim=1), w_mlp, b_mlp)


def gated_tcn(x, wf, bf, wg, bg, d):
    """gwnet.py:186-200. Conv2d kernel (1,2), dilation d: tap 0 reads t, tap 1 reads t+d."""
    T = x.shape[3]
    lo, hi = x[..., : T - d], x[..., d:]

    def conv(w, b):
        y = torch.einsum("bcvt,oc->bovt", lo, w[:, :, 0, 0]) + torch.einsum("bcvt,oc->bovt", hi, w[:, :, 0, 1])
        return y + b.view(1, -1, 1, 1)

    return torch.tanh(conv(wf, bf)) * torch.sigmoid(conv(wg, bg))


def gwnet_forward(sd: SD, x, prefix="gwnet.", training=True, relu_masks=None, relu_probe=None) -> Tuple[torch.Tensor, SD]:
    """gwnet.py:143-249 (gcn_bool, addaptadj, supports=[]).  x: (B,173,V,16) -> (B,173,V,4).

    Returns (out, bn_updates) where bn_updates holds the new running_mean / running_var /
    num_batches_tracked of all 8 BatchNorm2d layers (empty in eval mode).

    `relu_masks` = (mask_skip (B,256,V,4), mask_end (B,512,V,4)) of {0,1}: evaluate the two ReLUs of gwnet.py:240-242 as
    `x * mask` -- gradient checks at sizes with ~1e6 pre-activations would otherwise depend on which side of zero
    an implementation's rounding puts the few values that lie within 1e-7 of the kink.  `relu_probe` (a list) receives the
    oracle's own two pre-activations, so that a test can check that the masks it passed in differ from the oracle's sides
    only at the kink.
    """
    p = lambda n: sd[prefix + n]
    x = conv1x1_nchw(x, p("start_conv.weight"), p("start_conv.bias"))
    A = adjacency(p("nodevec1"), p("nodevec2"))
    skip = None
    updates: SD = {}
    for i, d in enumerate(GW_DILATIONS):
        residual = x
        u = gated_tcn(residual, p(f"filter_convs.{i}.weight"), p(f"filter_convs.{i}.bias"),
                      p(f"gate_convs.{i}.weight"), p(f"gate_convs.{i}.bias"), d)
        s = conv1x1_nchw(u, p(f"skip_convs.{i}.weight"), p(f"skip_convs.{i}.bias"))
        skip = s if skip is None else s + skip[..., -s.shape[3]:]          # gwnet.py:213-220
        h = gcn(u, A, p(f"gconv.{i}.mlp.mlp.weight"), p(f"gconv.{i}.mlp.mlp.bias"))
        y = h + residual[..., -h.shape[3]:]                                # gwnet.py:233
        if training:
            x, bmean, bvar = batchnorm_train(y, p(f"bn.{i}.weight"), p(f"bn.{i}.bias"), dims=(0, 2, 3))
            updates[prefix + f"bn.{i}.running_mean"] = (1 - BN_MOMENTUM) * p(f"bn.{i}.running_mean") + BN_MOMENTUM * bmean
            updates[prefix + f"bn.{i}.running_var"] = (1 - BN_MOMENTUM) * p(f"bn.{i}.running_var") + BN_MOMENTUM * bvar
            updates[prefix + f"bn.{i}.num_batches_tracked"] = p(f"bn.{i}.num_batches_tracked") + 1
        else:
            x = batchnorm_eval(y, p(f"bn.{i}.weight"), p(f"bn.{i}.bias"),
                               p(f"bn.{i}.running_mean"), p(f"bn.{i}.running_var"))
    relu1 = torch.relu if relu_masks is None else (lambda t: t * relu_masks[0].to(t.dtype))
    relu2 = torch.relu if relu_masks is None else (lambda t: t * relu_masks[1].to(t.dtype))
    x = relu1(skip)                                                        # gwnet.py:240
    pre2 = conv1x1_nchw(x, p("end_conv_1.weight"), p("end_conv_1.bias"))
    if relu_probe is not None:
        relu_probe += [skip.detach(), pre2.detach()]
    x = relu2(pre2)
    x = conv1x1_nchw(x, p("end_conv_2.weight"), p("end_conv_2.bias"))
    return x, updates


# ----------------------------------------------------------- reprogramming cross-attention
def reprogramming_layer(sd: SD, target, source, value, n_heads, prefix="reprogramming_layer.",
                        drop_mask: Optional[torch.Tensor] = None, p_drop: float = 0.0,
                        relu_mask: Optional[torch.Tensor] = None, relu_probe: Optional[list] = None):
    """HOP.py:271-299.  target (B,L,d_model); source/value (S,d_llm) -> (B,L,d_llm).

    `drop_mask` (B,H,L,S) of {0,1} reproduces nn.Dropout(p_drop) on the probabilities.  `relu_mask` (B,L,H*E) of
    {0,1} evaluates the ReLU of HOP.py:284 as `x * mask` (gradient checks with millions of pre-activations must not
    depend on which side of zero rounding puts the few values within 1e-7 of the kink); `relu_probe` (a list) receives
    the oracle's own pre-activation so that a test can check the mask against it.
    """
    p = lambda n: sd[prefix + n]
    B, L, _ = target.shape
    S = source.shape[0]
    q = linear(target, p("query_projection.weight"), p("query_projection.bias")).view(B, L, n_heads, -1)
    k = linear(source, p("key_projection.weight"), p("key_projection.bias")).view(S, n_heads, -1)
    v = linear(value, p("value_projection.weight"), p("value_projection.bias")).view(S, n_heads, -1)
    scale = 1.0 / math.sqrt(q.shape[-1])
    scores = torch.einsum("blhe,she->bhls", q, k)
    attn = torch.softmax(scale * scores, dim=-1)
    if drop_mask is not None:
        attn = attn * drop_mask / (1.0 - p_drop)
    out = torch.einsum("bhls,she->blhe", attn, v).reshape(B, L, -1)
    if relu_probe is not None:
        relu_probe.append(out.detach())
    act = torch.relu(out) if relu_mask is None else out * relu_mask.to(out.dtype)
    return linear(act, p("out_projection.weight"), p("out_projection.bias"))               # ReLU *before* out-proj


# ------------------------------------------------------------------------- frozen BERT
def bert_encoder(sd: SD, inputs_embeds, n_heads, prefix="llm_model.", eps=1e-12):
    """HF BertModel(inputs_embeds=...).last_hidden_state, dropout off (HOP.py:204).

    third-party transformers `modeling_bert.py` (BertEmbeddings / BertLayer); the
    arithmetic is pinned by the golden vector taken from the installed package.
    """
    p = lambda n: sd[prefix + n]
    B, L, D = inputs_embeds.shape
    h = inputs_embeds + p("embeddings.token_type_embeddings.weight")[0] + p("embeddings.position_embeddings.weight")[:L]
    h = layernorm(h, p("embeddings.LayerNorm.weight"), p("embeddings.LayerNorm.bias"), eps)
    dh = D // n_heads
    i = 0
    while (prefix + f"encoder.layer.{i}.attention.self.query.weight") in sd:
        lp = f"encoder.layer.{i}."
        q = linear(h, p(lp + "attention.self.query.weight"), p(lp + "attention.self.query.bias")).view(B, L, n_heads, dh)
        k = linear(h, p(lp + "attention.self.key.weight"), p(lp + "attention.self.key.bias")).view(B, L, n_heads, dh)
        v = linear(h, p(lp + "attention.self.value.weight"), p(lp + "attention.self.value.bias")).view(B, L, n_heads, dh)
        pr = torch.softmax(torch.einsum("blhe,bshe->bhls", q, k) / math.sqrt(dh), dim=-1)
        a = torch.einsum("bhls,bshe->blhe", pr, v).reshape(B, L, D)
        a = linear(a, p(lp + "attention.output.dense.weight"), p(lp + "attention.output.dense.bias"))
        h = layernorm(h + a, p(lp + "attention.output.LayerNorm.weight"), p(lp + "attention.output.LayerNorm.bias"), eps)
        f = gelu_erf(linear(h, p(lp + "intermediate.dense.weight"), p(lp + "intermediate.dense.bias")))
        f = linear(f, p(lp + "output.dense.weight"), p(lp + "output.dense.bias"))
        h = layernorm(h + f, p(lp + "output.LayerNorm.weight"), p(lp + "output.LayerNorm.bias"), eps)
        i += 1
    return h


# --------------------------------------------------------------------------------- GRU
def gru_bidir(sd: SD, x, prefix, num_layers, hidden, drop_masks=None, p_drop=0.0):
    """torch.nn.GRU(batch_first=True, bidirectional=True) as explicit cell arithmetic.

    HOP.py:166-167,248 (decoder) and multimodal_context_net.py:236-237,257 (discriminator).
    Gate order r,z,n; n = tanh(W_in x + b_in + r*(W_hn h + b_hn)); h' = (1-z)*n + z*h.
    """
    B, T, _ = x.shape
    inp = x
    for layer in range(num_layers):
        outs = []
        for sfx, rev in (("", False), ("_reverse", True)):
            w_ih, w_hh = sd[f"{prefix}weight_ih_l{layer}{sfx}"], sd[f"{prefix}weight_hh_l{layer}{sfx}"]
            b_ih, b_hh = sd[f"{prefix}bias_ih_l{layer}{sfx}"], sd[f"{prefix}bias_hh_l{layer}{sfx}"]
            gi_all = linear(inp, w_ih, b_ih)                       # (B,T,3H): input GEMM batched over time
            h = x.new_zeros(B, hidden)
            seq = [None] * T
            for t in (range(T - 1, -1, -1) if rev else range(T)):
                gi = gi_all[:, t]
                gh = linear(h, w_hh, b_hh)
                r = torch.sigmoid(gi[:, :hidden] + gh[:, :hidden])
                z = torch.sigmoid(gi[:, hidden:2 * hidden] + gh[:, hidden:2 * hidden])
                n = torch.tanh(gi[:, 2 * hidden:] + r * gh[:, 2 * hidden:])
                h = (1.0 - z) * n + z * h
                seq[t] = h
            outs.append(torch.stack(seq, dim=1))
        inp = torch.cat(outs, dim=2)
        if drop_masks is not None and layer < num_layers - 1:
            inp = inp * drop_masks[layer] / (1.0 - p_drop)
    return inp


# ---------------------------------------------------------------------- HOP.Model.forward
def audio_windows(in_audio):
    """HOP.py:210: unfold(1, 3400, 2191) -> (B,16,3400)."""
    return in_audio.unfold(1, 3400, 2191)


def beat_features(sd: SD, in_audio):
    """HOP.py:130-134,210-211 de-duplicated over the V-fold repeat: (B,16,170)."""
    w = audio_windows(in_audio)
    h = linear(w, sd["beat.0.weight"], sd["beat.0.bias"])
    h = torch.where(h >= 0, h, 0.2 * h)
    return linear(h, sd["beat.2.weight"], sd["beat.2.bias"])


def gwnet_input(pre_seq, feat, V):
    """HOP.py:212-217.  audio_feat[b,t,j] = F[b,(t*V+j)%16] (the .view scramble); returns
    the (B,173,V,16) NCHW tensor x0[b,c,j,t] = c<3 ? pre_seq[b,t,3j+c] : audio_feat[b,t,j,c-3]."""
    B = pre_seq.shape[0]
    t = torch.arange(16).view(16, 1)
    j = torch.arange(V).view(1, V)
    widx = ((t * V + j) % 16).to(feat.device)                      # (16,V)
    audio_feat = feat[:, widx]                                     # (B,16,V,170)
    seq = torch.cat([pre_seq.reshape(B, 16, V, 3), audio_feat], dim=3)
    return seq.permute(0, 3, 2, 1)


def model_forward(sd: SD, cfg, in_audio, x_enc, text, pre_seq, vid_indices, eps,
                  training=True, bert_heads=12) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, SD]:
    """HOP.py:181-252 (use_gwnet and use_reprograme both on, speaker z).

    `eps` (B,16) is the N(0,1) draw of embedding_net.py:10-13.  Returns
    (dec_out, z, z_mu, z_logvar, bn_updates).
    """
    B = pre_seq.shape[0]
    V = pre_seq.shape[2] // 3
    # speaker VAE, HOP.py:184-190
    zc = linear(sd["speaker_embedding.0.weight"][vid_indices], sd["speaker_embedding.1.weight"], sd["speaker_embedding.1.bias"])
    z_mu = linear(zc, sd["speaker_mu.weight"], sd["speaker_mu.bias"])
    z_logvar = linear(zc, sd["speaker_logvar.weight"], sd["speaker_logvar.bias"])
    z = z_mu + eps * torch.exp(0.5 * z_logvar)
    # text + reprogramming + LLM, HOP.py:198-204
    E = sd["word_embeddings"]
    text_emb = E[text.long()]
    S = sd["mapping_layer.weight"].matmul(E) + sd["mapping_layer.bias"].view(-1, 1)
    enc = reprogramming_layer(sd, x_enc, S, S, cfg.n_heads)
    llm_in = linear(torch.cat([enc, text_emb], dim=2), sd["align_layer.weight"], sd["align_layer.bias"])
    dec = bert_encoder(sd, llm_in, bert_heads)
    # audio windows + gwnet, HOP.py:210-219
    feat = beat_features(sd, in_audio)
    feature, bn_updates = gwnet_forward(sd, gwnet_input(pre_seq, feat, V), training=training)
    # HOP.py:221-231
    g_seq = feature[:, :3].reshape(B, 3 * V, 4).permute(0, 2, 1)          # channel-major xyz
    beat = feature[:, 3:].reshape(B, 34, -1)                              # raw reinterpretation
    pre = feature.new_zeros(B, 34, 3 * V + 1)
    pre[:, :4, :-1] = g_seq
    pre[:, :4, -1] = 1
    dec_in = torch.cat([pre, beat, dec, z.unsqueeze(1).expand(B, 34, z.shape[1])], dim=2)
    # decoder, HOP.py:248-250 ; LeakyReLU(True) has slope 1.0 == identity
    g = gru_bidir(sd, dec_in, "gru.", 4, 350)
    g = g[:, :, :350] + g[:, :, 350:]
    out = linear(linear(g, sd["out.0.weight"], sd["out.0.bias"]), sd["out.3.weight"], sd["out.3.bias"])
    return out, z, z_mu, z_logvar, bn_updates


# ------------------------------------------------------------------ ConvDiscriminator
def conv1d(x, w, b):
    """Valid Conv1d, stride 1: x (B,Cin,T), w (Cout,Cin,K)."""
    K = w.shape[2]
    T = x.shape[2] - K + 1
    y = sum(torch.einsum("bct,oc->bot", x[:, :, k:k + T], w[:, :, k]) for k in range(K))
    return y + b.view(1, -1, 1)


def conv_discriminator(sd: SD, poses, training=True, prefix="") -> Tuple[torch.Tensor, SD]:
    """multimodal_context_net.py:219-268 (GRU dropout forced to 0 for parity).

    LeakyReLU(True) is the identity.  Returns (sigmoid score (B,1), bn_updates).
    """
    p = lambda n: sd[prefix + n]
    x = poses.transpose(1, 2)
    updates: SD = {}
    for ci, bi in ((0, 1), (3, 4)):
        x = conv1d(x, p(f"pre_conv.{ci}.weight"), p(f"pre_conv.{ci}.bias"))
        if training:
            x, bm, bv = batchnorm_train(x, p(f"pre_conv.{bi}.weight"), p(f"pre_conv.{bi}.bias"), dims=(0, 2))
            updates[prefix + f"pre_conv.{bi}.running_mean"] = 0.9 * p(f"pre_conv.{bi}.running_mean") + 0.1 * bm
            updates[prefix + f"pre_conv.{bi}.running_var"] = 0.9 * p(f"pre_conv.{bi}.running_var") + 0.1 * bv
            updates[prefix + f"pre_conv.{bi}.num_batches_tracked"] = p(f"pre_conv.{bi}.num_batches_tracked") + 1
        else:
            x = batchnorm_eval(x, p(f"pre_conv.{bi}.weight"), p(f"pre_conv.{bi}.bias"),
                               p(f"pre_conv.{bi}.running_mean"), p(f"pre_conv.{bi}.running_var"))
    x = conv1d(x, p("pre_conv.6.weight"), p("pre_conv.6.bias")).transpose(1, 2)          # (B,28,8)
    g = gru_bidir(sd, x, prefix + "gru.", 4, 64)
    g = g[:, :, :64] + g[:, :, 64:]
    o = linear(g, p("out.weight"), p("out.bias")).reshape(poses.shape[0], -1)            # (B,28)
    return torch.sigmoid(linear(o, p("out2.weight"), p("out2.bias"))), updates


# ------------------------------------------------------------------------ losses / step
def smooth_l1(a, b):
    d = (a - b).abs()
    return torch.where(d < 1.0, 0.5 * d * d, d - 0.5)


def generator_losses(args, epoch, outputs, target, z, z_mu, z_logvar, out_rand, z_rand, dis_output):
    """train_llm.py:43-82.  Returns (loss, huber, kld, div_reg, gen_error)."""
    gen_error = -torch.mean(torch.log(dis_output + 1e-8))
    huber = smooth_l1(outputs / 0.1, target / 0.1).mean() * 0.1
    beta = 0.05
    pose_l1 = (smooth_l1(outputs / beta, out_rand.detach() / beta) * beta).sum(dim=1).sum(dim=1)
    z_l1 = (z.detach() - z_rand.detach()).abs().mean(1)
    div_reg = torch.clamp(-(pose_l1 / (z_l1 + 1.0e-5)), min=-1000).mean()
    kld = -0.5 * torch.mean(1 + z_logvar - z_mu.pow(2) - z_logvar.exp())
    loss = huber * args.loss_regression_weight + div_reg * args.loss_reg_weight + kld * args.loss_kld_weight
    if epoch > 10:
        loss = loss + gen_error * args.loss_gan_weight
    return loss, huber, kld, div_reg, gen_error


def train_llm_step(args, cfg, epoch, batch, g_sd: SD, d_sd: SD, g_optim, d_optim, rng, bert_heads=12):
    """train_llm.py:9-98 on functional state (z_type == 'speaker', loss_reg_weight > 0).

    `g_sd` / `d_sd`: name -> tensor, trainable entries are leaf tensors with
    requires_grad=True that the caller-owned optimisers hold.  `rng(kind, shape)`
    supplies the random draws in the reference's order: 'eps' (B,16) per generator
    forward, 'noise' for add_noise, 'perm' for randperm.  BN running stats are written
    back into the dicts like nn.BatchNorm does.
    """
    in_audio, mel, text, target, vid = (batch[k] for k in ("in_audio", "log_melspec", "text", "target_dir_vec", "vid_indices"))
    pre_seq = target[:, 0:16]
    B = target.shape[0]

    def G(vids):
        out, z, mu, lv, upd = model_forward(g_sd, cfg, in_audio, mel, text, pre_seq, vids, rng("eps", (B, 16)),
                                            training=True, bert_heads=bert_heads)
        with torch.no_grad():
            for k, v in upd.items():
                g_sd[k] = v
        return out, z, mu, lv

    def D(x):
        y, upd = conv_discriminator(d_sd, x, training=True)
        with torch.no_grad():
            for k, v in upd.items():
                d_sd[k] = v
        return y

    dis_error = None
    gan = epoch > 10 and args.loss_gan_weight > 0.0
    if gan:                                                             # train_llm.py:15-36
        d_optim.zero_grad()
        outputs, *_ = G(vid)
        noise_target = target + rng("noise", target.shape) * 0.1
        noise_out = outputs.detach() + rng("noise", outputs.shape) * 0.1
        dis_real, dis_fake = D(noise_target), D(noise_out)
        dis_error = torch.sum(-torch.mean(torch.log(dis_real + 1e-8) + torch.log(1 - dis_fake + 1e-8)))
        dis_error.backward()
        d_optim.step()
    g_optim.zero_grad()
    outputs, z, z_mu, z_logvar = G(vid)
    dis_output = D(outputs)
    rand_vids = vid[rng("perm", (B,))]
    out_rand, z_rand, _, _ = G(rand_vids)
    loss, huber, kld, div_reg, gen_error = generator_losses(args, epoch, outputs, target, z, z_mu, z_logvar,
                                                            out_rand, z_rand, dis_output)
    loss.backward()
    g_optim.step()
    ret = {"loss": args.loss_regression_weight * huber.item()}
    if kld:
        ret["KLD"] = args.loss_kld_weight * kld.item()
    if div_reg:
        ret["DIV_REG"] = args.loss_reg_weight * div_reg.item()
    if gan:
        ret["gen"] = args.loss_gan_weight * gen_error.item()
        ret["dis"] = dis_error.item()
    return ret, outputs.detach(), z_mu.detach(), z_logvar.detach(), out_rand.detach()
