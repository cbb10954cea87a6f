"""TEST INFRASTRUCTURE ONLY -- state_dict layout (key -> shape) of the reference modules.

Written out by hand from the reference constructors so the oracle can build its
parameter dicts without the reference being importable (GPU box).  Checked against the
real reference's `state_dict()` keys/shapes stored in tests/golden/model_V*.npz.
"""
from collections import OrderedDict

import torch

from . import fill

LONG = "int64"


def gcn_spec(prefix=""):
    """gwnet.py:24-31: gcn(64,64,support_len=1) -> mlp.mlp = Conv2d(192,64,1)."""
    return OrderedDict([(prefix + "mlp.mlp.weight", (64, 192, 1, 1)), (prefix + "mlp.mlp.bias", (64,))])


def gwnet_spec(V, prefix="gwnet."):
    """gwnet.py:50-137 with the HOP.py:143 arguments."""
    s = OrderedDict()
    s[prefix + "nodevec1"] = (V, 10)
    s[prefix + "nodevec2"] = (10, V)
    for grp, shape in (("filter_convs", (64, 64, 1, 2)), ("gate_convs", (64, 64, 1, 2)),
                       ("residual_convs", (64, 64, 1, 1)), ("skip_convs", (256, 64, 1, 1))):
        for i in range(8):
            s[f"{prefix}{grp}.{i}.weight"] = shape
            s[f"{prefix}{grp}.{i}.bias"] = (shape[0],)
    for i in range(8):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s[f"{prefix}bn.{i}.{leaf}"] = (64,)
        s[f"{prefix}bn.{i}.num_batches_tracked"] = LONG
    for i in range(8):
        s.update(gcn_spec(f"{prefix}gconv.{i}."))
    s[prefix + "start_conv.weight"] = (64, 173, 1, 1)
    s[prefix + "start_conv.bias"] = (64,)
    s[prefix + "end_conv_1.weight"] = (512, 256, 1, 1)
    s[prefix + "end_conv_1.bias"] = (512,)
    s[prefix + "end_conv_2.weight"] = (173, 512, 1, 1)
    s[prefix + "end_conv_2.bias"] = (173,)
    return s


def reprog_spec(d_llm, d_model=128, n_heads=8, d_keys=128, prefix="reprogramming_layer."):
    """HOP.py:256-265."""
    hk = n_heads * d_keys
    s = OrderedDict()
    for n, (o, i) in (("query_projection", (hk, d_model)), ("key_projection", (hk, d_llm)),
                      ("value_projection", (hk, d_llm)), ("out_projection", (d_llm, hk))):
        s[f"{prefix}{n}.weight"] = (o, i)
        s[f"{prefix}{n}.bias"] = (o,)
    return s


def bert_spec(cfg, prefix="llm_model."):
    """HF BertModel (transformers 5.x: no position_ids buffer in the state_dict)."""
    D, F = cfg.hidden_size, cfg.intermediate_size
    s = OrderedDict()
    s[prefix + "embeddings.word_embeddings.weight"] = (cfg.vocab_size, D)
    s[prefix + "embeddings.position_embeddings.weight"] = (cfg.max_position_embeddings, D)
    s[prefix + "embeddings.token_type_embeddings.weight"] = (cfg.type_vocab_size, D)
    s[prefix + "embeddings.LayerNorm.weight"] = (D,)
    s[prefix + "embeddings.LayerNorm.bias"] = (D,)
    for i in range(cfg.num_hidden_layers):
        lp = f"{prefix}encoder.layer.{i}."
        for n, (o, k) in (("attention.self.query", (D, D)), ("attention.self.key", (D, D)),
                          ("attention.self.value", (D, D)), ("attention.output.dense", (D, D))):
            s[lp + n + ".weight"] = (o, k)
            s[lp + n + ".bias"] = (o,)
        s[lp + "attention.output.LayerNorm.weight"] = (D,)
        s[lp + "attention.output.LayerNorm.bias"] = (D,)
        s[lp + "intermediate.dense.weight"] = (F, D)
        s[lp + "intermediate.dense.bias"] = (F,)
        s[lp + "output.dense.weight"] = (D, F)
        s[lp + "output.dense.bias"] = (D,)
        s[lp + "output.LayerNorm.weight"] = (D,)
        s[lp + "output.LayerNorm.bias"] = (D,)
    s[prefix + "pooler.dense.weight"] = (D, D)
    s[prefix + "pooler.dense.bias"] = (D,)
    return s


def wav_encoder_spec(prefix="audio_encoder."):
    """HOP.py:50-64 (never trained on the path, but part of the checkpoint layout)."""
    s = OrderedDict()
    convs = {0: (16, 1, 15), 3: (32, 16, 15), 6: (64, 32, 15), 9: (32, 64, 15)}
    bns = {1: 16, 4: 32, 7: 64}
    for i in range(10):
        if i in convs:
            s[f"{prefix}feat_extractor.{i}.weight"] = convs[i]
            s[f"{prefix}feat_extractor.{i}.bias"] = (convs[i][0],)
        elif i in bns:
            for leaf in ("weight", "bias", "running_mean", "running_var"):
                s[f"{prefix}feat_extractor.{i}.{leaf}"] = (bns[i],)
            s[f"{prefix}feat_extractor.{i}.num_batches_tracked"] = LONG
    return s


def gru_spec(prefix, in_size, hidden, layers=4):
    s = OrderedDict()
    for l in range(layers):
        k = in_size if l == 0 else 2 * hidden
        for sfx in ("", "_reverse"):
            s[f"{prefix}weight_ih_l{l}{sfx}"] = (3 * hidden, k)
            s[f"{prefix}weight_hh_l{l}{sfx}"] = (3 * hidden, hidden)
            s[f"{prefix}bias_ih_l{l}{sfx}"] = (3 * hidden,)
            s[f"{prefix}bias_hh_l{l}{sfx}"] = (3 * hidden,)
    return s


def model_spec(V, bert_cfg, n_spk):
    """HOP.py:73-175 in registration order (matches the reference's state_dict order)."""
    D = bert_cfg.hidden_size
    P = 3 * V
    s = OrderedDict()
    s["word_embeddings"] = (bert_cfg.vocab_size, D)
    s.update(bert_spec(bert_cfg))
    s.update(wav_encoder_spec())
    s["speaker_embedding.0.weight"] = (n_spk, 16)
    s["speaker_embedding.1.weight"] = (16, 16)
    s["speaker_embedding.1.bias"] = (16,)
    for n in ("speaker_mu", "speaker_logvar"):
        s[n + ".weight"] = (16, 16)
        s[n + ".bias"] = (16,)
    s["mapping_layer.weight"] = (1500, bert_cfg.vocab_size)
    s["mapping_layer.bias"] = (1500,)
    s["align_layer.weight"] = (D, 2 * D)
    s["align_layer.bias"] = (D,)
    s.update(reprog_spec(D))
    s["beat.0.weight"] = (1700, 3400)
    s["beat.0.bias"] = (1700,)
    s["beat.2.weight"] = (170, 1700)
    s["beat.2.bias"] = (170,)
    s.update(gwnet_spec(V))
    s.update(gru_spec("gru.", D + P + 1 + 16 + 20 * V, 350))
    s["out.0.weight"] = (175, 350)
    s["out.0.bias"] = (175,)
    s["out.3.weight"] = (P, 175)
    s["out.3.bias"] = (P,)
    return s


def disc_spec(P, prefix=""):
    """multimodal_context_net.py:220-239."""
    s = OrderedDict()
    for i, (co, ci) in ((0, (16, P)), (3, (8, 16)), (6, (8, 8))):
        s[f"{prefix}pre_conv.{i}.weight"] = (co, ci, 3)
        s[f"{prefix}pre_conv.{i}.bias"] = (co,)
    for i, c in ((1, 16), (4, 8)):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s[f"{prefix}pre_conv.{i}.{leaf}"] = (c,)
        s[f"{prefix}pre_conv.{i}.num_batches_tracked"] = LONG
    s.update(gru_spec(prefix + "gru.", 8, 64))
    s[prefix + "out.weight"] = (1, 64)
    s[prefix + "out.bias"] = (1,)
    s[prefix + "out2.weight"] = (1, 28)
    s[prefix + "out2.bias"] = (1,)
    return s


def build_sd(spec, salt=0, gains=None, aliases=(("word_embeddings", "llm_model.embeddings.word_embeddings.weight"),)):
    """Fill a spec exactly like fill.fill_state_ fills a module (sorted-key order, so an
    aliased tensor ends up with the value of its last key)."""
    sd = {}
    for name in sorted(spec.keys()):
        shape = spec[name]
        if shape == LONG:
            sd[name] = torch.zeros((), dtype=torch.int64)
        else:
            sd[name] = fill.fill_value(name, shape, salt, gains)
    for a, b in aliases:                       # HOP.py:111: one tensor under two keys
        if a in sd and b in sd:
            last = max(a, b)
            sd[a] = sd[b] = sd[last]
    return sd
