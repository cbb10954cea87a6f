"""TEST INFRASTRUCTURE ONLY -- helpers shared by tools/make_golden.py and tests/.

Checksums (so fixtures stay KB-sized), the stand-in objects the reference's
constructor contract asks for (HOP.py:73-111), and the tiny-BERT configuration the
end-to-end fixtures use (HOP.Model takes the LLM as a constructor argument).
"""
import types

import numpy as np
import torch

from . import fill


def checksum(t: torch.Tensor) -> np.ndarray:
    """[sum, abs-sum, pseudo-random-weighted sum] in float64; the weighted sum catches
    permutations that the first two miss."""
    t64 = t.detach().to("cpu", torch.float64).contiguous()
    w = fill.uniform("checksum.w", (t64.numel(),)).to(torch.float64).view(t64.shape)
    return np.array([t64.sum().item(), t64.abs().sum().item(), (t64 * w).sum().item()], dtype=np.float64)


def checksum_close(got: np.ndarray, want: np.ndarray, rel: float, atol: float = 0.0) -> bool:
    """Compare two checksums relative to the abs-sum (the natural scale of all three).

    `atol` (default none) is for tensors that are analytically zero, where both sides hold only
    rounding noise: pass `zero_grad_atol(name)` for gradients."""
    scale = max(abs(want[1]), 1e-30)
    return bool(np.all(np.abs(got - want) <= rel * scale + atol))


def zero_grad_param(name: str) -> bool:
    """Parameters whose gradient is analytically zero: biases that feed straight into a training-mode
    BatchNorm (gwnet's graph-conv bias, the discriminator's pre_conv conv biases and its first BatchNorm's shift, which
    the next conv turns into a per-channel constant in front of the next BatchNorm) and the key-projection bias
    (softmax is invariant to a per-key bias).  What either implementation holds there is rounding noise
    (~1e-7), which Adam turns into +-lr steps of arbitrary sign."""
    zero = ("pre_conv.0.bias", "pre_conv.1.bias", "pre_conv.3.bias")      # (1: a BatchNorm shift in front of conv 3 -> BatchNorm 4)
    return (name.endswith("mlp.mlp.bias") or name in zero or name.split(".", 1)[-1] in zero or name.endswith("key_projection.bias"))


def zero_grad_atol(name: str) -> float:
    """Absolute slack of a gradient checksum: 1e-4 for the analytically-zero gradients, none otherwise."""
    return 1e-4 if zero_grad_param(str(name)) else 0.0


def grad_table(module: torch.nn.Module):
    """(names, (n,3) checksum table, names_without_grad) over requires_grad parameters."""
    names, rows, nograd = [], [], []
    for n, p in module.named_parameters():
        if not p.requires_grad:
            continue
        if p.grad is None:
            nograd.append(n)
        else:
            names.append(n)
            rows.append(checksum(p.grad))
    return names, np.stack(rows) if rows else np.zeros((0, 3)), nograd


def tiny_bert_config():
    from transformers import BertConfig
    return BertConfig(vocab_size=96, hidden_size=48, num_hidden_layers=2, num_attention_heads=4,
                      intermediate_size=96, max_position_embeddings=64,
                      hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)


class SynthTok:
    """Tokenizer stand-in: HOP.py:83-88 only touches these three members."""
    eos_token = None
    pad_token = None

    def add_special_tokens(self, d):
        return 0


class SynthVocab:
    """Speaker-vocabulary stand-in: HOP.py:103 reads n_words only."""

    def __init__(self, n_words):
        self.n_words = n_words


def hop_cfg(V: int, llm_dim: int):
    """The attribute bag HOP.Model reads (HOP.py:75-81,119,121; run_ted.py:74-79)."""
    return types.SimpleNamespace(d_ff=128, llm_dim=llm_dim, use_gwnet=True, use_reprograme=True,
                                 d_model=128, n_heads=8, datasets="TED" if V == 9 else "TED_expressive")


def step_args(V: int):
    """Loss weights of run_ted.py:89-92 / run_expressive.py:86-89."""
    if V == 9:
        return types.SimpleNamespace(loss_regression_weight=600.0, loss_gan_weight=5.0, loss_kld_weight=0.6,
                                     loss_reg_weight=0.4, z_type="speaker")
    return types.SimpleNamespace(loss_regression_weight=2100.0, loss_gan_weight=5.0, loss_kld_weight=0.8,
                                 loss_reg_weight=0.5, z_type="speaker")


class Accel:
    """accelerate.Accelerator stand-in: train_llm.py:34,85 only call .backward(loss)."""

    def backward(self, loss):
        loss.backward()
