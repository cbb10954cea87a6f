"""Synthetic workload on the step's input contract: shapes / dtypes of the
reference's collate (lmdb_data_loader.py:47-62) without its LMDB/librosa pipeline, plus the
stand-ins `HOP.Model`'s constructor contract asks for (HOP.py:73-111)."""
import types

import torch


class SyntheticTokenizer:
    """HOP.py:83-88 touches eos_token / pad_token / add_special_tokens only."""
    eos_token = None
    pad_token = None

    def add_special_tokens(self, d):
        return 0


class SpeakerVocab:
    """HOP.py:103 reads n_words (TED train split has 1369 videos, data_preprocessor.py:29)."""

    def __init__(self, n_words=1370):
        self.n_words = n_words


def model_configs(datasets="TED", llm_dim=768):
    """run_ted.py:74-79,95-96."""
    return types.SimpleNamespace(d_model=128, n_heads=8, d_ff=128, llm_dim=llm_dim, use_gwnet=True,
                                 use_reprograme=True, datasets=datasets)


def step_args(datasets="TED"):
    """run_ted.py:89-92 / run_expressive.py:86-89."""
    if datasets == "TED":
        return types.SimpleNamespace(loss_regression_weight=600.0, loss_gan_weight=5.0, loss_kld_weight=0.6,
                                     loss_reg_weight=0.4, z_type="speaker")
    return types.SimpleNamespace(loss_regression_weight=2100.0, loss_gan_weight=5.0, loss_kld_weight=0.8,
                                 loss_reg_weight=0.5, z_type="speaker")


def build_bert(num_layers=6, **overrides):
    """run_ted.py:177-195: BERT-base geometry, first `num_layers` layers, random init (no network)."""
    from transformers import BertConfig, BertModel
    return BertModel(BertConfig(num_hidden_layers=num_layers, **overrides))


def synthetic_batch(B, V, seed, device, vocab=30522, n_spk=1370):
    g = torch.Generator(device="cpu").manual_seed(seed)
    batch = dict(
        in_audio=torch.randn(B, 36267, generator=g),
        log_melspec=torch.randn(B, 34, 128, generator=g),
        text=torch.randint(0, vocab, (B, 34), generator=g),
        target_dir_vec=torch.randn(B, 34, 3 * V, generator=g) * 0.1,
        vid_indices=torch.randint(0, n_spk, (B,), generator=g),
    )
    return {k: v.to(device) for k, v in batch.items()}
