"""CPU: the C-ABI library loads and exports what include/hopmi.h declares; host-side module
logic (state_dict layout, loud failure without a GPU).  No kernel is launched here."""
import ctypes
import os
import re

import pytest
import torch

import hopmi
from hopmi import _lib
from oracle import spec
from oracle.golden_util import SynthTok, SynthVocab, hop_cfg, tiny_bert_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "hopmi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hopmi_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    syms = header_symbols()
    assert "hopmi_gcn_fwd" in syms and "hopmi_gcn_bwd" in syms
    handle = ctypes.CDLL(_lib._LIB_PATH)
    for s in syms:
        assert hasattr(handle, s), f"{s} declared in hopmi.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in _lib.py"
    assert sorted(_lib.SIGNATURES) == syms
    assert _lib.version().startswith("hopmi")


def test_bad_arguments_are_rejected_without_a_gpu():
    L = _lib.lib()
    assert L.hopmi_gcn_fwd(None, None, None, None, None, 4, 9, None) == -1
    assert b"null pointer" in L.hopmi_last_error()
    assert L.hopmi_gcn_bwd_ws_floats(4, 9) == 64 * 192 + 64 + 2 * 81        # one workgroup tile
    assert L.hopmi_gcn_bwd_ws_floats(4, 49) == 0
    assert L.hopmi_gcn_prep_floats(9) == 12 * 48 + 20 * 16 and L.hopmi_gcn_prep_floats(49) == 0


@pytest.mark.parametrize("V", [9, 42])
def test_state_dict_layout_matches_reference(golden, V):
    from transformers import BertModel
    g = golden(f"model_V{V}")
    bcfg = tiny_bert_config()
    m = hopmi.Model(hop_cfg(V, bcfg.hidden_size), BertModel(bcfg), SynthTok(), SynthVocab(11))
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["state_keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g["state_shapes"]]
    trainable = {n for n, p in m.named_parameters() if p.requires_grad}
    assert not any(n.startswith("llm_model.") or n == "word_embeddings" for n in trainable)
    assert set(spec.model_spec(V, bcfg, 11).keys()) == set(sd.keys())


def test_hot_path_refuses_cpu_tensors():
    x = torch.zeros(2, 3, 9, 64)
    A = torch.eye(9)
    with pytest.raises(_lib.HopmiError, match="no CPU fallback"):
        hopmi.ops.gcn(x, A, A, torch.zeros(64, 192), torch.zeros(64))
    with pytest.raises(_lib.HopmiError, match="no CPU fallback"):
        hopmi.ops.gcn_prepare(A, A)


def test_discriminator_cpu_matches_golden(golden):
    """ConvDiscriminator stays on stock torch ops, so it can be checked on CPU too."""
    from oracle import fill
    g = golden("disc_P27")
    d = hopmi.ConvDiscriminator(27)
    d.gru.dropout = 0.0
    fill.fill_state_(d)
    d.train()
    y = d(fill.normal("disc.poses", (3, 34, 27), 0.3))
    assert torch.allclose(y, torch.from_numpy(g["out"]), rtol=1e-4, atol=1e-6)


def test_trimodal_api_matches_reference_golden(golden):
    """Secondary boundary: PoseGenerator.forward(pre_seq, in_text, in_audio, vid) and
    train_iter_gan(...) keep the reference's signatures and arithmetic (stock torch ops, so checked on CPU)."""
    import types
    from oracle import fill
    from oracle.golden_util import SynthVocab, checksum, checksum_close
    g = golden("trimodal")
    args = types.SimpleNamespace(n_pre_poses=4, n_poses=34, input_context="both", hidden_size=24, n_layers=2, dropout_prob=0.0,
                                 freeze_wordembed=False, loss_warmup=-1, loss_gan_weight=5.0, loss_regression_weight=600.0,
                                 loss_kld_weight=0.6, loss_reg_weight=0.4, z_type="speaker")
    B, P, n_words, n_spk = 2, 27, 40, 7
    gen = hopmi.PoseGenerator(args, P, n_words, 16, None, SynthVocab(n_spk))
    for mod in gen.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    assert list(gen.state_dict().keys()) == [str(k) for k in g["state_keys"]]
    fill.fill_state_(gen)
    d = hopmi.ConvDiscriminator(P)
    d.gru.dropout = 0.0
    fill.fill_state_(d, salt=1)
    gen.train(); d.train()
    text = fill.integers("gan.text", (B, 34), n_words)
    audio = fill.normal("gan.audio", (B, 36267))
    poses = fill.normal("gan.poses", (B, 34, P), 0.1)
    vid = fill.integers("gan.vid", (B,), n_spk)
    pre = poses.new_zeros(B, 34, P + 1)
    pre[:, :4, :-1] = poses[:, :4]
    pre[:, :4, -1] = 1
    torch.manual_seed(99)
    out, _, mu, _ = gen(pre, text, audio, vid)
    assert torch.allclose(out, torch.from_numpy(g["out"]), rtol=1e-4, atol=1e-6)
    assert torch.allclose(mu, torch.from_numpy(g["z_mu"]), rtol=1e-4, atol=1e-6)
    g_opt = torch.optim.Adam(gen.parameters(), lr=1e-3, betas=(0.5, 0.999))
    d_opt = torch.optim.Adam(d.parameters(), lr=1e-4, betas=(0.5, 0.999))
    torch.manual_seed(99)
    ret = hopmi.train_iter_gan(args, 0, text, audio, poses, vid, gen, d, g_opt, d_opt)
    assert sorted(ret.keys()) == [str(k) for k in g["ret_keys"]]
    for k, want in zip(g["ret_keys"], g["ret_vals"]):
        assert abs(ret[str(k)] - want) <= 2e-4 * max(abs(want), 1e-6), (k, ret[str(k)], want)
    sd = gen.state_dict()
    for n, want in zip(g["g_names"], g["g_cs"]):
        assert checksum_close(checksum(sd[str(n)]), want, 2e-4, 2.5e-3 * sd[str(n)].numel() if "bias" in str(n) else 1e-4), n


def test_host_switches_without_a_device():
    """Host-side switches of the package behave on a machine without a GPU: the GEMM table is not enabled, the
    precision switch validates its argument and round-trips, and the shipped table is a TunableOp file for gfx950."""
    import hopmi
    from hopmi import tuning
    if not torch.cuda.is_available():
        assert hopmi.use_tuned_gemms() is False
    assert os.path.isfile(tuning.DEFAULT_TABLE)
    head = open(tuning.DEFAULT_TABLE).read(400)
    assert "Validator,GCN_ARCH_NAME,gfx950" in head and "Validator,PT_VERSION" in head
    prev = hopmi.mixed_precision("bf16")
    try:
        assert hopmi.mixed_precision(None) == "bf16"
        with pytest.raises(ValueError):
            hopmi.mixed_precision("fp8")
    finally:
        hopmi.mixed_precision(prev)
