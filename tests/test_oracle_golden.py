"""CPU: the oracle restatement (oracle/ref_cpu.py) against the golden vectors that
tools/make_golden.py took from the real reference.  Tolerance: fp32 re-association
only (1e-5 relative to the tensor's scale), far inside the 1e-3 the HIP path is held to."""
import numpy as np
import pytest
import torch

from oracle import fill, ref_cpu, spec
from oracle.golden_util import checksum, checksum_close, hop_cfg, step_args, tiny_bert_config

RTOL = 2e-5


def close(got, want, rtol=RTOL):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    scale = max(np.abs(want).max(), 1e-30)
    err = np.abs(got - want).max() / scale
    assert err <= rtol, f"max err {err:.3e} (scale {scale:.3e})"


def grads_close(sd, names, table, rtol=1e-4):
    for n, want in zip(names, table):
        g = sd[str(n)].grad
        assert g is not None, n
        assert checksum_close(checksum(g), want, rtol, zero_grad_atol(n)), (n, checksum(g), want)


def require_grad(sd, pred):
    for k, v in sd.items():
        if v.is_floating_point() and pred(k) and "running_" not in k:
            v.requires_grad_(True)


@pytest.mark.parametrize("V", [9, 42])
def test_state_dict_spec_matches_reference(golden, V):
    g = golden(f"model_V{V}")
    s = spec.model_spec(V, tiny_bert_config(), 11)
    assert list(s.keys()) == [str(k) for k in g["state_keys"]]
    for (k, shp), want in zip(s.items(), g["state_shapes"]):
        assert (str(tuple(shp)) if shp != spec.LONG else "()") == str(want), k


@pytest.mark.parametrize("V", [9, 42])
def test_gcn(golden, V):
    g = golden(f"gcn_V{V}")
    sd = spec.build_sd(spec.gcn_spec())
    x = fill.uniform("gcn.x", (2, 64, V, 5)).requires_grad_()
    A = ref_cpu.adjacency(fill.uniform("gcn.nodevec1", (V, 10)), fill.uniform("gcn.nodevec2", (10, V))).requires_grad_()
    w, b = sd["mlp.mlp.weight"].requires_grad_(), sd["mlp.mlp.bias"].requires_grad_()
    h = ref_cpu.gcn(x, A, w, b)
    close(h, g["h"])
    (h * fill.uniform("gcn.gout", h.shape)).sum().backward()
    close(x.grad, g["dx"]); close(A.grad, g["dA"]); close(w.grad, g["dW"]); close(b.grad, g["db"])


@pytest.mark.parametrize("V", [9, 42])
@pytest.mark.parametrize("training", [True, False])
def test_gwnet(golden, V, training):
    g = golden(f"gwnet_V{V}_{'train' if training else 'eval'}")
    sd = spec.build_sd(spec.gwnet_spec(V, prefix=""))
    require_grad(sd, lambda k: True)
    x0 = fill.uniform("gwnet.x0", (2, 173, V, 16)).requires_grad_()
    out, upd = ref_cpu.gwnet_forward(sd, x0, prefix="", training=training)
    close(out, g["out"])
    (out * fill.uniform("gwnet.gout", out.shape)).sum().backward()
    assert checksum_close(checksum(x0.grad), g["dx0_cs"], 1e-4)
    close(x0.grad.flatten()[::97], g["dx0_sample"], 1e-4)
    grads_close(sd, g["grad_names"], g["grad_cs"])
    for n in g["nograd_names"]:
        assert sd[str(n)].grad is None or float(sd[str(n)].grad.abs().sum()) == 0.0, n
    for i in range(8):
        rm = upd[f"bn.{i}.running_mean"] if training else sd[f"bn.{i}.running_mean"]
        rv = upd[f"bn.{i}.running_var"] if training else sd[f"bn.{i}.running_var"]
        close(rm, g[f"bn{i}_rm"]); close(rv, g[f"bn{i}_rv"])


@pytest.mark.parametrize("tag,B,S,d_llm", [("tiny", 2, 50, 48), ("real", 1, 1500, 768)])
def test_reprogramming(golden, tag, B, S, d_llm):
    g = golden(f"reprog_{tag}")
    sd = spec.build_sd(spec.reprog_spec(d_llm, prefix=""))
    require_grad(sd, lambda k: True)
    tgt = fill.normal("reprog.target", (B, 34, 128)).requires_grad_()
    src = fill.uniform("reprog.source", (S, d_llm), 0.5).requires_grad_()
    out = ref_cpu.reprogramming_layer(sd, tgt, src, src, 8, prefix="")
    close(out, g["out"])
    (out * fill.uniform("reprog.gout", out.shape)).sum().backward()
    close(tgt.grad, g["dtarget"], 1e-4)
    assert checksum_close(checksum(src.grad), g["dsource_cs"], 1e-4)
    close(src.grad.flatten()[::53], g["dsource_sample"], 1e-4)
    grads_close(sd, g["grad_names"], g["grad_cs"])


@pytest.mark.parametrize("tag", ["tiny", "base2"])
def test_bert(golden, tag):
    from transformers import BertConfig
    g = golden(f"bert_{tag}")
    cfg = tiny_bert_config() if tag == "tiny" else BertConfig(num_hidden_layers=2)
    B = 2 if tag == "tiny" else 1
    sd = spec.build_sd(spec.bert_spec(cfg, prefix=""))
    x = fill.uniform("bert.inputs_embeds", (B, 34, cfg.hidden_size)).requires_grad_()
    out = ref_cpu.bert_encoder(sd, x, cfg.num_attention_heads, prefix="")
    close(out, g["out"])
    (out * fill.uniform("bert.gout", out.shape)).sum().backward()
    close(x.grad, g["dx"], 1e-4)


def _model_sd(V):
    sd = spec.build_sd(spec.model_spec(V, tiny_bert_config(), 11))
    require_grad(sd, lambda k: not k.startswith("llm_model.") and k != "word_embeddings")
    return sd


@pytest.mark.parametrize("V", [9, 42])
def test_model_forward_backward(golden, V):
    g = golden(f"model_V{V}")
    bcfg = tiny_bert_config()
    sd = _model_sd(V)
    inp = fill.hot_path_inputs(2, V, bcfg.vocab_size, 11)
    torch.manual_seed(4321)
    eps = torch.randn(2, 16)
    out, z, mu, lv, upd = ref_cpu.model_forward(sd, hop_cfg(V, bcfg.hidden_size), inp["in_audio"], inp["log_melspec"],
                                                inp["text"], inp["target_dir_vec"][:, :16], inp["vid_indices"], eps,
                                                training=True, bert_heads=bcfg.num_attention_heads)
    close(out, g["out"]); close(z, g["z"]); close(mu, g["z_mu"]); close(lv, g["z_logvar"])
    ((out * fill.uniform("model.gout", out.shape)).sum() + 0.3 * z.sum() + 0.1 * (mu * mu).sum() + 0.2 * lv.exp().sum()).backward()
    grads_close(sd, g["grad_names"], g["grad_cs"], 2e-4)
    for n in g["nograd_names"]:
        assert sd[str(n)].grad is None, n
    for i in range(8):
        close(upd[f"gwnet.bn.{i}.running_mean"], g[f"bn{i}_rm"]); close(upd[f"gwnet.bn.{i}.running_var"], g[f"bn{i}_rv"])


@pytest.mark.parametrize("V", [9, 42])
def test_model_eval(golden, V):
    g = golden(f"model_V{V}_eval")
    bcfg = tiny_bert_config()
    sd = spec.build_sd(spec.model_spec(V, bcfg, 11))
    inp = fill.hot_path_inputs(2, V, bcfg.vocab_size, 11)
    torch.manual_seed(4321)
    eps = torch.randn(2, 16)
    with torch.no_grad():
        out, *_ = ref_cpu.model_forward(sd, hop_cfg(V, bcfg.hidden_size), inp["in_audio"], inp["log_melspec"], inp["text"],
                                        inp["target_dir_vec"][:, :16], inp["vid_indices"], eps, training=False,
                                        bert_heads=bcfg.num_attention_heads)
    close(out, g["out"])


@pytest.mark.parametrize("P", [27, 126])
def test_discriminator(golden, P):
    g = golden(f"disc_P{P}")
    sd = spec.build_sd(spec.disc_spec(P))
    require_grad(sd, lambda k: True)
    x = fill.normal("disc.poses", (3, 34, P), 0.3).requires_grad_()
    y, upd = ref_cpu.conv_discriminator(sd, x, training=True)
    close(y, g["out"])
    torch.log(y + 1e-8).sum().backward()
    close(x.grad, g["dx"], 1e-4)
    grads_close(sd, g["grad_names"], g["grad_cs"])
    close(upd["pre_conv.1.running_mean"], g["bn1_rm"]); close(upd["pre_conv.1.running_var"], g["bn1_rv"])


from oracle.golden_util import zero_grad_param, zero_grad_atol      # noqa: E402


def cpu_rng(kind, shape):
    return torch.randperm(shape[0]) if kind == "perm" else torch.randn(shape)


@pytest.mark.parametrize("V", [9, 42])
@pytest.mark.parametrize("epoch", [0, 11])
def test_train_llm_step(golden, V, epoch):
    g = golden(f"train_llm_V{V}_e{epoch}")
    bcfg = tiny_bert_config()
    g_sd = _model_sd(V)
    d_sd = spec.build_sd(spec.disc_spec(3 * V), salt=1)
    require_grad(d_sd, lambda k: True)
    g_opt = torch.optim.Adam([v for v in g_sd.values() if v.requires_grad], lr=1e-3, betas=(0.5, 0.999))
    d_opt = torch.optim.Adam([v for v in d_sd.values() if v.requires_grad], lr=1e-4, betas=(0.5, 0.999))
    inp = fill.hot_path_inputs(2, V, bcfg.vocab_size, 11)
    torch.manual_seed(777)
    ret, *_ = ref_cpu.train_llm_step(step_args(V), hop_cfg(V, bcfg.hidden_size), epoch, inp, g_sd, d_sd, g_opt, d_opt,
                                     cpu_rng, bert_heads=bcfg.num_attention_heads)
    assert sorted(ret.keys()) == [str(k) for k in g["ret_keys"]]
    for k, want in zip(g["ret_keys"], g["ret_vals"]):
        assert abs(ret[str(k)] - want) <= 2e-4 * max(abs(want), 1e-6), (k, ret[str(k)], want)
    for names, table, sd, lr in ((g["g_names"], g["g_cs"], g_sd, 1e-3), (g["d_names"], g["d_cs"], d_sd, 1e-4)):
        for n, want in zip(names, table):
            atol = 2.5 * lr * sd[str(n)].numel() if zero_grad_param(str(n)) else 0.0
            assert checksum_close(checksum(sd[str(n)]), want, 2e-4, atol), (n, checksum(sd[str(n)]), want)
