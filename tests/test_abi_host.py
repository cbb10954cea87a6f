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
    assert L.hopmi_gcn_prep_floats(9) == 12 * 48 + 20 * 16 + 4 and L.hopmi_gcn_prep_floats(49) == 0


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


def test_new_entry_points_check_their_arguments_without_a_gpu():
    """hopmi_colsum and the dtype argument of the reprogramming attention validate sizes / pointers / dtype before any launch."""
    L = _lib.lib()
    assert L.hopmi_colsum_ws_floats(128, 64) == 0 and L.hopmi_colsum_ws_floats(129, 64) == 2 * 64 and L.hopmi_colsum_ws_floats(0, 64) == 0
    assert L.hopmi_colsum(None, 0, 4, 4, None, None, None) == -1 and b"null pointer" in L.hopmi_last_error()
    assert L.hopmi_colsum(None, 0, 0, 4, None, None, None) == -1 and b"bad sizes" in L.hopmi_last_error()
    assert L.hopmi_colsum(None, 7, 4, 4, None, None, None) == -1 and b"dtype" in L.hopmi_last_error()
    one = ctypes.c_void_p(16)                       # (never dereferenced: the checks come first)
    assert L.hopmi_reprog_attn_fwd_dt(one, one, one, one, 2, one, one, 4, 4, 8, 128, 0.1, 0.0, 0, None, None) == -1
    assert b"dtype" in L.hopmi_last_error()
    assert L.hopmi_reprog_attn_fwd_dt(one, one, one, one, 1, one, one, 4, 4, 8, 64, 0.1, 0.0, 0, None, None) == -1
    assert b"head dim 128" in L.hopmi_last_error()


def test_linear_bias_gradient_has_no_host_path():
    """ops.linear is F.linear in the forward (any device); its bias gradient is hopmi_colsum, which host tensors cannot
    reach: the backward raises instead of summing with torch."""
    x = torch.randn(5, 7, requires_grad=True)
    w = torch.randn(3, 7, requires_grad=True)
    b = torch.randn(3, requires_grad=True)
    y = hopmi.ops.linear(x, w, b)
    assert torch.equal(y, torch.nn.functional.linear(x, w, b))
    with pytest.raises(_lib.HopmiError, match="no CPU fallback"):
        y.sum().backward()
    with torch.no_grad():                            # no gradient wanted: plain F.linear
        assert torch.equal(hopmi.ops.linear(x, w, b), torch.nn.functional.linear(x, w, b))
    lin = hopmi.ops.Linear(7, 3)
    assert isinstance(lin, torch.nn.Linear) and sorted(lin.state_dict()) == ["bias", "weight"]


def test_discriminator_and_generator_have_no_host_path():
    """ConvDiscriminator and Model run their recurrences / graph-wavenet block in libhopmi only: host tensors raise
    instead of taking a stock-torch path (the golden comparisons of both live in the -m gpu tests)."""
    from oracle import fill
    d = hopmi.ConvDiscriminator(27)
    with pytest.raises(_lib.HopmiError, match="no CPU fallback"):
        d(fill.normal("disc.poses", (3, 34, 27), 0.3))


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


def test_skip_pack_gradients_equal_cat_stack_composition():
    """gwnet._SkipPack (the eight skip convs' weights side by side, biases summed, through one autograd function) against the
    tensor-operation recipe it replaces (torch.cat / torch.stack + sum): same packed tensors, same gradients for every
    parameter, bit for bit (pure torch: runs on the host)."""
    import torch
    import importlib
    import hopmi  # noqa: F401
    gwnet = importlib.import_module("hopmi.gwnet")              # (the package exports the class under the same name)
    g = torch.Generator().manual_seed(5)
    ws = [torch.randn(256, 64, 1, 1, generator=g, requires_grad=True) for _ in range(8)]
    bs = [torch.randn(256, generator=g, requires_grad=True) for _ in range(8)]
    gw, gb = torch.randn(256, 512, generator=g), torch.randn(256, generator=g)
    pw, pb = gwnet._SkipPack.apply(*ws, *bs)
    rw = torch.cat([w.flatten(1) for w in ws], 1)
    rb = torch.stack(bs).sum(0)
    assert torch.equal(pw, rw) and torch.equal(pb, rb)
    got = torch.autograd.grad([pw, pb], ws + bs, [gw, gb])
    want = torch.autograd.grad([rw, rb], ws + bs, [gw, gb])
    for a, b, p in zip(got, want, ws + bs):
        assert a.shape == p.shape and a.is_contiguous() and torch.equal(a, b)


def test_diagnostic_library_is_the_same_abi_plus_the_split_status_setters():
    """`make dbg` (csrc/Makefile `all`): libhopmi_dbg.so exports every symbol of hopmi.h AND hopmi_debug_set_split_status_<file> for
    the eight translation units that split operands into fp16 hi/lo parts; the production library exports none of those."""
    dbg = os.path.join(os.path.dirname(_lib._LIB_PATH), "libhopmi_dbg.so")
    assert os.path.exists(dbg), f"{dbg} is missing (make -C csrc all)"
    d, p = ctypes.CDLL(dbg), ctypes.CDLL(_lib._LIB_PATH)
    for s in header_symbols():
        assert hasattr(d, s), s
    for name in ("gemm", "gemm_tn", "elementwise", "attn", "bert_attn", "gru", "wavenet", "wavenet_stack"):
        assert hasattr(d, "hopmi_debug_set_split_status_" + name), name
        assert not hasattr(p, "hopmi_debug_set_split_status_" + name), name


def test_one_invalidation_point_for_the_derived_operand_caches():
    """ops.reset_all_caches is the only place the cache tables are cleared: scope "recording" keeps the epoch and the frozen images made
    by eager calls, scope "all" (= invalidate_weight_images) moves CACHE_EPOCH and clears the frozen table; registered resetters see
    the scope; the constants table survives both."""
    from hopmi import ops
    seen = []
    ops.register_cache_resetter(seen.append)
    try:
        ops._CAST_CACHE["x"] = 1
        ops._F16_IMG["x"] = 1
        ops._F16_IMG_FROZEN["eager"] = (None, None, None, False)
        ops._F16_IMG_FROZEN["captured"] = (None, None, None, True)
        ops._UNIT_RS["c"] = 1
        e0 = ops.CACHE_EPOCH
        ops.cast_cache_reset()
        assert ops.CACHE_EPOCH == e0 and not ops._CAST_CACHE and not ops._F16_IMG
        assert "eager" in ops._F16_IMG_FROZEN and "captured" not in ops._F16_IMG_FROZEN and "c" in ops._UNIT_RS
        ops.invalidate_weight_images()
        assert ops.CACHE_EPOCH == e0 + 1 and not ops._F16_IMG_FROZEN and "c" in ops._UNIT_RS
        assert seen == ["recording", "all"]
        with pytest.raises(ValueError):
            ops.reset_all_caches("some")
    finally:
        ops._EXTRA_RESETTERS.remove(seen.append) if seen.append in ops._EXTRA_RESETTERS else ops._EXTRA_RESETTERS.pop()
        ops._UNIT_RS.pop("c", None)
