"""Fast path for the frozen BERT the reference passes into `HOP.Model` (HOP.py:90-91,204; built at
run_ted.py:177-195 as HF `BertModel` truncated to 6 layers).

`FrozenBertEncoder(llm)(inputs_embeds)` computes `llm(inputs_embeds=...).last_hidden_state` from the module's own
parameters with the same arithmetic (HF modeling_bert: absolute position + token-type-0 embeddings, LayerNorm
eps from the config, post-LN encoder layers, erf-GELU, no attention mask), but
  * Q, K, V projections are one GEMM (N = 3*hidden) instead of three,
  * the attention of one (clip, head) is one workgroup of a HIP kernel that reads the fused projection output and
    writes the output-projection input in place of transposes + library attention (ops.bert_attention; head dim 64),
  * bias + dropout + residual + LayerNorm and bias + GELU are single HIP kernels (ops.bias_*),
  * the pooler (unused by HOP) and the output_attentions / output_hidden_states tuples are skipped,
  * no parameter gradients are produced (the LLM is frozen); activation gradients flow as usual.
Dropout follows the module's train/eval state with the config's probabilities (the reference leaves the LLM in
train mode).  If the module is not BERT-shaped (the reference also supports LLaMA, run_ted.py:133-175) `supports()`
is False and the caller invokes the module itself.
"""
import os

import torch
import torch.nn.functional as F

from . import ops


# under bf16 autocast: the bf16-storage operators between the GEMMs (False: the fp32-storage operators + autocast's casts)
BF16_STORAGE = True
# the activation of BertIntermediate and its gradient as epilogues of the split GEMMs around them (False: launches of their own)
FUSED_FFN = os.environ.get("HOPMI_FUSED_FFN", "1") != "0"
# rows (tokens) from which the frozen linears go through hopmi_gemm_split
SPLIT_MIN_ROWS = 1024


def supports(llm) -> bool:
    try:
        cfg = llm.config
        lay = llm.encoder.layer[0]
        _ = (llm.embeddings.position_embeddings.weight, llm.embeddings.token_type_embeddings.weight,
             llm.embeddings.LayerNorm.weight, lay.attention.self.query.weight, lay.attention.output.dense.weight,
             lay.attention.output.LayerNorm.weight, lay.intermediate.dense.weight, lay.output.dense.weight,
             lay.output.LayerNorm.weight)
        return (getattr(cfg, "hidden_act", None) == "gelu" and getattr(cfg, "position_embedding_type", "absolute") == "absolute"
                and cfg.hidden_size % 4 == 0 and cfg.hidden_size <= 1024 and cfg.intermediate_size % 4 == 0
                and not any(p.requires_grad for p in llm.parameters()))
    except (AttributeError, IndexError, TypeError):
        return False


class FrozenBertEncoder:
    _calls = 0

    def __init__(self, llm):
        self.llm = llm
        self._qkv = {}
        self._img = {}
        self._bounds = {}                        # layer -> (weight versions, norm bounds of the FFN weights)
        self._w16 = {}
        # (the caches below carry ops.CACHE_EPOCH in their keys: ops.reset_all_caches("all") / invalidate_weight_images() retires them)

    def _fused_qkv(self, i, att):
        ver = (att.query.weight._version, att.key.weight._version, att.value.weight._version, att.query.weight.data_ptr(),
               att.query.bias._version, att.key.bias._version, att.value.bias._version, ops.CACHE_EPOCH)
        hit = self._qkv.get(i)

        def build():
            with torch.no_grad():
                return (torch.cat([att.query.weight, att.key.weight, att.value.weight], 0).contiguous(),
                        torch.cat([att.query.bias, att.key.bias, att.value.bias], 0).contiguous())

        if hit is None or hit[0] != ver:
            hit = (ver,) + build()
            self._qkv[i] = hit
        elif ops._checking():
            w, b = build()
            ops._check_equal("FrozenBertEncoder fused QKV weight", hit[1], w)
            ops._check_equal("FrozenBertEncoder fused QKV bias", hit[2], b)
        return hit[1], hit[2]

    def _linear(self, key, x, weight, bias, rowmax=None):
        """x W^T (+ bias) for one of the module's frozen linears: the library's fp32 GEMM, or -- ops.gemm_parts(2 | 3) --
        hopmi_gemm_split on images of W and W^T that are built once per weight version.  `rowmax` (a list, fp16 form): receives
        the product's partial row maxima."""
        imgs = self._images(key, x, weight)
        if imgs is None:
            return F.linear(x, weight, bias)
        N, K = weight.shape
        return ops.split_linear(x, imgs[0], imgs[1], bias, N, K, ops.GEMM_PARTS, rowmax=rowmax)

    def _images(self, key, x, weight):
        """Part images of a frozen weight and of its transpose (built once per weight version), or None where the linear
        goes to the library instead."""
        parts = ops.GEMM_PARTS
        N, K = weight.shape
        # (below ~1 000 rows -- inference windows, tiny batches -- the 128 x 128 tiles leave the chip empty and the library's
        # small-M kernels win: a 34-row window forward took 2.4 ms with the split GEMMs against 1.7 ms without)
        if (parts == 0 or torch.is_autocast_enabled() or x.numel() // x.shape[-1] < SPLIT_MIN_ROWS
                or not ops.split_gemm_supported(N, K) or not ops.split_gemm_supported(K, N)):
            return None
        ver = (weight._version, weight.data_ptr(), parts, ops.CACHE_EPOCH)
        hit = self._img.get(key)
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                hit = (ver, ops.split_weight_image(weight, parts), ops.split_weight_image(weight.t().contiguous(), parts))
            self._img[key] = hit
        elif ops._checking():
            with torch.no_grad():
                ops._check_equal(f"FrozenBertEncoder weight image {key}", hit[1], ops.split_weight_image(weight, parts))
                ops._check_equal(f"FrozenBertEncoder transposed weight image {key}", hit[2], ops.split_weight_image(weight.t().contiguous(), parts))
        return hit[1], hit[2]

    def _ffn_bounds(self, i, lay):
        """(largest row 2-norm of W1, largest |b1|, largest column 2-norm of W2) of the frozen FFN weights, as Python floats (one
        host read per weight version, in the first -- eager -- call): what the image-emitting GEMM epilogues bound their outputs
        with (ops._SplitFfnFn)."""
        w1, b1, w2 = lay.intermediate.dense.weight, lay.intermediate.dense.bias, lay.output.dense.weight
        ver = (w1._version, w1.data_ptr(), b1._version, w2._version, w2.data_ptr(), ops.CACHE_EPOCH)
        hit = self._bounds.get(i)

        def measure():
            with torch.no_grad():
                t = torch.stack([w1.float().norm(dim=1).max(), b1.float().abs().max(), w2.float().norm(dim=0).max()]).tolist()
            return (t[0] * 1.000001, t[1], t[2] * 1.000001)

        if hit is None or hit[0] != ver:
            if w1.is_cuda and torch.cuda.is_current_stream_capturing():
                # never read back under capture.  A bound of ANOTHER weight version is not a bound: without a current one the FFN
                # takes the split form (row scales from the data) for this recording
                return None
            hit = (ver, measure())
            self._bounds[i] = hit
        elif ops._checking():
            ops._check_count()
            if measure() != hit[1]:
                ops._check_fail(f"FrozenBertEncoder FFN norm bounds of layer {i}", f"cached {hit[1]} against {measure()}")
        return hit[1]

    def _ffn(self, i, lay, h):
        """gelu(h W1^T + b1) W2^T of layer i (BertIntermediate + BertOutput.dense without its bias)."""
        w1, w2 = lay.intermediate.dense.weight, lay.output.dense.weight
        i1, i2 = self._images((i, "f1"), h, w1), self._images((i, "f2"), h, w2)
        if FUSED_FFN and i1 is not None and i2 is not None:
            return ops.split_ffn(h, i1[0], i1[1], lay.intermediate.dense.bias, i2[0], i2[1], w1.shape[0], w1.shape[1], ops.GEMM_PARTS,
                                 bounds=self._ffn_bounds(i, lay))
        f = ops.bias_gelu(self._linear((i, "f1"), h, w1, None), lay.intermediate.dense.bias)
        return self._linear((i, "f2"), f, w2, None)

    def _bf16(self, key, weight, bias=None):
        """bf16 copies of a frozen weight (and bias), made once per weight version: under autocast the library would
        otherwise re-cast every weight in every step."""
        ver = (weight._version, weight.data_ptr(), None if bias is None else bias._version, ops.CACHE_EPOCH)
        hit = self._w16.get(key)
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                hit = (ver, weight.detach().to(torch.bfloat16).contiguous(), None if bias is None else bias.detach().to(torch.bfloat16).contiguous())
            self._w16[key] = hit
        return hit[1], hit[2]

    def _call_bf16(self, inputs_embeds):
        """The bf16-autocast form: bf16 library GEMMs against cached bf16 weights, and the bf16-storage forms of the HIP
        operators in between (ops.*_bf16: fp32 arithmetic, bf16 in / out), the residual stream in fp32 -- the values of the
        fp32-storage operators under autocast, without the cast kernels around every operator."""
        llm, cfg = self.llm, self.llm.config
        B, L, D = inputs_embeds.shape
        H = cfg.num_attention_heads
        train = llm.training
        p_h = cfg.hidden_dropout_prob if train else 0.0
        p_a = cfg.attention_probs_dropout_prob if train else 0.0
        emb = llm.embeddings
        h = F.layer_norm(inputs_embeds.float() + (emb.token_type_embeddings.weight[0] + emb.position_embeddings.weight[:L]),
                         (D,), emb.LayerNorm.weight, emb.LayerNorm.bias, cfg.layer_norm_eps).float()
        if p_h > 0:
            h = F.dropout(h, p_h, True)
        h16 = h.to(torch.bfloat16)
        for i, lay in enumerate(llm.encoder.layer):
            att = lay.attention
            wqkv, bqkv = self._fused_qkv(i, att.self)
            w16, b16 = self._bf16((i, "qkv"), wqkv, bqkv)
            qkv = F.linear(h16, w16, b16).view(B, L, 3, H, D // H)
            a16 = ops.bert_attention_bf16(qkv, p_a, self._seed())
            o16 = F.linear(a16, self._bf16((i, "ao"), att.output.dense.weight)[0])
            h, h16 = ops.bias_dropout_residual_layernorm_bf16(o16, att.output.dense.bias, h, att.output.LayerNorm.weight,
                                                              att.output.LayerNorm.bias, cfg.layer_norm_eps, p_h, self._seed())
            f16 = ops.bias_gelu_bf16(F.linear(h16, self._bf16((i, "f1"), lay.intermediate.dense.weight)[0]), lay.intermediate.dense.bias)
            o16 = F.linear(f16, self._bf16((i, "f2"), lay.output.dense.weight)[0])
            h, h16 = ops.bias_dropout_residual_layernorm_bf16(o16, lay.output.dense.bias, h, lay.output.LayerNorm.weight,
                                                              lay.output.LayerNorm.bias, cfg.layer_norm_eps, p_h, self._seed())
        return h

    def _seed(self):
        FrozenBertEncoder._calls += 1
        return (ops.base_seed() * 2246822519 + FrozenBertEncoder._calls * 3266489917) & 0xFFFFFFFF

    def __call__(self, inputs_embeds):
        llm, cfg = self.llm, self.llm.config
        B, L, D = inputs_embeds.shape
        H = cfg.num_attention_heads
        if (BF16_STORAGE and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
                and D // H == 64 and L <= 64):
            return self._call_bf16(inputs_embeds)
        train = llm.training
        p_h = cfg.hidden_dropout_prob if train else 0.0
        p_a = cfg.attention_probs_dropout_prob if train else 0.0
        emb = llm.embeddings
        # BertEmbeddings: LayerNorm(inputs_embeds + token_type[0] + position[:L]) then dropout
        h = F.layer_norm(inputs_embeds + (emb.token_type_embeddings.weight[0] + emb.position_embeddings.weight[:L]),
                         (D,), emb.LayerNorm.weight, emb.LayerNorm.bias, cfg.layer_norm_eps)
        if p_h > 0:
            h = F.dropout(h, p_h, True)
        hr = h                                               # the residual stream's handle on h
        for i, lay in enumerate(llm.encoder.layer):
            att = lay.attention
            wqkv, bqkv = self._fused_qkv(i, att.self)
            # (fp16 form, enough rows for the LDS-DMA path: the QKV product leaves its partial row maxima, from which the attention
            # kernel scales the operand image it writes for the attention-output product -- no image pass in between)
            rm = [] if (ops.ATTN_IMG and ops.IMG_FUSED and ops.GEMM_PARTS == ops.F16_PARTS and B * L >= ops.IMG_MIN_ROWS
                        and D // H == 64 and L <= 64 and D % 128 == 0) else None
            qkv = self._linear((i, "qkv"), h, wqkv, bqkv, rowmax=rm).view(B, L, 3, H, D // H)
            if D // H == 64 and L <= 64:
                a = ops.bert_attention(qkv, p_a, self._seed(), v_rowmax=rm[0] if rm else None)   # (B,L,D), HIP
            else:                                                                                # other head sizes
                qkv = qkv.permute(2, 0, 3, 1, 4)                                                 # (3,B,H,L,dh)
                a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], dropout_p=p_a)
                a = a.transpose(1, 2).reshape(B, L, D)
            o = self._linear((i, "ao"), a, att.output.dense.weight, None)
            # (every LayerNorm output feeds a GEMM and the next residual add: handed out twice, so that the two gradients
            # meet inside the backward kernel instead of in an add launch of their own)
            h, hr = ops.bias_dropout_residual_layernorm2(o, att.output.dense.bias, hr, att.output.LayerNorm.weight,
                                                         att.output.LayerNorm.bias, cfg.layer_norm_eps, p_h, self._seed())
            o = self._ffn(i, lay, h)
            h, hr = ops.bias_dropout_residual_layernorm2(o, lay.output.dense.bias, hr, lay.output.LayerNorm.weight,
                                                         lay.output.LayerNorm.bias, cfg.layer_norm_eps, p_h, self._seed())
        return h
