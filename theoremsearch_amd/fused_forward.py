"""Fused forwards of the three encoder families the reference embeds with: the GEMMs stay
in the BLAS library (north_star: the encoder forward is PyTorch-ROCm), everything around them is kernels of libtsearch.

* `FusedBertForward`   - BERT family: math-similarity/Bert-MLM_arXiv-MP-class_zbMath (compare_embeddings.py:11-12,
  app_create_embeddings.py:8, parsed_papers_to_vector_rds/embeddings.py:29)
* `FusedQwen3Forward`  - Qwen/Qwen3-Embedding-0.6B, the production embedder (streamlit_app.py:55, ec2/generate_embeddings/embedders.py:1-4)
* `FusedGemma3Forward` - google/embeddinggemma-300m, the `gemma` alias of ec2/generate_embeddings/embedders.py:1-4

Each keeps the model's weights (stacked projections are rebuilt when a parameter changes), its order of operations and the
roundings of the modules it replaces; `covers(model)` says whether a loaded model can take the fused path, `SentenceEncoder`
(encoder.py) asks.  Tested against the models' own forwards in tests/test_mirrors_gpu.py.
"""
from __future__ import annotations

import os
from typing import Optional

import torch


def split_pieces(x: torch.Tensor, pattern: int) -> torch.Tensor:
    """``x`` fp32 ``[rows x k]`` -> bf16 ``[rows x 3k]``: ``[hi | lo | hi]`` (pattern 0, activations) or ``[hi | hi | lo]``
    (pattern 1, weights), hi = bf16(x), lo = bf16(x - hi) (``ts_split_pieces``)."""
    import ctypes as C
    from . import _ffi
    x = x.contiguous()
    rows, k = x.shape
    out = torch.empty((rows, 3 * k), dtype=torch.bfloat16, device=x.device)
    _ffi.check(_ffi.load().ts_split_pieces(x.device.index or 0, C.c_void_p(x.data_ptr()), rows, k, pattern, C.c_void_p(out.data_ptr()),
                                           C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
    return out


def pieces_linear(x: torch.Tensor, w_pieces: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``F.linear(x, w, bias)`` for fp32 ``x`` and an fp32 weight given as its pieces (``split_pieces(w, 1)``), computed on the
    bf16 matrix pipe: ONE bf16 GEMM with fp32 accumulation over the three-fold depth (hipBLASLt through ``torch.mm(..,
    out_dtype=float32)``) = x_hi w_hi + x_lo w_hi + x_hi w_lo - the fp32 product up to ~2^-17 |x||w| per term, at a third of
    the bf16 GEMM rate instead of the fp32 matrix rate (157 TF against 2.5 PF dense on MI355X).  The encoder forward at the
    reference's fp32 storage (SentenceTransformer(name) without a dtype, streamlit_app.py:55,173) with ``fp32_gemm="bf16x3"``."""
    shape = x.shape
    x3 = split_pieces(x.reshape(-1, shape[-1]), 0)
    if bias is not None:
        y = torch.addmm(bias, x3, w_pieces.t(), out_dtype=torch.float32)
    else:
        y = torch.mm(x3, w_pieces.t(), out_dtype=torch.float32)
    return y.view(*shape[:-1], w_pieces.shape[0])


def pieces_mm(x_pieces: torch.Tensor, w_pieces: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The GEMM of `pieces_linear` on operands that already are pieces (``[rows x 3k]`` bf16 each): fp32 ``[rows x n]``."""
    if bias is not None:
        return torch.addmm(bias, x_pieces, w_pieces.t(), out_dtype=torch.float32)
    return torch.mm(x_pieces, w_pieces.t(), out_dtype=torch.float32)


def act_pieces(x: torch.Tensor, kind: int, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 activation straight into pieces (``ts_act_pieces``): kind 0 = gelu (erf) of ``[rows x n]``; 1 = silu(gate) * up,
    2 = gelu_tanh(gate) * up of ``[rows x 2n]``; ``bias`` (over the input's width) is added first.  Returns bf16 ``[rows x 3n]``."""
    import ctypes as C
    from . import _ffi
    x = x.contiguous()
    rows = x.numel() // x.shape[-1]
    n = x.shape[-1] if kind == 0 else x.shape[-1] // 2
    out = torch.empty((rows, 3 * n), dtype=torch.bfloat16, device=x.device)
    _ffi.check(_ffi.load().ts_act_pieces(x.device.index or 0, C.c_void_p(x.data_ptr()), C.c_void_p(bias.data_ptr()) if bias is not None else None,
                                         rows, n, kind, C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
    return out


def attention_float(qkv: torch.Tensor, key_mask: Optional[torch.Tensor], B: int, S: int, hq: int, hkv: int, hd: int, causal: bool,
                    scale: float, want_pieces: bool = False, bias: Optional[torch.Tensor] = None, want_context: bool = True):
    """fp32 attention of at most 128 tokens on the exact-fp32 matrix instructions, straight from the stacked projection's output
    ``qkv [B x S x (hq + 2 hkv) hd]`` (``ts_attention_float``; ``bias``: the projection's bias when its GEMM ran without one):
    ``(context fp32 [B x S x hq hd] - None with ``want_context=False`` - , its bf16 pieces or None)``."""
    import ctypes as C
    from . import _ffi
    qkv = qkv.contiguous()
    ctx = torch.empty((B, S, hq * hd), dtype=torch.float32, device=qkv.device) if (want_context or not want_pieces) else None
    pieces = torch.empty((B * S, 3 * hq * hd), dtype=torch.bfloat16, device=qkv.device) if want_pieces else None
    _ffi.check(_ffi.load().ts_attention_float(
        qkv.device.index or 0, C.c_void_p(qkv.data_ptr()), C.c_void_p(bias.data_ptr()) if bias is not None else None,
        C.c_void_p(key_mask.data_ptr()) if key_mask is not None else None, B, S, hq,
        hkv, hd, 1 if causal else 0, float(scale), C.c_void_p(ctx.data_ptr()) if ctx is not None else None,
        C.c_void_p(pieces.data_ptr()) if pieces is not None else None,
        C.c_void_p(torch.cuda.current_stream(qkv.device).cuda_stream)))
    return ctx, pieces


_FLOAT_ATTENTION_MAX_SEQ = {64: 512, 128: 256, 256: 128}      # what fits the CU's LDS as V^T (attn_f32_max_seq, kernels_attention.h)


def float_attention_applies(x: torch.Tensor, S: int, hd: int) -> bool:
    """fp32 hidden states, a head size the kernel serves and a sequence whose V^T fits the LDS (512 / 256 / 128 tokens for heads
    of 64 / 128 / 256); TS_ENCODER_ATTENTION=0 keeps torch's attention."""
    return (x.dtype == torch.float32 and S <= _FLOAT_ATTENTION_MAX_SEQ.get(hd, 0) and os.environ.get("TS_ENCODER_ATTENTION", "1") != "0")


class FusedBertForward:
    """The forward of a BERT-family encoder (``BertModel``: what ``math-similarity/Bert-MLM_arXiv-MP-class_zbMath`` is,
    compare_embeddings.py:11) with the launches that do not pay for themselves folded together:

    * query / key / value projections as ONE GEMM over the concatenated weight (three 8,192 x 768 x 768 GEMMs fill the
      chip a third each: 24 us apiece against 41 us for the fused one, measured per layer at 256 x 32 tokens);
    * ``LayerNorm(dense_out + input)`` as ONE HIP kernel (``ts_add_layernorm``) instead of an add and a layer_norm launch
      (18 + 7 us of device time twice per layer);
    * the input layer (three embedding gathers, two adds, LayerNorm) as ONE HIP kernel (``ts_embed_layernorm``);
    * for bf16 models with 64-wide heads and at most 128 tokens, the attention as ONE wave per (sequence, head)
      (``ts_attention_short``; ``TS_ENCODER_ATTENTION=0`` keeps ``scaled_dot_product_attention``).

    Same weights, same order of operations, exact erf GELU (whatever ``config.hidden_act`` names); the attention is
    ``scaled_dot_product_attention`` with the padding mask, as the model's own ``sdpa`` path.  Used on a GPU for bf16 / fp32
    models whose config this form covers; anything else runs the model's own forward."""

    def __init__(self, model):
        cfg = model.config
        self.model, self.cfg = model, cfg
        self.heads = cfg.num_attention_heads
        self.eps = float(cfg.layer_norm_eps)
        from transformers.activations import ACT2FN
        self.act = ACT2FN[cfg.hidden_act] if isinstance(cfg.hidden_act, str) else cfg.hidden_act
        self.pieces = False           # fp32 models: the GEMMs on the bf16 matrix pipe from bf16 pieces (pieces_linear)
        self._stamp = None
        self._refresh()

    def _sources(self):
        """The parameters the stacked projection weights are copies of."""
        for layer in self.model.encoder.layer:
            att = layer.attention.self
            yield from (att.query.weight, att.key.weight, att.value.weight, att.query.bias, att.key.bias, att.value.bias)

    def _refresh(self):
        """(Re)build the stacked query / key / value weights when the model's own have changed (load_state_dict, .to(dtype),
        an edit in place): the other weights are live references, the stacked ones are copies."""
        stamp = tuple((p.data_ptr(), p._version, p.dtype) for p in self._sources()) + (self.pieces,)
        if stamp == self._stamp:
            return
        self._stamp = stamp
        self.layers = []
        for layer in self.model.encoder.layer:
            att, so = layer.attention.self, layer.attention.output
            self.layers.append({
                "wqkv": torch.cat([att.query.weight, att.key.weight, att.value.weight], dim=0).contiguous(),
                "bqkv": torch.cat([att.query.bias, att.key.bias, att.value.bias], dim=0).contiguous(),
                "wo": so.dense.weight, "bo": so.dense.bias, "ln1": so.LayerNorm,
                "w1": layer.intermediate.dense.weight, "b1": layer.intermediate.dense.bias,
                "w2": layer.output.dense.weight, "b2": layer.output.dense.bias, "ln2": layer.output.LayerNorm,
            })
            if self.pieces and self.layers[-1]["wo"].dtype == torch.float32:
                L = self.layers[-1]
                for name in ("wqkv", "wo", "w1", "w2"):
                    L[name + "_p"] = split_pieces(L[name].detach(), 1)

    @staticmethod
    def covers(model) -> bool:
        cfg = getattr(model, "config", None)
        if cfg is None or getattr(cfg, "model_type", "") != "bert" or not hasattr(model, "encoder"):
            return False
        if getattr(cfg, "position_embedding_type", "absolute") != "absolute" or getattr(cfg, "is_decoder", False):
            return False
        p = next(model.parameters())
        return (p.is_cuda and p.dtype in (torch.float32, torch.bfloat16) and cfg.hidden_size % 8 == 0 and cfg.hidden_size <= 1024
                and cfg.hidden_size % cfg.num_attention_heads == 0)

    def _embed(self, input_ids: torch.Tensor, token_type_ids: Optional[torch.Tensor]) -> torch.Tensor:
        """BertEmbeddings (word + token type + position, LayerNorm; dropout is the identity in eval) as ONE HIP kernel
        (``ts_embed_layernorm``) instead of three gathers, two adds and a layer_norm launch.  Anything the kernel's form does not
        cover (a sequence longer than the position table, a module without the three tables) runs the module itself."""
        import ctypes as C
        from . import _ffi
        emb = self.model.embeddings
        tables = [getattr(emb, n, None) for n in ("word_embeddings", "position_embeddings", "token_type_embeddings")]
        ln = getattr(emb, "LayerNorm", None)
        B, S = input_ids.shape
        if any(t is None for t in tables) or ln is None or S > tables[1].weight.shape[0] or input_ids.dtype != torch.int64:
            return emb(input_ids=input_ids, token_type_ids=token_type_ids)
        w, p, t = (m.weight for m in tables)
        ids = input_ids.contiguous()
        tt = token_type_ids.contiguous().to(torch.int64) if token_type_ids is not None else None
        H = w.shape[1]
        out = torch.empty((B, S, H), dtype=w.dtype, device=w.device)
        _ffi.check(_ffi.load().ts_embed_layernorm(
            w.device.index or 0, C.c_void_p(ids.data_ptr()), C.c_void_p(tt.data_ptr()) if tt is not None else None,
            C.c_void_p(w.data_ptr()), C.c_void_p(p.data_ptr()), C.c_void_p(t.data_ptr()), w.shape[0], p.shape[0], t.shape[0],
            C.c_void_p(ln.weight.data_ptr()), C.c_void_p(ln.bias.data_ptr()), self.eps, B * S, S, H,
            1 if w.dtype == torch.bfloat16 else 0, C.c_void_p(out.data_ptr()),
            C.c_void_p(torch.cuda.current_stream(w.device).cuda_stream)))
        return out

    def _attention(self, qkv: torch.Tensor, key_mask: Optional[torch.Tensor], B: int, S: int) -> torch.Tensor:
        import ctypes as C
        from . import _ffi
        qkv = qkv.contiguous()
        out = torch.empty((B, S, self.heads * 64), dtype=torch.bfloat16, device=qkv.device)
        _ffi.check(_ffi.load().ts_attention_short(
            qkv.device.index or 0, C.c_void_p(qkv.data_ptr()), C.c_void_p(key_mask.data_ptr()) if key_mask is not None else None,
            B, S, self.heads, 64, C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(qkv.device).cuda_stream)))
        return out

    def _add_ln_pieces(self, a: torch.Tensor, b: torch.Tensor, ln, a_bias: Optional[torch.Tensor] = None):
        """`_add_ln` of fp32 operands (``a + a_bias + b``) that also writes the pieces of its output: ``(out fp32, pieces bf16
        [rows x 3d])``."""
        import ctypes as C
        from . import _ffi
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        rows, d = a.numel() // a.shape[-1], a.shape[-1]
        pieces = torch.empty((rows, 3 * d), dtype=torch.bfloat16, device=a.device)
        _ffi.check(_ffi.load().ts_add_layernorm_pieces(
            a.device.index or 0, C.c_void_p(a.data_ptr()), C.c_void_p(a_bias.data_ptr()) if a_bias is not None else None,
            C.c_void_p(b.data_ptr()), C.c_void_p(ln.weight.data_ptr()),
            C.c_void_p(ln.bias.data_ptr()), self.eps, rows, d, C.c_void_p(out.data_ptr()), C.c_void_p(pieces.data_ptr()),
            C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)))
        return out, pieces

    def _add_ln(self, a: torch.Tensor, b: torch.Tensor, ln) -> torch.Tensor:
        import ctypes as C
        from . import _ffi
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        rows, d = a.numel() // a.shape[-1], a.shape[-1]
        _ffi.check(_ffi.load().ts_add_layernorm(
            a.device.index or 0, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(ln.weight.data_ptr()),
            C.c_void_p(ln.bias.data_ptr()), self.eps, rows, d, 1 if a.dtype == torch.bfloat16 else 0, C.c_void_p(out.data_ptr()),
            C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)))
        return out

    def __call__(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, token_type_ids: Optional[torch.Tensor] = None,
                 no_padding: bool = False):
        F = torch.nn.functional
        self._refresh()
        x = self._embed(input_ids, token_type_ids)
        B, S, H = x.shape
        hd = H // self.heads
        # padding keys are never attended to: ONE additive mask per forward (a boolean mask is expanded to a bias inside every
        # scaled_dot_product_attention call: two fill launches per layer); none at all when the caller knows the batch has no
        # padding (every sequence as long as the batch: 50 instead of 60 us per layer for projections + attention)
        # short sequences (one sentence per query: app_showcase_model.py:92) of a bf16 model with 64-wide heads: the attention as
        # ONE wave per (sequence, head), straight from the fused projection to the context layout (``ts_attention_short``)
        short = x.dtype == torch.bfloat16 and hd == 64 and S <= 128 and os.environ.get("TS_ENCODER_ATTENTION", "1") != "0"
        # (the most negative finite value, not -inf: a sequence without a single token would otherwise soften to NaN)
        short32 = float_attention_applies(x, S, hd)          # fp32: the library's fp32 attention (ts_attention_float)
        mask = None if (no_padding or short or short32) else torch.zeros((B, 1, 1, S), dtype=x.dtype, device=x.device).masked_fill_(
            ~attention_mask[:, None, None, :].to(torch.bool), torch.finfo(x.dtype).min)
        key_mask = None if (no_padding or not (short or short32)) else attention_mask.to(torch.int64).contiguous()
        pieces = self.pieces and x.dtype == torch.float32 and "wo_p" in self.layers[0]
        if pieces:
            return self._forward_pieces(x, mask, key_mask, short32, B, S, H, hd)
        for L in self.layers:
            qkv = F.linear(x, L["wqkv"], L["bqkv"])
            if short:
                ctx = self._attention(qkv, key_mask, B, S)
            elif short32:
                ctx = attention_float(qkv, key_mask, B, S, self.heads, self.heads, hd, False, hd ** -0.5)[0]
            else:
                qkv = qkv.view(B, S, 3, self.heads, hd).permute(2, 0, 3, 1, 4)
                ctx = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], attn_mask=mask)
                ctx = ctx.transpose(1, 2).reshape(B, S, H)
            x = self._add_ln(F.linear(ctx, L["wo"], L["bo"]), x, L["ln1"])
            h = self.act(F.linear(x, L["w1"], L["b1"]))
            x = self._add_ln(F.linear(h, L["w2"], L["b2"]), x, L["ln2"])
        return x

    def _forward_pieces(self, x: torch.Tensor, mask: Optional[torch.Tensor], key_mask: Optional[torch.Tensor], short32: bool, B: int,
                        S: int, H: int, hd: int) -> torch.Tensor:
        """The layers of an fp32 model with every GEMM on the bf16 matrix pipe (``fp32_gemm="bf16x3"``): weights are pieces
        (``_refresh``), every activation reaches its GEMM as pieces written by the kernel that produced it (LayerNorm, GELU, the
        fp32 attention up to 128 tokens; beyond, torch's attention output passes through ``ts_split_pieces``)."""
        F = torch.nn.functional
        exact_gelu = self.cfg.hidden_act == "gelu"
        xp = split_pieces(x.view(B * S, H), 0)
        for L in self.layers:
            # every GEMM runs WITHOUT its bias: the kernel that reads its output adds it on the way in (torch.addmm with an output
            # type first copies the broadcast bias into the result: one more pass over every GEMM's output)
            qkv = pieces_mm(xp, L["wqkv_p"])
            if short32:                # the attention writes the pieces of its output itself
                cp = attention_float(qkv, key_mask, B, S, self.heads, self.heads, hd, False, hd ** -0.5, want_pieces=True, bias=L["bqkv"], want_context=False)[1]
            else:
                qkv = (qkv + L["bqkv"]).view(B, S, 3, self.heads, hd).permute(2, 0, 3, 1, 4)
                ctx = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], attn_mask=mask)
                cp = split_pieces(ctx.transpose(1, 2).reshape(B * S, H), 0)
            x, xp = self._add_ln_pieces(pieces_mm(cp, L["wo_p"]).view(B, S, H), x, L["ln1"], a_bias=L["bo"])
            h = pieces_mm(xp, L["w1_p"])
            hp = act_pieces(h, 0, bias=L["b1"]) if exact_gelu else split_pieces(self.act(h + L["b1"]), 0)
            x, xp = self._add_ln_pieces(pieces_mm(hp, L["w2_p"]).view(B, S, H), x, L["ln2"], a_bias=L["b2"])
        return x


class FusedQwen3Forward:
    """The forward of a Qwen3-family encoder (``Qwen3Model``: what ``Qwen/Qwen3-Embedding-0.6B`` is, the embedder of the
    production app, streamlit_app.py:55) with everything around its GEMMs as kernels of libtsearch:

    * query / key / value projections as ONE GEMM over the stacked weight, gate / up projections as ONE;
    * ``residual + sublayer`` followed by the next RMSNorm as ONE kernel (``ts_add_rmsnorm``: PyTorch runs an add and six
      launches per norm, twice per layer);
    * the per-head RMSNorm of queries and keys + the rotary embedding as ONE kernel, in place (``ts_qk_norm_rope``:
      twenty-two launches per layer in PyTorch);
    * ``silu(gate) * up`` as ONE kernel (``ts_swiglu``).

    * causal grouped-query attention of short sequences (bf16, up to 128 tokens) as ONE kernel straight from the stacked
      projection (``ts_attention_gqa``: one wave per (sequence, query head); torch's flash-attention launch took 202 us per
      layer at 256 sequences x 32 tokens, 504 us at 128); longer sequences and fp32 keep ``scaled_dot_product_attention``.

    Same weights, same order of operations, the roundings of the modules replaced."""

    def __init__(self, model):
        cfg = model.config
        self.model, self.cfg = model, cfg
        self.hq, self.hkv, self.hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        self.eps = float(cfg.rms_norm_eps)
        self.pieces = False           # fp32 models: the GEMMs on the bf16 matrix pipe from bf16 pieces (pieces_linear)
        self._stamp = None
        self._gqa_native = True
        self._refresh()

    def _sources(self):
        for layer in self.model.layers:
            att, mlp = layer.self_attn, layer.mlp
            yield from (att.q_proj.weight, att.k_proj.weight, att.v_proj.weight, mlp.gate_proj.weight, mlp.up_proj.weight)

    def _refresh(self):
        stamp = tuple((p.data_ptr(), p._version, p.dtype) for p in self._sources()) + (self.pieces,)
        if stamp == self._stamp:
            return
        self._stamp = stamp
        self.layers = []
        for layer in self.model.layers:
            att, mlp = layer.self_attn, layer.mlp
            self.layers.append({
                "wqkv": torch.cat([att.q_proj.weight, att.k_proj.weight, att.v_proj.weight], dim=0).contiguous(),
                "wo": att.o_proj.weight, "qn": att.q_norm.weight, "kn": att.k_norm.weight,
                "wgu": torch.cat([mlp.gate_proj.weight, mlp.up_proj.weight], dim=0).contiguous(), "wd": mlp.down_proj.weight,
                "ln1": layer.input_layernorm.weight, "ln2": layer.post_attention_layernorm.weight,
            })
            if self.pieces and self.layers[-1]["wo"].dtype == torch.float32:
                L = self.layers[-1]
                for name in ("wqkv", "wo", "wgu", "wd"):
                    L[name + "_p"] = split_pieces(L[name].detach(), 1)

    @staticmethod
    def covers(model) -> bool:
        cfg = getattr(model, "config", None)
        if cfg is None or getattr(cfg, "model_type", "") != "qwen3" or not hasattr(model, "layers"):
            return False
        if getattr(cfg, "attention_bias", False) or getattr(cfg, "head_dim", 0) != 128:
            return False
        if any(t != "full_attention" for t in (getattr(cfg, "layer_types", None) or [])):
            return False
        if getattr(cfg, "hidden_act", "silu") != "silu":
            return False
        p = next(model.parameters())
        vec = 8 if p.dtype == torch.bfloat16 else 4
        return (p.is_cuda and p.dtype in (torch.float32, torch.bfloat16) and cfg.hidden_size % vec == 0 and
                cfg.hidden_size <= 256 * vec and cfg.intermediate_size % vec == 0 and
                cfg.num_attention_heads % cfg.num_key_value_heads == 0)

    def _add_rmsnorm(self, a: torch.Tensor, b: Optional[torch.Tensor], gamma: torch.Tensor, want_sum: bool):
        import ctypes as C
        from . import _ffi
        d = a.shape[-1]
        rows = a.numel() // d
        out = torch.empty_like(a)
        new_res = torch.empty_like(a) if (want_sum and b is not None) else None
        _ffi.check(_ffi.load().ts_add_rmsnorm(
            a.device.index or 0, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()) if b is not None else None,
            C.c_void_p(gamma.data_ptr()), self.eps, rows, d, 1 if a.dtype == torch.bfloat16 else 0,
            C.c_void_p(new_res.data_ptr()) if new_res is not None else None, C.c_void_p(out.data_ptr()),
            C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)))
        return (new_res if new_res is not None else a), out

    def _add_rmsnorm_pieces(self, a: torch.Tensor, b: Optional[torch.Tensor], gamma: torch.Tensor, want_sum: bool):
        """`_add_rmsnorm` of fp32 operands that also writes the pieces of the normalised rows: ``(residual, normed, pieces)``."""
        import ctypes as C
        from . import _ffi
        d = a.shape[-1]
        rows = a.numel() // d
        out = torch.empty_like(a)
        new_res = torch.empty_like(a) if (want_sum and b is not None) else None
        pieces = torch.empty((rows, 3 * d), dtype=torch.bfloat16, device=a.device)
        _ffi.check(_ffi.load().ts_add_rmsnorm_pieces(
            a.device.index or 0, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()) if b is not None else None,
            C.c_void_p(gamma.data_ptr()), self.eps, rows, d, C.c_void_p(new_res.data_ptr()) if new_res is not None else None,
            C.c_void_p(out.data_ptr()), C.c_void_p(pieces.data_ptr()), C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)))
        return (new_res if new_res is not None else a), out, pieces

    def _sdpa(self, qkv: torch.Tensor, mask: Optional[torch.Tensor], B: int, S: int, nq: int, nkv: int, hd: int) -> torch.Tensor:
        """torch's attention on the stacked projection (longer sequences, fp32): causal, grouped-query, as the model's own sdpa path."""
        F = torch.nn.functional
        q = qkv[..., :nq].view(B, S, self.hq, hd).transpose(1, 2)
        k = qkv[..., nq:nq + nkv].view(B, S, self.hkv, hd).transpose(1, 2)
        v = qkv[..., nq + nkv:].view(B, S, self.hkv, hd).transpose(1, 2)
        ctx = None
        if self._gqa_native:
            try:
                ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, is_causal=mask is None, enable_gqa=True)
            except (RuntimeError, TypeError):
                self._gqa_native = False
        if ctx is None:
            rep = self.hq // self.hkv
            ctx = F.scaled_dot_product_attention(q, k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1),
                                                 attn_mask=mask, is_causal=mask is None)
        return ctx.transpose(1, 2).reshape(B, S, nq)

    def __call__(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, token_type_ids: Optional[torch.Tensor] = None,
                 no_padding: bool = False):
        import ctypes as C
        from . import _ffi
        F = torch.nn.functional
        lib = _ffi.load()
        self._refresh()
        m = self.model
        x = m.embed_tokens(input_ids).contiguous()
        B, S, H = x.shape
        dt = 1 if x.dtype == torch.bfloat16 else 0
        dev = x.device.index or 0
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        pos = torch.arange(S, device=x.device).unsqueeze(0)              # positions count from the left edge, padding included
        cos, sin = m.rotary_emb(x, pos)                                  # [1 x S x 128] of the model's type
        cos, sin = cos[0].contiguous(), sin[0].contiguous()
        mask = None
        # short sequences in bf16: the library's own causal grouped-query attention (TS_ENCODER_ATTENTION=0 keeps torch's)
        short = (x.dtype == torch.bfloat16 and self.hd == 128 and S <= 128 and x.is_contiguous() and
                 os.environ.get("TS_ENCODER_ATTENTION", "1") != "0")
        short32 = float_attention_applies(x, S, self.hd)      # fp32: the library's fp32 attention (causal, grouped-query)
        key_mask = None if (no_padding or not (short or short32)) else attention_mask.to(torch.int64).contiguous()
        if not no_padding and not short and not short32:
            # causal, and padding keys are never attended to (the most negative finite value: rows of padding stay finite)
            neg = torch.finfo(x.dtype).min
            causal = torch.ones((S, S), dtype=torch.bool, device=x.device).tril_()
            keep = causal[None, None] & attention_mask[:, None, None, :].to(torch.bool)
            mask = torch.zeros((B, 1, S, S), dtype=x.dtype, device=x.device).masked_fill_(~keep, neg)
        nq, nkv, hd = self.hq * self.hd, self.hkv * self.hd, self.hd
        pieces = self.pieces and x.dtype == torch.float32 and "wo_p" in self.layers[0]
        if pieces:
            # every GEMM on the bf16 matrix pipe from pieces: weights split once (`_refresh`), activations written as pieces by
            # the kernel that produces them (RMSNorm, SwiGLU); only torch's attention output passes through ts_split_pieces
            _, h, hp = self._add_rmsnorm_pieces(x, None, self.layers[0]["ln1"], False)
            for li, L in enumerate(self.layers):
                qkv = pieces_mm(hp, L["wqkv_p"]).view(B, S, -1)
                _ffi.check(lib.ts_qk_norm_rope(dev, C.c_void_p(qkv.data_ptr()), C.c_void_p(L["qn"].data_ptr()), C.c_void_p(L["kn"].data_ptr()),
                                               C.c_void_p(cos.data_ptr()), C.c_void_p(sin.data_ptr()), self.eps, B * S, S, self.hq, self.hkv,
                                               hd, dt, stream))
                if short32:
                    cp = attention_float(qkv, key_mask, B, S, self.hq, self.hkv, hd, True, hd ** -0.5, want_pieces=True, want_context=False)[1]
                else:
                    cp = split_pieces(self._sdpa(qkv, mask, B, S, nq, nkv, hd).reshape(B * S, nq), 0)
                x, h, hp = self._add_rmsnorm_pieces(x, pieces_mm(cp, L["wo_p"]).view(B, S, H), L["ln2"], True)
                ap = act_pieces(pieces_mm(hp, L["wgu_p"]), 1)
                last = li + 1 == len(self.layers)
                gamma = m.norm.weight if last else self.layers[li + 1]["ln1"]
                x, h, hp = self._add_rmsnorm_pieces(x, pieces_mm(ap, L["wd_p"]).view(B, S, H), gamma, not last)
            return h
        h = self._add_rmsnorm(x, None, self.layers[0]["ln1"], False)[1]
        for li, L in enumerate(self.layers):
            qkv = F.linear(h, L["wqkv"])
            _ffi.check(lib.ts_qk_norm_rope(dev, C.c_void_p(qkv.data_ptr()), C.c_void_p(L["qn"].data_ptr()), C.c_void_p(L["kn"].data_ptr()),
                                           C.c_void_p(cos.data_ptr()), C.c_void_p(sin.data_ptr()), self.eps, B * S, S, self.hq, self.hkv,
                                           hd, dt, stream))
            if short:
                # one wave per (sequence, query head), straight from the stacked projection's output (ts_attention_gqa)
                ctx = torch.empty((B, S, nq), dtype=x.dtype, device=x.device)
                _ffi.check(lib.ts_attention_gqa(dev, C.c_void_p(qkv.data_ptr()), C.c_void_p(key_mask.data_ptr()) if key_mask is not None else None,
                                               B, S, self.hq, self.hkv, hd, 1, C.c_void_p(ctx.data_ptr()), stream))
            elif short32:
                ctx = attention_float(qkv, key_mask, B, S, self.hq, self.hkv, hd, True, hd ** -0.5)[0]
            else:
                ctx = self._sdpa(qkv, mask, B, S, nq, nkv, hd)
            x, h = self._add_rmsnorm(x, F.linear(ctx, L["wo"]), L["ln2"], True)
            gu = F.linear(h, L["wgu"])
            inter = gu.shape[-1] // 2
            act = torch.empty((B, S, inter), dtype=x.dtype, device=x.device)
            _ffi.check(lib.ts_swiglu(dev, C.c_void_p(gu.data_ptr()), B * S, inter, dt, C.c_void_p(act.data_ptr()), stream))
            last = li + 1 == len(self.layers)
            gamma = m.norm.weight if last else self.layers[li + 1]["ln1"]
            x, h = self._add_rmsnorm(x, F.linear(act, L["wd"]), gamma, not last)
        return h


class FusedGemma3Forward:
    """The forward of a Gemma3 text encoder with bidirectional attention (``Gemma3TextModel``: what
    ``google/embeddinggemma-300m`` is, the reference's second embedder, ec2/generate_embeddings/embedders.py:1-4) with everything
    around its GEMMs and its attention as kernels of libtsearch:

    * query / key / value projections as ONE GEMM over the stacked weight, gate / up projections as ONE;
    * the post-sublayer RMSNorm, the residual add and the pre-norm of the next sublayer as ONE kernel (``ts_gemma_norm``:
      PyTorch runs two norms of seven launches each and an add, twice per layer);
    * the per-head RMSNorm of queries and keys + the rotary embedding as ONE kernel, in place (``ts_gemma_qk_norm_rope``;
      sliding and full attention layers have their own cos / sin tables);
    * ``gelu_tanh(gate) * up`` as ONE kernel (``ts_geglu``).

    The attention is ``scaled_dot_product_attention`` (heads of 256, grouped-query, every key visible: sequences shorter than
    the sliding window).  Same weights, same order of operations, the roundings of the modules replaced.  Longer sequences
    than the sliding window take the model's own forward."""

    def __init__(self, model):
        cfg = model.config
        self.model, self.cfg = model, cfg
        self.hq, self.hkv, self.hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        self.eps = float(cfg.rms_norm_eps)
        self.scaling = float(cfg.query_pre_attn_scalar) ** -0.5
        self.pieces = False           # fp32 models: the GEMMs on the bf16 matrix pipe from bf16 pieces (pieces_linear)
        self._stamp = None
        self._gqa_native = True
        self._refresh()

    def _sources(self):
        for layer in self.model.layers:
            att, mlp = layer.self_attn, layer.mlp
            yield from (att.q_proj.weight, att.k_proj.weight, att.v_proj.weight, mlp.gate_proj.weight, mlp.up_proj.weight)

    def _refresh(self):
        stamp = tuple((p.data_ptr(), p._version, p.dtype) for p in self._sources()) + (self.pieces,)
        if stamp == self._stamp:
            return
        self._stamp = stamp
        self.layers = []
        for layer in self.model.layers:
            att, mlp = layer.self_attn, layer.mlp
            self.layers.append({
                "wqkv": torch.cat([att.q_proj.weight, att.k_proj.weight, att.v_proj.weight], dim=0).contiguous(),
                "wo": att.o_proj.weight, "qn": att.q_norm.weight, "kn": att.k_norm.weight,
                "wgu": torch.cat([mlp.gate_proj.weight, mlp.up_proj.weight], dim=0).contiguous(), "wd": mlp.down_proj.weight,
                "ln_in": layer.input_layernorm.weight, "ln_post_attn": layer.post_attention_layernorm.weight,
                "ln_pre_ffn": layer.pre_feedforward_layernorm.weight, "ln_post_ffn": layer.post_feedforward_layernorm.weight,
                "type": att.layer_type,
            })
            if self.pieces and self.layers[-1]["wo"].dtype == torch.float32:
                L = self.layers[-1]
                for name in ("wqkv", "wo", "wgu", "wd"):
                    L[name + "_p"] = split_pieces(L[name].detach(), 1)

    @staticmethod
    def covers(model) -> bool:
        cfg = getattr(model, "config", None)
        if cfg is None or getattr(cfg, "model_type", "") != "gemma3_text" or not hasattr(model, "layers"):
            return False
        if not getattr(cfg, "use_bidirectional_attention", False) or getattr(cfg, "attention_bias", False):
            return False
        if getattr(cfg, "attn_logit_softcapping", None) or getattr(cfg, "head_dim", 0) != 256:
            return False
        if getattr(cfg, "hidden_activation", "") != "gelu_pytorch_tanh":
            return False
        p = next(model.parameters())
        vec = 8 if p.dtype == torch.bfloat16 else 4
        return (p.is_cuda and p.dtype in (torch.float32, torch.bfloat16) and cfg.hidden_size % vec == 0 and
                cfg.hidden_size <= 256 * vec and cfg.intermediate_size % vec == 0 and
                cfg.num_attention_heads % cfg.num_key_value_heads == 0)

    def _norm(self, y: Optional[torch.Tensor], x: torch.Tensor, w_post: Optional[torch.Tensor], w_next: torch.Tensor, want_sum: bool):
        import ctypes as C
        from . import _ffi
        d = x.shape[-1]
        rows = x.numel() // d
        out = torch.empty_like(x)
        new_res = torch.empty_like(x) if (want_sum and y is not None) else None
        _ffi.check(_ffi.load().ts_gemma_norm(
            x.device.index or 0, C.c_void_p(y.data_ptr()) if y is not None else None, C.c_void_p(x.data_ptr()),
            C.c_void_p(w_post.data_ptr()) if w_post is not None else None, C.c_void_p(w_next.data_ptr()), self.eps, rows, d,
            1 if x.dtype == torch.bfloat16 else 0, C.c_void_p(new_res.data_ptr()) if new_res is not None else None,
            C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        return (new_res if new_res is not None else x), out

    def _sdpa(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask: Optional[torch.Tensor]) -> torch.Tensor:
        """torch's attention (sequences past 128 tokens, bf16): every key visible, grouped-query, Gemma3's own scaling."""
        F = torch.nn.functional
        if self._gqa_native:
            try:
                return F.scaled_dot_product_attention(q, k, v, attn_mask=mask, scale=self.scaling, enable_gqa=True)
            except (RuntimeError, TypeError):
                self._gqa_native = False
        rep = self.hq // self.hkv
        return F.scaled_dot_product_attention(q, k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1), attn_mask=mask,
                                              scale=self.scaling)

    def _norm_pieces(self, y: Optional[torch.Tensor], x: torch.Tensor, w_post: Optional[torch.Tensor], w_next: torch.Tensor, want_sum: bool):
        """`_norm` of fp32 operands that also writes the pieces of the pre-normed rows: ``(residual, normed, pieces)``."""
        import ctypes as C
        from . import _ffi
        d = x.shape[-1]
        rows = x.numel() // d
        out = torch.empty_like(x)
        new_res = torch.empty_like(x) if (want_sum and y is not None) else None
        pieces = torch.empty((rows, 3 * d), dtype=torch.bfloat16, device=x.device)
        _ffi.check(_ffi.load().ts_gemma_norm_pieces(
            x.device.index or 0, C.c_void_p(y.data_ptr()) if y is not None else None, C.c_void_p(x.data_ptr()),
            C.c_void_p(w_post.data_ptr()) if w_post is not None else None, C.c_void_p(w_next.data_ptr()), self.eps, rows, d,
            C.c_void_p(new_res.data_ptr()) if new_res is not None else None, C.c_void_p(out.data_ptr()), C.c_void_p(pieces.data_ptr()),
            C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        return (new_res if new_res is not None else x), out, pieces

    def __call__(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, token_type_ids: Optional[torch.Tensor] = None,
                 no_padding: bool = False):
        import ctypes as C
        from . import _ffi
        F = torch.nn.functional
        B, S = input_ids.shape
        if S >= int(self.cfg.sliding_window):                        # the sliding layers would hide keys: the model's own masks
            return self.model(input_ids=input_ids, attention_mask=attention_mask).last_hidden_state
        lib = _ffi.load()
        self._refresh()
        m = self.model
        x = m.embed_tokens(input_ids).contiguous()                    # scaled by sqrt(hidden) in the storage type, as the module does
        dt = 1 if x.dtype == torch.bfloat16 else 0
        dev = x.device.index or 0
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        pos = torch.arange(S, device=x.device).unsqueeze(0)
        tables = {}
        for lt in set(self.cfg.layer_types):
            cos, sin = m.rotary_emb(x, pos, lt)                       # [1 x S x 256] of the model's type
            tables[lt] = (cos[0].contiguous(), sin[0].contiguous())
        mask = None
        short32 = float_attention_applies(x, S, self.hd)      # fp32: the library's fp32 attention (every key visible, heads of 256)
        key_mask = attention_mask.to(torch.int64).contiguous() if (short32 and not no_padding) else None
        if not no_padding and not short32:
            neg = torch.finfo(x.dtype).min
            mask = torch.zeros((B, 1, 1, S), dtype=x.dtype, device=x.device).masked_fill_(~attention_mask[:, None, None, :].to(torch.bool), neg)
        nq, nkv, hd = self.hq * self.hd, self.hkv * self.hd, self.hd
        pieces = self.pieces and x.dtype == torch.float32 and "wo_p" in self.layers[0]
        H = x.shape[-1]
        hp = None
        if pieces:
            _, h, hp = self._norm_pieces(None, x, None, self.layers[0]["ln_in"], False)
        else:
            h = self._norm(None, x, None, self.layers[0]["ln_in"], False)[1]
        for li, L in enumerate(self.layers):
            qkv = pieces_mm(hp, L["wqkv_p"]).view(B, S, -1) if pieces else F.linear(h, L["wqkv"])
            cos, sin = tables[L["type"]]
            _ffi.check(lib.ts_gemma_qk_norm_rope(dev, C.c_void_p(qkv.data_ptr()), C.c_void_p(L["qn"].data_ptr()), C.c_void_p(L["kn"].data_ptr()),
                                                 C.c_void_p(cos.data_ptr()), C.c_void_p(sin.data_ptr()), self.eps, B * S, S, self.hq, self.hkv,
                                                 hd, dt, stream))
            cp = None
            if short32:
                ctx, cp = attention_float(qkv, key_mask, B, S, self.hq, self.hkv, hd, False, self.scaling, want_pieces=pieces, want_context=not pieces)
            else:
                q = qkv[..., :nq].view(B, S, self.hq, hd).transpose(1, 2)
                k = qkv[..., nq:nq + nkv].view(B, S, self.hkv, hd).transpose(1, 2)
                v = qkv[..., nq + nkv:].view(B, S, self.hkv, hd).transpose(1, 2)
                ctx = self._sdpa(q, k, v, mask).transpose(1, 2).reshape(B, S, nq)
            last = li + 1 == len(self.layers)
            w_next = m.norm.weight if last else self.layers[li + 1]["ln_in"]
            if pieces:
                # GEMMs on the bf16 matrix pipe from pieces; the norms and the GeGLU write the pieces of what they produce
                if cp is None:
                    cp = split_pieces(ctx.reshape(B * S, nq), 0)
                x, h, hp = self._norm_pieces(pieces_mm(cp, L["wo_p"]).view(B, S, H), x, L["ln_post_attn"], L["ln_pre_ffn"], True)
                ap = act_pieces(pieces_mm(hp, L["wgu_p"]), 2)
                x, h, hp = self._norm_pieces(pieces_mm(ap, L["wd_p"]).view(B, S, H), x, L["ln_post_ffn"], w_next, not last)
                continue
            x, h = self._norm(F.linear(ctx, L["wo"]), x, L["ln_post_attn"], L["ln_pre_ffn"], True)
            gu = F.linear(h, L["wgu"])
            inter = gu.shape[-1] // 2
            act = torch.empty((B, S, inter), dtype=x.dtype, device=x.device)
            _ffi.check(lib.ts_geglu(dev, C.c_void_p(gu.data_ptr()), B * S, inter, dt, C.c_void_p(act.data_ptr()), stream))
            x, h = self._norm(F.linear(act, L["wd"]), x, L["ln_post_ffn"], w_next, not last)
        return h
