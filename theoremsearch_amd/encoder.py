"""Sentence encoder with the ``SentenceTransformer.encode`` call shape used by the reference.

PyTorch-ROCm runs the transformer forward (the only place this package uses torch for compute); pooling +
optional L2 normalisation are one fused HIP kernel on the GPU (``ts_pool_normalize``).  Call sites mirrored:

    model.encode(texts, convert_to_numpy=True, normalize_embeddings=True, show_progress_bar=False)
                                                        parsed_papers_to_vector_rds/embeddings.py:31-37
    embedder.encode(texts, normalize_embeddings=True, show_progress_bar=False, batch_size=16)
    embedder.encode_multi_process(texts, pool=None, normalize_embeddings=True, ...)
                                                        ec2/generate_embeddings/embeddings.py:24-38
    model.encode(corpus_texts, convert_to_tensor=True, show_progress_bar=True)   app_create_embeddings.py:81
    model.encode(query, convert_to_tensor=True)                                   app_showcase_model.py:92
    model.encode(query or "", normalize_embeddings=True, convert_to_numpy=True)   streamlit_app.py:173

Checkpoints.  ``SentenceEncoder(name)`` loads a sentence-transformers checkpoint from a LOCAL directory (``name``
itself, or ``$TS_MODEL_DIR/<name>``; nothing can be downloaded here) and runs the module pipeline the checkpoint
declares, read from its own files - never from a table in this module:

    modules.json                        the ordered module list (Transformer, Pooling, Dense*, Normalize)
    sentence_bert_config.json           max_seq_length of the Transformer module
    <n>_Pooling/config.json             pooling mode (cls / mean / max / mean_sqrt_len / lasttoken)
    <n>_Dense/config.json + weights     Linear (+ bias) + activation (embeddinggemma carries two of them)
    config_sentence_transformers.json   prompts / default_prompt_name
    tokenizer files                     padding side, truncation (Qwen3 pads on the left; last-token pooling follows)

A module type or option this class cannot run raises ``NotImplementedError`` - it never substitutes another pipeline
silently.  A directory without ``modules.json`` is a plain transformers checkpoint: sentence-transformers then applies
mean pooling, and so does this class.  Pretrained weights run in fp32 by default (as the reference does) with the
checkpoint's own ``max_seq_length``.  Two opt-ins trade arithmetic for speed (measured on the random-init stand-ins at the
published depths, tests/test_fulldepth_gpu.py, profiles/r05_encoder_topk_agreement.jsonl):
``fp32_gemm="bf16x3"`` keeps fp32 weights and activations and runs every GEMM on the bf16 matrix pipe from bf16 pieces
(embeddings 1 - cos <= 8.2e-8 from the fp32 forward's; 99.3-99.8 % of the top-10 positions over a 10M-row bf16 index unchanged;
1.6-2.2 x the throughput); ``dtype=torch.bfloat16`` runs the whole forward in bf16 (1 - cos up to 1.4e-4; it REORDERS 14-32 % of
the top-10 positions of random rows and changes the best row of up to 3.5 % of the queries; 2.6-4.3 x): a speed option for
corpora whose neighbouring scores lie further apart than that, not the reference's answer.

No weights offline.  When no checkpoint is found the constructor RAISES, unless ``allow_random_init=True`` (or
``TS_ALLOW_RANDOM_ENCODER=1``) asks for the stand-in used by the benchmark and the tests: a randomly initialised,
seeded BERT of the architecture family's width with a hashing word-piece tokenizer.  Its vectors carry no meaning;
numerical parity with the published checkpoints is unpinned.
"""
from __future__ import annotations

import json
import os
import re
import zlib
from typing import Iterable, List, Optional, Sequence, Union

import numpy as np
import torch

from .fused_forward import FusedBertForward, FusedGemma3Forward, FusedQwen3Forward   # noqa: F401  (re-exported)

# Stand-in architectures for allow_random_init only (no weights offline): the published shapes of the reference's embedders
# (ec2/generate_embeddings/embedders.py:1-4, app_create_embeddings.py:8).  family "bert": BertModel; family "qwen3": Qwen3Model
# (decoder-style: RMSNorm, rotary positions, 16 query / 8 key-value heads of 128, gated MLP, last-token pooling); family "gemma3":
# Gemma3TextModel with bidirectional attention (sandwich RMSNorms, 3 query heads over 1 key-value head of 256, GeGLU, sliding /
# full attention layers 5 : 1, mean pooling, two Dense modules 768 -> 3072 -> 768 behind the pooling).
ARCHITECTURES = {
    "math-similarity/Bert-MLM_arXiv-MP-class_zbMath": dict(family="bert", hidden=768, layers=12, heads=12, ffn=3072, pooling="mean", max_len=512),
    "google/embeddinggemma-300m": dict(family="gemma3", hidden=768, layers=24, heads=3, kv_heads=1, head_dim=256, ffn=1152,
                                       pooling="mean", max_len=2048, vocab=262144, sliding_window=512, rms_eps=1e-6,
                                       dense=(3072, 768)),
    "Qwen/Qwen3-Embedding-0.6B": dict(family="qwen3", hidden=1024, layers=28, heads=16, kv_heads=8, head_dim=128, ffn=3072,
                                      pooling="lasttoken", max_len=8192, vocab=151669, rope_theta=1000000.0, rms_eps=1e-6),
}
DEFAULT_ARCH = ARCHITECTURES["math-similarity/Bert-MLM_arXiv-MP-class_zbMath"]

_TOKEN_RE = re.compile(r"\\[A-Za-z]+|[A-Za-z]+|\d+|[^\sA-Za-z\d]")
_POOL_CODES = {"mean": 0, "lasttoken": 1, "cls": 2}      # TS_POOL_* of include/tsearch.h
_ACTIVATIONS = {
    "torch.nn.modules.linear.Identity": torch.nn.Identity,
    "torch.nn.modules.activation.Tanh": torch.nn.Tanh,
    "torch.nn.modules.activation.ReLU": torch.nn.ReLU,
    "torch.nn.modules.activation.GELU": torch.nn.GELU,
    "torch.nn.modules.activation.Sigmoid": torch.nn.Sigmoid,
}


class HashingTokenizer:
    """Stand-in tokenizer: LaTeX-aware word split, stable CRC32 hashing into a BERT-sized vocabulary,
    [CLS] ... [SEP], right padding, truncation.  Deterministic across processes and hosts."""

    def __init__(self, vocab_size: int = 30522, max_length: int = 512):
        self.vocab_size, self.max_length = vocab_size, max_length
        self.pad_id, self.cls_id, self.sep_id = 0, 101, 102

    def token_ids(self, text: str) -> List[int]:
        ids = [1000 + zlib.crc32(t.lower().encode("utf-8")) % (self.vocab_size - 1000) for t in _TOKEN_RE.findall(text)]
        return [self.cls_id] + ids[: self.max_length - 2] + [self.sep_id]

    def __call__(self, texts: Sequence[str]):
        rows = [self.token_ids(t) for t in texts]
        width = max(len(r) for r in rows)
        ids = np.full((len(rows), width), self.pad_id, dtype=np.int64)
        mask = np.zeros((len(rows), width), dtype=np.int64)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = r
            mask[i, : len(r)] = 1
        return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}


def _local_dir(name: str) -> Optional[str]:
    for cand in (name, os.path.join(os.environ.get("TS_MODEL_DIR", ""), name)):
        if cand and os.path.isdir(cand) and (os.path.exists(os.path.join(cand, "config.json")) or
                                             os.path.exists(os.path.join(cand, "modules.json"))):
            return cand
    return None


def _read_json(path: str, default=None):
    if not os.path.exists(path):
        return default
    with open(path, encoding="utf-8") as f:
        return json.load(f)


def _load_state(folder: str) -> dict:
    st = os.path.join(folder, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st)
    pt = os.path.join(folder, "pytorch_model.bin")
    if os.path.exists(pt):
        return torch.load(pt, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"no weights (model.safetensors / pytorch_model.bin) in {folder}")


class _Pipeline:
    """What a checkpoint's modules.json declares, reduced to what runs after the transformer forward."""

    def __init__(self):
        self.transformer_path = ""
        self.max_seq_length: Optional[int] = None
        self.pooling = "mean"
        self.include_prompt = True
        self.dense: List[torch.nn.Module] = []     # applied in order after pooling
        self.normalize = False                     # a Normalize module: the output is always unit length
        self.prompts: dict = {}
        self.default_prompt_name: Optional[str] = None
        self.similarity_fn_name: str = "cosine"


def read_pipeline(folder: str) -> _Pipeline:
    """Parse ``modules.json`` and the per-module configs of a sentence-transformers checkpoint directory."""
    p = _Pipeline()
    cst = _read_json(os.path.join(folder, "config_sentence_transformers.json"), {}) or {}
    p.prompts = dict(cst.get("prompts") or {})
    p.default_prompt_name = cst.get("default_prompt_name")
    p.similarity_fn_name = cst.get("similarity_fn_name") or "cosine"
    modules = _read_json(os.path.join(folder, "modules.json"))
    if modules is None:
        # a plain transformers checkpoint: sentence-transformers wraps it as Transformer + mean Pooling
        sb = _read_json(os.path.join(folder, "sentence_bert_config.json"), {}) or {}
        p.max_seq_length = sb.get("max_seq_length")
        return p
    seen_pooling = False
    for m in sorted(modules, key=lambda x: x.get("idx", 0)):
        kind = str(m.get("type", "")).rsplit(".", 1)[-1]
        sub = os.path.join(folder, m.get("path", "") or "")
        if kind == "Transformer":
            if m.get("idx", 0) != 0:
                raise NotImplementedError("a Transformer module that is not the first module")
            p.transformer_path = m.get("path", "") or ""
            sb = _read_json(os.path.join(sub, "sentence_bert_config.json"), {}) or {}
            p.max_seq_length = sb.get("max_seq_length")
        elif kind == "Pooling":
            cfg = _read_json(os.path.join(sub, "config.json"), {}) or {}
            modes = [name for name, key in (("cls", "pooling_mode_cls_token"), ("mean", "pooling_mode_mean_tokens"),
                                            ("max", "pooling_mode_max_tokens"), ("mean_sqrt_len", "pooling_mode_mean_sqrt_len_tokens"),
                                            ("weightedmean", "pooling_mode_weightedmean_tokens"), ("lasttoken", "pooling_mode_lasttoken"))
                     if cfg.get(key)]
            if len(modes) != 1:
                raise NotImplementedError(f"pooling with modes {modes}: exactly one mode is supported")
            if modes[0] == "weightedmean":
                raise NotImplementedError("pooling_mode_weightedmean_tokens")
            p.pooling = modes[0]
            p.include_prompt = bool(cfg.get("include_prompt", True))
            seen_pooling = True
        elif kind == "Dense":
            if not seen_pooling:
                raise NotImplementedError("a Dense module in front of the Pooling module")
            cfg = _read_json(os.path.join(sub, "config.json"))
            if cfg is None:
                raise FileNotFoundError(f"{sub}/config.json")
            act = cfg.get("activation_function", "torch.nn.modules.activation.Tanh")
            if act not in _ACTIVATIONS:
                raise NotImplementedError(f"Dense activation {act}")
            lin = torch.nn.Linear(int(cfg["in_features"]), int(cfg["out_features"]), bias=bool(cfg.get("bias", True)))
            state = _load_state(sub)
            lin.load_state_dict({k.split("linear.", 1)[1]: v for k, v in state.items() if k.startswith("linear.")})
            p.dense.append(torch.nn.Sequential(lin, _ACTIVATIONS[act]()))
        elif kind == "Normalize":
            p.normalize = True
        else:
            raise NotImplementedError(f"sentence-transformers module {m.get('type')!r} is not supported by SentenceEncoder")
    return p


class SentenceEncoder:
    """Object with the ``.encode`` surface of ``sentence_transformers.SentenceTransformer``."""

    def __init__(self, model_name: str = "math-similarity/Bert-MLM_arXiv-MP-class_zbMath", device: Optional[str] = None,
                 dtype: Optional[torch.dtype] = None, seed: int = 0, num_layers: Optional[int] = None,
                 allow_random_init: Optional[bool] = None, trust_remote_code: bool = False, fp32_gemm: Optional[str] = None):
        """``fp32_gemm`` (fp32 models on a GPU): ``"blas"`` = the library's fp32 GEMMs (default: the reference's arithmetic,
        SentenceTransformer(name) without a dtype) or ``"bf16x3"`` = the same fp32 weights and activations as bf16 pieces on the
        bf16 matrix pipe (`fused_forward.pieces_linear`: sixteen significant bits per factor, fp32 accumulation; embeddings
        within the cosine DESIGN.md section 8 states of the fp32-GEMM forward); ``TS_ENCODER_FP32_GEMM`` sets the default."""
        self.model_name = model_name
        self.device = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
        fp32_gemm = fp32_gemm or os.environ.get("TS_ENCODER_FP32_GEMM", "") or "blas"
        if fp32_gemm not in ("blas", "bf16x3"):
            raise ValueError(f"fp32_gemm must be 'blas' or 'bf16x3', got {fp32_gemm!r}")
        self.fp32_gemm = fp32_gemm
        self._ctor = dict(model_name=model_name, dtype=dtype, seed=seed, num_layers=num_layers,
                          allow_random_init=allow_random_init, trust_remote_code=trust_remote_code, fp32_gemm=fp32_gemm)
        self._pool = None
        local = _local_dir(model_name)
        if allow_random_init is None:
            allow_random_init = os.environ.get("TS_ALLOW_RANDOM_ENCODER", "") not in ("", "0")
        if local is not None:
            from transformers import AutoModel, AutoTokenizer
            self.pipeline = read_pipeline(local)
            tdir = os.path.join(local, self.pipeline.transformer_path) if self.pipeline.transformer_path else local
            self.tokenizer = AutoTokenizer.from_pretrained(tdir, trust_remote_code=trust_remote_code)
            self.model = AutoModel.from_pretrained(tdir, trust_remote_code=trust_remote_code)
            self._hf_tokenizer = True
            self.pretrained = True
            limit = getattr(self.model.config, "max_position_embeddings", None)
            self.max_seq_length = self.pipeline.max_seq_length or limit or 512
            if limit:
                self.max_seq_length = min(self.max_seq_length, limit)
            if dtype is None:
                dtype = torch.float32                  # what the reference runs; bf16 is an opt-in
        else:
            if not allow_random_init:
                raise FileNotFoundError(
                    f"no local sentence-transformers checkpoint for {model_name!r} (looked at that path and under "
                    f"$TS_MODEL_DIR); nothing can be downloaded here.  Pass allow_random_init=True (or set "
                    f"TS_ALLOW_RANDOM_ENCODER=1) for the randomly initialised stand-in used by benchmarks and tests - "
                    f"its embeddings are meaningless")
            arch = ARCHITECTURES.get(model_name, DEFAULT_ARCH)
            pooling, max_len = arch["pooling"], arch["max_len"]
            gen_state = torch.random.get_rng_state()
            torch.manual_seed(seed)
            if arch["family"] == "qwen3":
                from transformers import Qwen3Config, Qwen3Model
                cfg = Qwen3Config(vocab_size=arch["vocab"], hidden_size=arch["hidden"], num_hidden_layers=num_layers or arch["layers"],
                                  num_attention_heads=arch["heads"], num_key_value_heads=arch["kv_heads"], head_dim=arch["head_dim"],
                                  intermediate_size=arch["ffn"], max_position_embeddings=max_len, rms_norm_eps=arch["rms_eps"],
                                  rope_theta=arch["rope_theta"], attention_bias=False, use_sliding_window=False)
                self.model = Qwen3Model(cfg)
            elif arch["family"] == "gemma3":
                from transformers import Gemma3TextConfig, Gemma3TextModel
                cfg = Gemma3TextConfig(vocab_size=arch["vocab"], hidden_size=arch["hidden"], num_hidden_layers=num_layers or arch["layers"],
                                       num_attention_heads=arch["heads"], num_key_value_heads=arch["kv_heads"], head_dim=arch["head_dim"],
                                       intermediate_size=arch["ffn"], max_position_embeddings=max_len, rms_norm_eps=arch["rms_eps"],
                                       sliding_window=arch["sliding_window"], query_pre_attn_scalar=arch["head_dim"],
                                       use_bidirectional_attention=True, attention_bias=False)
                self.model = Gemma3TextModel(cfg)
            else:
                from transformers import BertConfig, BertModel
                cfg = BertConfig(vocab_size=30522, hidden_size=arch["hidden"], num_hidden_layers=num_layers or arch["layers"],
                                 num_attention_heads=arch["heads"], intermediate_size=arch["ffn"],
                                 max_position_embeddings=min(max_len, 512))
                self.model = BertModel(cfg, add_pooling_layer=False)
            torch.random.set_rng_state(gen_state)
            self.tokenizer = HashingTokenizer(cfg.vocab_size, min(max_len, 512))
            self._hf_tokenizer = False
            self.pretrained = False
            self.pipeline = _Pipeline()
            self.pipeline.pooling = pooling
            if arch.get("dense"):                         # the checkpoint's Dense modules (no bias, identity activation) + Normalize
                gen_state = torch.random.get_rng_state()
                torch.manual_seed(seed + 1)
                width = arch["hidden"]
                for out_f in arch["dense"]:
                    self.pipeline.dense.append(torch.nn.Sequential(torch.nn.Linear(width, out_f, bias=False), torch.nn.Identity()))
                    width = out_f
                torch.random.set_rng_state(gen_state)
                self.pipeline.normalize = True
            self.max_seq_length = min(max_len, 512)
            if dtype is None:
                dtype = torch.float32                  # the stand-in runs what a real checkpoint runs; bf16 is an opt-in for both
        self.pooling = self.pipeline.pooling
        self.model.to(self.device, dtype=dtype).eval()
        # BERT-family and Qwen3-family models on a GPU run a fused forward (FusedBertForward / FusedQwen3Forward);
        # TS_ENCODER_FUSED=0 keeps the model's own
        self._fused = None
        if os.environ.get("TS_ENCODER_FUSED", "1") != "0" and self.device.type == "cuda":
            if FusedBertForward.covers(self.model):
                self._fused = FusedBertForward(self.model)
            elif FusedQwen3Forward.covers(self.model):
                self._fused = FusedQwen3Forward(self.model)
            elif FusedGemma3Forward.covers(self.model):
                self._fused = FusedGemma3Forward(self.model)
        if self._fused is not None and fp32_gemm == "bf16x3" and next(self.model.parameters()).dtype == torch.float32:
            self._fused.pieces = True
            self._fused._refresh()
        for m in self.pipeline.dense:
            m.to(self.device, dtype=torch.float32).eval()
        self.embedding_dim = (self.pipeline.dense[-1][0].out_features if self.pipeline.dense
                              else self.model.config.hidden_size)

    def get_sentence_embedding_dimension(self) -> int:
        return self.embedding_dim

    def eval(self):
        self.model.eval()
        return self

    # -- text -> tokens ----------------------------------------------------------------------------------------------
    def _prompted(self, texts: Sequence[str], prompt_name: Optional[str], prompt: Optional[str]) -> List[str]:
        if prompt is None:
            name = prompt_name if prompt_name is not None else self.pipeline.default_prompt_name
            if name is not None:
                if name not in self.pipeline.prompts:
                    raise ValueError(f"prompt name {name!r} is not in the checkpoint's prompts {sorted(self.pipeline.prompts)}")
                prompt = self.pipeline.prompts[name]
        if not prompt:
            return list(texts)
        if not self.pipeline.include_prompt:
            raise NotImplementedError("Pooling with include_prompt = false (prompt tokens excluded from the pooling)")
        return [prompt + t for t in texts]

    def _tokenize(self, texts: Sequence[str]):
        if self._hf_tokenizer:
            return self.tokenizer(list(texts), padding=True, truncation=True, max_length=self.max_seq_length,
                                  return_tensors="pt")
        return self.tokenizer(texts)

    # -- tokens -> embeddings ----------------------------------------------------------------------------------------
    @torch.inference_mode()
    def encode_device(self, texts: Sequence[str], batch_size: int = 32, normalize_embeddings: bool = False,
                      prompt_name: Optional[str] = None, prompt: Optional[str] = None) -> torch.Tensor:
        """fp32 ``[n x d]`` tensor on the model device (no host hop): what the index upload consumes."""
        texts = self._prompted(texts, prompt_name, prompt)
        order = np.argsort([-len(t) for t in texts], kind="stable")       # longest first, like sentence-transformers
        out = torch.empty((len(texts), self.embedding_dim), dtype=torch.float32, device=self.device)
        for start in range(0, len(texts), batch_size):
            sel = order[start:start + batch_size]
            tok = self._tokenize([texts[i] for i in sel])
            no_padding = bool(tok["attention_mask"].all())                  # decided on the host copy: no device sync
            enc = {k: v.to(self.device) for k, v in tok.items()}
            fwd = {k: v for k, v in enc.items() if k in ("input_ids", "attention_mask", "token_type_ids")}
            hidden = self.forward_hidden(**fwd, no_padding=no_padding)
            out[torch.as_tensor(sel, device=self.device)] = self.pool(hidden, enc["attention_mask"], normalize_embeddings)
        return out

    def forward_hidden(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, token_type_ids=None,
                       no_padding: bool = False) -> torch.Tensor:
        """Last hidden state ``[n x seq x d]`` of the transformer: the fused BERT forward where it applies, else the model's.
        ``no_padding=True``: the caller knows ``attention_mask`` is all ones (the attention then runs without a mask)."""
        if self._fused is not None:
            return self._fused(input_ids, attention_mask, token_type_ids, no_padding=no_padding)
        kw = {"input_ids": input_ids, "attention_mask": attention_mask}
        if token_type_ids is not None:
            kw["token_type_ids"] = token_type_ids
        return self.model(**kw).last_hidden_state

    def pool(self, hidden: torch.Tensor, attention_mask: torch.Tensor, normalize: bool,
             out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """Everything behind the transformer forward: Pooling, the checkpoint's Dense modules, Normalize -> fp32
        ``[n x d]``.  On a GPU pooling (cls / mean / last token) + L2 normalisation are ONE fused HIP kernel
        (``ts_pool_normalize``) instead of five torch ops; Dense modules, when the checkpoint has them, run between the
        pooling and the normalisation.  On CPU the torch expression of sentence-transformers' modules.
        ``out_dtype=torch.bfloat16``: the fused kernel rounds (to nearest even) on the way out, which is the form a bf16
        inner-product index multiplies - `TheoremIndex.search_device` then reads the tensor in place, no preparation
        launch ("writing straight into the query buffer consumed by ts_search")."""
        normalize = bool(normalize or self.pipeline.normalize)
        fused = hidden.is_cuda and hidden.dtype in (torch.float32, torch.bfloat16) and self.pooling in _POOL_CODES
        if out_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("out_dtype must be torch.float32 or torch.bfloat16")
        if out_dtype == torch.bfloat16 and not (fused and not self.pipeline.dense):
            return self.pool(hidden, attention_mask, normalize).to(torch.bfloat16)
        if fused:
            import ctypes as C
            from . import _ffi
            hidden = hidden.contiguous()
            mask = attention_mask.to(torch.int64).contiguous()
            n, seq, d = hidden.shape
            emb = torch.empty((n, d), dtype=out_dtype, device=hidden.device)
            norm_here = normalize and not self.pipeline.dense
            _ffi.check(_ffi.load().ts_pool_normalize(
                hidden.device.index or 0, C.c_void_p(hidden.data_ptr()), 1 if hidden.dtype == torch.bfloat16 else 0,
                C.c_void_p(mask.data_ptr()), n, seq, d, _POOL_CODES[self.pooling], 1 if norm_here else 0,
                C.c_void_p(emb.data_ptr()), 1 if out_dtype == torch.bfloat16 else 0, d,
                C.c_void_p(torch.cuda.current_stream(hidden.device).cuda_stream)))
            if not self.pipeline.dense:
                return emb
        else:
            emb = pool_reference(hidden.float(), attention_mask, self.pooling)
        for m in self.pipeline.dense:
            emb = m(emb)
        if normalize:
            emb = torch.nn.functional.normalize(emb, p=2, dim=1)
        return emb

    def similarity(self, embeddings1, embeddings2) -> torch.Tensor:
        """``SentenceTransformer.similarity(a, b)`` as the reference's experiments call it (experiments/first_experiment.py:195,
        205, second_experiment.py): the ``[len(a) x len(b)]`` fp32 matrix of the checkpoint's similarity function - cosine
        (``util.cos_sim``, the default) or dot product - as a torch tensor, computed by libtsearch (``ts_scores``).  Meant for
        the small shapes of those scripts; a search is ``TheoremIndex.search``."""
        from .index import TheoremIndex, _host_rows
        name = getattr(getattr(self, "pipeline", None), "similarity_fn_name", None) or "cosine"
        if name not in ("cosine", "dot"):
            raise NotImplementedError(f"similarity function {name!r}")
        a, b = _host_rows(embeddings1), _host_rows(embeddings2)
        if a.shape[1] != b.shape[1]:
            raise ValueError(f"dimension mismatch: {a.shape[1]} vs {b.shape[1]}")
        with TheoremIndex.from_embeddings(b, dtype="f32", metric="cos" if name == "cosine" else "ip") as ix:
            return torch.from_numpy(ix.scores(a))

    def encode(self, sentences: Union[str, Iterable[str]], batch_size: int = 32, show_progress_bar: Optional[bool] = None,
               convert_to_numpy: bool = True, convert_to_tensor: bool = False, normalize_embeddings: bool = False,
               prompt_name: Optional[str] = None, prompt: Optional[str] = None, **_ignored):
        single = isinstance(sentences, str)
        texts = [sentences] if single else list(sentences)
        if len(texts) == 0:
            empty = torch.empty((0, self.embedding_dim), dtype=torch.float32)
            return empty if convert_to_tensor else empty.numpy()
        emb = self.encode_device(texts, batch_size=batch_size, normalize_embeddings=normalize_embeddings,
                                 prompt_name=prompt_name, prompt=prompt)
        if single:
            emb = emb[0]
        if convert_to_tensor:
            return emb
        return emb.cpu().numpy()

    # -- replica fan-out (ec2/generate_embeddings/embeddings.py:32-38) ---------------------------------------------------
    def start_multi_process_pool(self, target_devices: Optional[Sequence[str]] = None) -> dict:
        """One worker process per target device (default: every visible GPU, or four CPU workers), each holding its own
        replica of this encoder - replicas only, nothing is sharded.  Mirrors
        ``SentenceTransformer.start_multi_process_pool``."""
        import torch.multiprocessing as mp
        if target_devices is None:
            n = torch.cuda.device_count()
            target_devices = [f"cuda:{i}" for i in range(n)] if n > 0 else ["cpu"] * 4
        ctx = mp.get_context("spawn")
        workers = []
        for dev in target_devices:
            inq, outq = ctx.Queue(), ctx.Queue()
            p = ctx.Process(target=_replica_main, args=(self._ctor, str(dev), inq, outq), daemon=True)
            p.start()
            workers.append((p, inq, outq))
        return {"workers": workers, "devices": list(target_devices)}

    @staticmethod
    def stop_multi_process_pool(pool: dict) -> None:
        for p, inq, _ in pool["workers"]:
            try:
                inq.put(None)
            except Exception:
                pass
        for p, _, _ in pool["workers"]:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()

    def encode_multi_process(self, sentences, pool: Optional[dict] = None, batch_size: int = 32,
                             normalize_embeddings: bool = False, show_progress_bar: Optional[bool] = None,
                             chunk_size: Optional[int] = None, **_ignored) -> np.ndarray:
        """The texts split into contiguous chunks, one per replica; results concatenated in the input order.
        ``pool=None`` (what the reference passes): a pool over every visible GPU is started on first use and kept on
        the encoder (`close` stops it); with fewer than two visible GPUs the call is `encode` in this process - the
        same numbers either way, every replica runs the same weights."""
        texts = list(sentences)
        if pool is None:
            if self._pool is None and torch.cuda.device_count() >= 2:
                self._pool = self.start_multi_process_pool()
            pool = self._pool
        if pool is None or len(pool["workers"]) < 2 or len(texts) == 0:
            return self.encode(texts, batch_size=batch_size, normalize_embeddings=normalize_embeddings)
        nw = len(pool["workers"])
        per = chunk_size or -(-len(texts) // nw)
        bounds = [(a, min(len(texts), a + per)) for a in range(0, len(texts), per)]
        for j, (a, b) in enumerate(bounds):
            pool["workers"][j % nw][1].put((j, texts[a:b], batch_size, normalize_embeddings))
        parts = {}
        import queue as _queue
        for j in range(len(bounds)):
            proc, _, outq = pool["workers"][j % nw]
            while True:
                # a replica that died without answering (killed for memory, a GPU fault, a crash while loading the model)
                # would leave a plain get() waiting forever: poll, and look at the process in between
                try:
                    idx, arr = outq.get(timeout=2.0)
                    break
                except _queue.Empty:
                    if not proc.is_alive():
                        dev = pool["devices"][j % nw]
                        self._drop_pool(pool)
                        raise RuntimeError(f"the encoder replica on {dev} died (exit code {proc.exitcode}) without answering; "
                                           "the pool has been stopped - the next call starts a fresh one")
            if isinstance(arr, Exception):
                self._drop_pool(pool)              # the other replicas may still hold chunks of this call: start afresh
                raise arr
            parts[idx] = arr
        return np.concatenate([parts[j] for j in range(len(bounds))], axis=0)

    def _drop_pool(self, pool: dict) -> None:
        """Stop every worker of a pool whose call failed (its queues may hold answers of that call)."""
        for p, _, _ in pool["workers"]:
            if p.is_alive():
                p.terminate()
        for p, _, _ in pool["workers"]:
            p.join(timeout=10)
        if pool is self._pool:
            self._pool = None

    def close(self) -> None:
        if self._pool is not None:
            self.stop_multi_process_pool(self._pool)
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pool_reference(hidden: torch.Tensor, attention_mask: torch.Tensor, mode: str) -> torch.Tensor:
    """sentence-transformers' Pooling module in torch (every supported mode)."""
    mask = attention_mask.unsqueeze(-1).to(hidden.dtype)
    if mode == "cls":
        return hidden[:, 0]
    if mode == "lasttoken":
        seq = attention_mask.shape[1]
        values, first_from_end = attention_mask.flip(1).max(1)
        first_from_end = torch.where(values == 0, torch.full_like(first_from_end, seq - 1), first_from_end)
        last = seq - first_from_end - 1
        return hidden[torch.arange(hidden.shape[0], device=hidden.device), last]
    if mode == "max":
        return hidden.masked_fill(mask == 0, -1e9).max(dim=1).values
    summed = (hidden * mask).sum(dim=1)
    count = mask.sum(dim=1).clamp(min=1e-9)
    if mode == "mean":
        return summed / count
    if mode == "mean_sqrt_len":
        return summed / count.sqrt()
    raise NotImplementedError(f"pooling mode {mode}")


def _replica_main(ctor: dict, device: str, inq, outq) -> None:
    """Worker process of `start_multi_process_pool`: builds its replica, then encodes the chunks it is handed."""
    try:
        enc = SentenceEncoder(device=device, **ctor)
    except Exception as e:            # report instead of dying silently: the parent would wait forever
        while True:
            item = inq.get()
            if item is None:
                return
            outq.put((item[0], RuntimeError(f"replica on {device} failed to start: {e!r}")))
    while True:
        item = inq.get()
        if item is None:
            return
        j, texts, batch_size, normalize = item
        try:
            outq.put((j, enc.encode(texts, batch_size=batch_size, normalize_embeddings=normalize)))
        except Exception as e:
            outq.put((j, RuntimeError(f"replica on {device}: {e!r}")))
