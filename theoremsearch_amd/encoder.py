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
checkpoint's own ``max_seq_length``; ``dtype=torch.bfloat16`` is an opt-in.

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
                 allow_random_init: Optional[bool] = None, trust_remote_code: bool = False):
        self.model_name = model_name
        self.device = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
        self._ctor = dict(model_name=model_name, dtype=dtype, seed=seed, num_layers=num_layers,
                          allow_random_init=allow_random_init, trust_remote_code=trust_remote_code)
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
                dtype = torch.bfloat16 if self.device.type == "cuda" else torch.float32
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
        stamp = tuple((p.data_ptr(), p._version, p.dtype) for p in self._sources())
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
        mask = None if (no_padding or short) else torch.zeros((B, 1, 1, S), dtype=x.dtype, device=x.device).masked_fill_(
            ~attention_mask[:, None, None, :].to(torch.bool), torch.finfo(x.dtype).min)
        key_mask = None if (no_padding or not short) else attention_mask.to(torch.int64).contiguous()
        for L in self.layers:
            qkv = F.linear(x, L["wqkv"], L["bqkv"])
            if short:
                ctx = self._attention(qkv, key_mask, B, S)
            else:
                qkv = qkv.view(B, S, 3, self.heads, hd).permute(2, 0, 3, 1, 4)
                ctx = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], attn_mask=mask)
                ctx = ctx.transpose(1, 2).reshape(B, S, H)
            x = self._add_ln(F.linear(ctx, L["wo"], L["bo"]), x, L["ln1"])
            h = self.act(F.linear(x, L["w1"], L["b1"]))
            x = self._add_ln(F.linear(h, L["w2"], L["b2"]), x, L["ln2"])
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
        self._stamp = None
        self._gqa_native = True
        self._refresh()

    def _sources(self):
        for layer in self.model.layers:
            att, mlp = layer.self_attn, layer.mlp
            yield from (att.q_proj.weight, att.k_proj.weight, att.v_proj.weight, mlp.gate_proj.weight, mlp.up_proj.weight)

    def _refresh(self):
        stamp = tuple((p.data_ptr(), p._version, p.dtype) for p in self._sources())
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
        key_mask = None if (no_padding or not short) else attention_mask.to(torch.int64).contiguous()
        if not no_padding and not short:
            # causal, and padding keys are never attended to (the most negative finite value: rows of padding stay finite)
            neg = torch.finfo(x.dtype).min
            causal = torch.ones((S, S), dtype=torch.bool, device=x.device).tril_()
            keep = causal[None, None] & attention_mask[:, None, None, :].to(torch.bool)
            mask = torch.zeros((B, 1, S, S), dtype=x.dtype, device=x.device).masked_fill_(~keep, neg)
        nq, nkv, hd = self.hq * self.hd, self.hkv * self.hd, self.hd
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
        self._stamp = None
        self._gqa_native = True
        self._refresh()

    def _sources(self):
        for layer in self.model.layers:
            att, mlp = layer.self_attn, layer.mlp
            yield from (att.q_proj.weight, att.k_proj.weight, att.v_proj.weight, mlp.gate_proj.weight, mlp.up_proj.weight)

    def _refresh(self):
        stamp = tuple((p.data_ptr(), p._version, p.dtype) for p in self._sources())
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
        if not no_padding:
            neg = torch.finfo(x.dtype).min
            mask = torch.zeros((B, 1, 1, S), dtype=x.dtype, device=x.device).masked_fill_(~attention_mask[:, None, None, :].to(torch.bool), neg)
        nq, nkv, hd = self.hq * self.hd, self.hkv * self.hd, self.hd
        h = self._norm(None, x, None, self.layers[0]["ln_in"], False)[1]
        for li, L in enumerate(self.layers):
            qkv = F.linear(h, L["wqkv"])
            cos, sin = tables[L["type"]]
            _ffi.check(lib.ts_gemma_qk_norm_rope(dev, C.c_void_p(qkv.data_ptr()), C.c_void_p(L["qn"].data_ptr()), C.c_void_p(L["kn"].data_ptr()),
                                                 C.c_void_p(cos.data_ptr()), C.c_void_p(sin.data_ptr()), self.eps, B * S, S, self.hq, self.hkv,
                                                 hd, dt, stream))
            q = qkv[..., :nq].view(B, S, self.hq, hd).transpose(1, 2)
            k = qkv[..., nq:nq + nkv].view(B, S, self.hkv, hd).transpose(1, 2)
            v = qkv[..., nq + nkv:].view(B, S, self.hkv, hd).transpose(1, 2)
            ctx = None
            if self._gqa_native:
                try:
                    ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, scale=self.scaling, enable_gqa=True)
                except (RuntimeError, TypeError):
                    self._gqa_native = False
            if ctx is None:
                rep = self.hq // self.hkv
                ctx = F.scaled_dot_product_attention(q, k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1),
                                                     attn_mask=mask, scale=self.scaling)
            ctx = ctx.transpose(1, 2).reshape(B, S, nq)
            x, h = self._norm(F.linear(ctx, L["wo"]), x, L["ln_post_attn"], L["ln_pre_ffn"], True)
            gu = F.linear(h, L["wgu"])
            inter = gu.shape[-1] // 2
            act = torch.empty((B, S, inter), dtype=x.dtype, device=x.device)
            _ffi.check(lib.ts_geglu(dev, C.c_void_p(gu.data_ptr()), B * S, inter, dt, C.c_void_p(act.data_ptr()), stream))
            last = li + 1 == len(self.layers)
            w_next = m.norm.weight if last else self.layers[li + 1]["ln_in"]
            x, h = self._norm(F.linear(act, L["wd"]), x, L["ln_post_ffn"], w_next, not last)
        return h


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
