"""Sentence encoder with the ``SentenceTransformer.encode`` call shape used by the reference.

PyTorch-ROCm runs the transformer forward (the only place this package uses torch for compute);
pooling + optional L2 normalisation follow sentence-transformers' semantics.  Call sites mirrored:

    model.encode(texts, convert_to_numpy=True, normalize_embeddings=True, show_progress_bar=False)
                                                        parsed_papers_to_vector_rds/embeddings.py:31-37
    embedder.encode(texts, normalize_embeddings=True, show_progress_bar=False, batch_size=16)
                                                        ec2/generate_embeddings/embeddings.py:24-30
    model.encode(corpus_texts, convert_to_tensor=True, show_progress_bar=True)   app_create_embeddings.py:81
    model.encode(query, convert_to_tensor=True)                                   app_showcase_model.py:92
    model.encode(query or "", normalize_embeddings=True, convert_to_numpy=True)   streamlit_app.py:173

No model weights can be downloaded here.  ``SentenceEncoder(name)`` therefore loads real weights and
tokenizer only when ``name`` is a local directory (or ``TS_MODEL_DIR/<name>`` exists) readable by
``transformers``; otherwise it builds a randomly initialised model of the same architecture family
and width (seeded, so runs are reproducible) with a hashing word-piece stand-in tokenizer.  Numerical
parity with the published checkpoints is therefore unpinned; shapes, pooling,
normalisation and the call surface are what is tested.
"""
from __future__ import annotations

import os
import re
import zlib
from typing import Iterable, List, Optional, Sequence, Union

import numpy as np
import torch

# name -> (hidden, layers, heads, ffn, pooling, max_seq_length)
ARCHITECTURES = {
    "math-similarity/Bert-MLM_arXiv-MP-class_zbMath": (768, 12, 12, 3072, "mean", 512),
    "google/embeddinggemma-300m": (768, 24, 12, 3072, "mean", 2048),
    "Qwen/Qwen3-Embedding-0.6B": (1024, 28, 16, 3072, "last", 8192),
}
DEFAULT_ARCH = (768, 12, 12, 3072, "mean", 512)

_TOKEN_RE = re.compile(r"\\[A-Za-z]+|[A-Za-z]+|\d+|[^\sA-Za-z\d]")


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
        if cand and os.path.isdir(cand) and os.path.exists(os.path.join(cand, "config.json")):
            return cand
    return None


class SentenceEncoder:
    """Object with the ``.encode`` surface of ``sentence_transformers.SentenceTransformer``."""

    def __init__(self, model_name: str = "math-similarity/Bert-MLM_arXiv-MP-class_zbMath", device: Optional[str] = None,
                 dtype: Optional[torch.dtype] = None, seed: int = 0, num_layers: Optional[int] = None):
        self.model_name = model_name
        self.device = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
        hidden, layers, heads, ffn, pooling, max_len = ARCHITECTURES.get(model_name, DEFAULT_ARCH)
        self.pooling, self.max_seq_length = pooling, max_len
        local = _local_dir(model_name)
        if local is not None:
            from transformers import AutoModel, AutoTokenizer
            self.tokenizer = AutoTokenizer.from_pretrained(local)
            self.model = AutoModel.from_pretrained(local)
            self._hf_tokenizer = True
            self.pretrained = True
        else:
            from transformers import BertConfig, BertModel
            cfg = BertConfig(vocab_size=30522, hidden_size=hidden, num_hidden_layers=num_layers or layers,
                             num_attention_heads=heads, intermediate_size=ffn,
                             max_position_embeddings=min(max_len, 512))
            gen_state = torch.random.get_rng_state()
            torch.manual_seed(seed)
            self.model = BertModel(cfg, add_pooling_layer=False)
            torch.random.set_rng_state(gen_state)
            self.tokenizer = HashingTokenizer(cfg.vocab_size, min(max_len, 512))
            self._hf_tokenizer = False
            self.pretrained = False
        if dtype is None:
            dtype = torch.bfloat16 if self.device.type == "cuda" else torch.float32
        self.model.to(self.device, dtype=dtype).eval()
        self.embedding_dim = self.model.config.hidden_size

    def get_sentence_embedding_dimension(self) -> int:
        return self.embedding_dim

    def eval(self):
        self.model.eval()
        return self

    def _tokenize(self, texts: Sequence[str]):
        if self._hf_tokenizer:
            return self.tokenizer(list(texts), padding=True, truncation=True, max_length=min(self.max_seq_length, 512),
                                  return_tensors="pt")
        return self.tokenizer(texts)

    @torch.inference_mode()
    def encode_device(self, texts: Sequence[str], batch_size: int = 32, normalize_embeddings: bool = False) -> torch.Tensor:
        """fp32 ``[n x d]`` tensor on the model device (no host hop): what the index upload consumes."""
        order = np.argsort([-len(t) for t in texts], kind="stable")       # longest first, like sentence-transformers
        out = torch.empty((len(texts), self.embedding_dim), dtype=torch.float32, device=self.device)
        for start in range(0, len(texts), batch_size):
            sel = order[start:start + batch_size]
            enc = {k: v.to(self.device) for k, v in self._tokenize([texts[i] for i in sel]).items()}
            hidden = self.model(**enc).last_hidden_state
            out[torch.as_tensor(sel, device=self.device)] = self.pool(hidden, enc["attention_mask"], normalize_embeddings)
        return out

    def pool(self, hidden: torch.Tensor, attention_mask: torch.Tensor, normalize: bool) -> torch.Tensor:
        """Pooling + optional L2 normalisation of a last hidden state -> fp32 ``[n x d]``.  On a GPU this is ONE
        fused HIP kernel (``ts_pool_normalize``) instead of five torch ops; on CPU the torch expression of
        sentence-transformers' Pooling + Normalize modules."""
        if hidden.is_cuda and hidden.dtype in (torch.float32, torch.bfloat16):
            import ctypes as C
            from . import _ffi
            hidden = hidden.contiguous()
            mask = attention_mask.to(torch.int64).contiguous()
            n, seq, d = hidden.shape
            emb = torch.empty((n, d), dtype=torch.float32, device=hidden.device)
            _ffi.check(_ffi.load().ts_pool_normalize(
                hidden.device.index or 0, C.c_void_p(hidden.data_ptr()), 1 if hidden.dtype == torch.bfloat16 else 0,
                C.c_void_p(mask.data_ptr()), n, seq, d, 1 if self.pooling == "last" else 0, 1 if normalize else 0,
                C.c_void_p(emb.data_ptr()), 0, d, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            return emb
        hidden = hidden.float()
        mask = attention_mask.unsqueeze(-1).float()
        if self.pooling == "last":
            last = attention_mask.sum(dim=1) - 1
            emb = hidden[torch.arange(hidden.shape[0], device=hidden.device), last]
        else:
            emb = (hidden * mask).sum(dim=1) / mask.sum(dim=1).clamp(min=1e-9)
        if normalize:
            emb = torch.nn.functional.normalize(emb, p=2, dim=1)
        return emb

    def encode(self, sentences: Union[str, Iterable[str]], batch_size: int = 32, show_progress_bar: Optional[bool] = None,
               convert_to_numpy: bool = True, convert_to_tensor: bool = False, normalize_embeddings: bool = False,
               **_ignored):
        single = isinstance(sentences, str)
        texts = [sentences] if single else list(sentences)
        if len(texts) == 0:
            empty = torch.empty((0, self.embedding_dim), dtype=torch.float32)
            return empty if convert_to_tensor else empty.numpy()
        emb = self.encode_device(texts, batch_size=batch_size, normalize_embeddings=normalize_embeddings)
        if single:
            emb = emb[0]
        if convert_to_tensor:
            return emb
        return emb.cpu().numpy()

    # ec2/generate_embeddings/embeddings.py:32 uses the multi-process variant for big pages; one
    # process per GPU is this package's model, so it is the same single-device path.
    def encode_multi_process(self, sentences, pool=None, batch_size: int = 32, normalize_embeddings: bool = False,
                             show_progress_bar: Optional[bool] = None, **_ignored):
        return self.encode(sentences, batch_size=batch_size, normalize_embeddings=normalize_embeddings)
