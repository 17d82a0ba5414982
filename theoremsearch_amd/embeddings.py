"""Mirror of ``parsed_papers_to_vector_rds/embeddings.py`` (reference lines 11-39).

Same two names, same signatures, same return type (``list[list[float]]`` of L2-normalised vectors).
Unlike the reference, the embedder is built once and cached instead of being re-instantiated on
every call (reference line 29).
"""
from __future__ import annotations

import functools

from .encoder import SentenceEncoder

MODEL_NAME = "math-similarity/Bert-MLM_arXiv-MP-class_zbMath"


@functools.lru_cache(maxsize=1)
def _get_embedder() -> SentenceEncoder:
    return SentenceEncoder(MODEL_NAME)


def embed_texts(texts_to_embed: list[str]) -> list[list[float]]:
    """Embeds a list of texts into L2-normalised vectors (reference lines 14-39)."""
    embedder = _get_embedder()
    all_embeddings = embedder.encode(texts_to_embed, convert_to_numpy=True, normalize_embeddings=True,
                                     show_progress_bar=False)
    return all_embeddings.tolist()
