"""The search lines of the reference's showcase app (``app_showcase_model.py:92-129``)::

    query_emb     = model.encode(query, convert_to_tensor=True)
    cosine_scores = util.cos_sim(query_emb, embeddings_db)[0]
    top_indices   = torch.topk(cosine_scores, k=min(200, len(theorems_data)), sorted=True).indices
    for idx in top_indices: ... the sidebar's predicates ... until filters['top_k'] results

as ONE filtered search of a device-resident index (`filters.search_filtered`: the predicates evaluated once to one bit per
row, the bit tested inside the top-k kernel).  `search_and_display` stops where the reference's ``filtered_results`` list is
complete and RETURNS it; rendering it (sub-header, expanders, LaTeX clean-up: ``app_showcase_model.py:131-157``) stays the
app's own code.  The list is the ``top_k`` best rows that pass the filters - the reference's list whenever its pool of 200
holds that many, and complete where that pool runs dry.
"""
from __future__ import annotations

from . import filters as _filters
from .index import TheoremIndex

MODEL_NAME = "math-similarity/Bert-MLM_arXiv-MP-class_zbMath"      # app_showcase_model.py:10
EMBEDDING_LIBRARY_DIR = "./app_embeds"                               # app_showcase_model.py:11


def load_model():
    """The embedding model of the app (``app_showcase_model.py:32-38``); raises what went wrong (the app shows it)."""
    from .encoder import SentenceEncoder
    return SentenceEncoder(MODEL_NAME)


def load_embedding_library(directory):
    """``(embeddings, theorems_data)`` of ``corpus_embeddings.pt`` + ``theorems_data.pkl`` (``app_showcase_model.py:41-58``),
    ``(None, None)`` when the directory does not hold them."""
    from . import app_create_embeddings as ace
    return ace.load_embedding_library(directory)


def search_and_display(query, model, theorems_data, embeddings_db, filters, mask=None):
    """The ``filtered_results`` of ``app_showcase_model.py:92-129``: ``[{"info": theorems_data[row], "similarity": float}]``,
    best first, at most ``filters["top_k"]`` entries; ``None`` where the reference returns before searching (no query, no
    source selected).  ``embeddings_db``: a `TheoremIndex` (kept across calls) or the ``[N x d]`` matrix / tensor
    ``load_embedding_library`` returns (indexed for this call); ``mask``: the ``filters.filter_mask`` of this sidebar state
    when the caller keeps it across queries."""
    if not query or not filters["sources"]:
        return None
    query_emb = model.encode(query, convert_to_tensor=True)
    own = not isinstance(embeddings_db, TheoremIndex)
    index = TheoremIndex.from_embeddings(embeddings_db, metric="cos") if own else embeddings_db
    try:
        return _filters.search_filtered(index, query_emb, theorems_data, filters, mask)
    finally:
        if own:
            index.close()
