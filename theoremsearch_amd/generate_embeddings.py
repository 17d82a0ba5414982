"""Mirror of ``ec2/generate_embeddings/{embedders,embeddings}.py`` (the production embedding job's
two helpers) plus a device-resident variant that feeds the search index without a host hop.

    EMBEDDERS                                  embedders.py:1-4
    get_embedder(embedder_alias)               embeddings.py:10-14
    embed_texts(embedder, texts, batch_size)   embeddings.py:16-40  -> list[list[float]], normalised

The RDS paging / upsert driver around them (``__main__.py:10-105``) is out of scope (network I/O);
``embed_into_index`` is what replaces its ``upsert_rows`` step when the destination is the HBM index.
"""
from __future__ import annotations

from .encoder import SentenceEncoder
from .index import TheoremIndex

EMBEDDERS = {
    "qwen": "Qwen/Qwen3-Embedding-0.6B",
    "gemma": "google/embeddinggemma-300m",
}


def get_embedder(embedder_alias: str) -> SentenceEncoder:
    """``SentenceTransformer(EMBEDDERS[alias], device=...)`` + ``eval()``: needs the checkpoint in a local directory
    (``$TS_MODEL_DIR/<name>``); raises when it is absent (no silent stand-in)."""
    model = SentenceEncoder(EMBEDDERS[embedder_alias])
    model.eval()
    return model


def embed_texts(embedder, texts_to_embed: list[str], batch_size: int = 16):
    """Normalised embeddings as ``list[list[float]]`` (reference returns ``embeddings.tolist()``): a page smaller
    than the batch size is encoded in this process, a larger one is fanned out over one replica per visible GPU
    (``encode_multi_process(pool=None)``, embeddings.py:24-38)."""
    if len(texts_to_embed) < batch_size:
        embeddings = embedder.encode(texts_to_embed, normalize_embeddings=True, show_progress_bar=False,
                                     batch_size=batch_size)
    else:
        embeddings = embedder.encode_multi_process(texts_to_embed, pool=None, normalize_embeddings=True,
                                                   show_progress_bar=False, batch_size=batch_size)
    return embeddings.tolist()


def embed_into_index(embedder: SentenceEncoder, index: TheoremIndex, texts: list[str], row0,
                     batch_size: int = 16):
    """Encode ``texts`` and store them as index rows ``[row0, row0 + len(texts))`` (upsert-by-position,
    the HBM counterpart of ``ON CONFLICT (slogan_id) DO UPDATE``, __main__.py:85-99); ``row0=None`` appends them
    behind the last row (the INSERT half of the upsert) and returns the id of the first new row.  On a GPU the
    encoder output goes device-to-device through ``ts_index_upload_device`` / ``ts_index_append_device`` on the
    stream the encoder ran on (a default-stream handle of 0 means the index's own stream, which is ordered with
    the legacy default stream; any other stream is used as is), and the call returns after the rows have landed."""
    emb = embedder.encode_device(texts, batch_size=batch_size, normalize_embeddings=True)
    first = row0
    if emb.is_cuda:
        import torch
        stream = torch.cuda.current_stream(emb.device).cuda_stream
        if row0 is None:
            first = index.append_device(emb.data_ptr(), "f32", emb.stride(0), emb.shape[0], stream)
        else:
            index.upload_device(emb.data_ptr(), "f32", emb.stride(0), row0, emb.shape[0], stream)
        index.synchronize()          # `emb` is released when this function returns: the upload must have read it
    elif row0 is None:
        first = index.append(emb.numpy())
    else:
        index.upload(emb.numpy(), row0)
    return first


def generate_embeddings(pages, embedder_alias: str, index: TheoremIndex, slot_of=None, batch_size: int = 16,
                        overwrite: bool = False, embedder=None, slots: dict | None = None) -> int:
    """The loop of ``ec2/generate_embeddings/__main__.py:10-105`` with the RDS calls factored out.

    The reference pages ``{"slogan_id", "slogan"}`` rows out of Postgres with keyset pagination
    (page size 128, ``__main__.py:70-77``), skips slogans that already have an embedding unless
    ``--overwrite`` (``:32-41``), embeds each page (``:78``) and upserts ``{"slogan_id", "embedding"}``
    into ``theorem_embedding_<alias>`` with ``ON CONFLICT (slogan_id) DO UPDATE`` (``:85-99``).
    Here ``pages`` is any iterable of such pages (the database cursor stays the caller's business),
    the destination is the HBM index, and ``slot_of(slogan_id) -> row`` is the upsert key (default:
    the slogan id is the row).  Rows are written device-to-device.  Returns the number of rows embedded.

    ``slots`` (a dict ``slogan_id -> row``, kept by the caller across runs) switches to the growing form: a
    slogan_id it knows is an UPDATE of its row (skipped unless ``overwrite``), one it does not know is an INSERT -
    its embedding is appended behind the last row (``ts_index_append_device``) and recorded in ``slots``; the index
    needs no rebuild when new slogans arrive.
    """
    if embedder is None:
        embedder = get_embedder(embedder_alias)
    if slots is not None:
        n_done = 0
        for page in pages:
            fresh = [r for r in page if r["slogan_id"] not in slots]
            if fresh:
                first = embed_into_index(embedder, index, [r["slogan"] for r in fresh], None, batch_size=batch_size)
                for j, r in enumerate(fresh):
                    slots[r["slogan_id"]] = first + j
                n_done += len(fresh)
            if overwrite:
                new_ids = {r["slogan_id"] for r in fresh}
                for r in page:
                    if r["slogan_id"] not in new_ids:
                        embed_into_index(embedder, index, [r["slogan"]], slots[r["slogan_id"]] - index.row_offset,
                                         batch_size=batch_size)
                        n_done += 1
        return n_done
    slot_of = slot_of or (lambda slogan_id: int(slogan_id))
    seen = getattr(index, "_filled", None)
    if seen is None:
        seen = index._filled = set()
    n_done = 0
    for page in pages:
        todo = [r for r in page if overwrite or slot_of(r["slogan_id"]) not in seen]
        if not todo:
            continue
        todo.sort(key=lambda r: slot_of(r["slogan_id"]))
        # contiguous runs of slots go down in one upload each
        start = 0
        while start < len(todo):
            end = start + 1
            while end < len(todo) and slot_of(todo[end]["slogan_id"]) == slot_of(todo[end - 1]["slogan_id"]) + 1:
                end += 1
            run = todo[start:end]
            embed_into_index(embedder, index, [r["slogan"] for r in run], slot_of(run[0]["slogan_id"]), batch_size=batch_size)
            seen.update(slot_of(r["slogan_id"]) for r in run)
            n_done += len(run)
            start = end
    return n_done
