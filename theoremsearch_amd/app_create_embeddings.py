"""Mirror of ``app_create_embeddings.py`` (reference lines 8-97) and of the loader in
``app_showcase_model.py:40-58``.

``create_embedding_library()`` reads ``PARSED_PAPERS_DIR/*.json`` (schema written by
``arxiv_analyzer_app_showcase.py:191-201``), builds the per-theorem metadata and ``text_to_embed``
strings exactly as the reference does, encodes the corpus and writes the same two files
(``corpus_embeddings.pt`` = fp32 ``[N x 768]`` tensor, not normalised; ``theorems_data.pkl``).
``load_embedding_index`` goes one step further than the reference's loader and puts the matrix into
HBM as a :class:`TheoremIndex` (normalised once, instead of inside every ``cos_sim`` call).
"""
from __future__ import annotations

import json
import os
import pickle

import torch

from .encoder import SentenceEncoder
from .index import TheoremIndex

# --- Configuration (reference lines 8-10) ---
MODEL_NAME = "math-similarity/Bert-MLM_arXiv-MP-class_zbMath"
PARSED_PAPERS_DIR = "./app_papers"
OUTPUT_DIR = "./app_embeds"


def build_global_context(data: dict) -> str:
    blocks = (("Global Notations", "global_notations"), ("Global Definitions", "global_definitions"),
              ("Global Assumptions", "global_assumptions"))
    return "\n\n".join(f"**{title}:**\n{data.get(key, '')}" for title, key in blocks)


def theorem_records(data: dict) -> list:
    """The twelve-key metadata dict per theorem (reference lines 57-70)."""
    context = build_global_context(data)
    out = []
    for theorem in data.get("theorems", []):
        out.append({
            "paper_title": data.get("title", "N/A"),
            "paper_url": data.get("url", ""),
            "authors": data.get("authors", []),
            "citations": data.get("citations", 0),
            "primary_math_tag": data.get("primary_math_tag", "N/A"),
            "year": data.get("year"),
            "source": data.get("source"),
            "journal_published": data.get("journal_published"),
            "type": theorem["type"],
            "content": theorem["content"],
            "global_context": context,
            "text_to_embed": f"{context}\n\n**{theorem['type'].capitalize()}:**\n{theorem['content']}",
        })
    return out


def read_parsed_papers(papers_dir: str) -> list:
    """Theorem records of every ``*.json`` under ``papers_dir``, in directory order; a file that cannot be read or lacks a
    field is skipped with one line on stderr (the reference warns and goes on, app_create_embeddings.py:73-74)."""
    import sys
    records = []
    for name in os.listdir(papers_dir):
        if not name.endswith(".json"):
            continue
        path = os.path.join(papers_dir, name)
        try:
            with open(path, "r", encoding="utf-8") as f:
                records.extend(theorem_records(json.load(f)))
        except Exception as e:                                   # noqa: BLE001
            print(f"skipped {path}: {e}", file=sys.stderr)
    return records


def write_library(out_dir: str, embeddings, records: list) -> tuple:
    """The two files the apps load (app_create_embeddings.py:85-93): the tensor as ``torch.save`` writes it, the records pickled."""
    os.makedirs(out_dir, exist_ok=True)
    paths = os.path.join(out_dir, "corpus_embeddings.pt"), os.path.join(out_dir, "theorems_data.pkl")
    torch.save(embeddings, paths[0])
    with open(paths[1], "wb") as f:
        pickle.dump(records, f)
    return paths


def create_embedding_library(model=None):
    """``PARSED_PAPERS_DIR/*.json`` -> ``OUTPUT_DIR/corpus_embeddings.pt`` + ``OUTPUT_DIR/theorems_data.pkl`` (the reference's
    contract: module constants in, two files out, ``None`` back; a missing model, directory or corpus ends the call with a
    message instead of an exception, as the script does).  ``model``: an encoder to reuse (default: ``SentenceEncoder(MODEL_NAME)``)."""
    if model is None:
        try:
            model = SentenceEncoder(MODEL_NAME)
        except Exception as e:                                   # noqa: BLE001
            print(f"cannot load {MODEL_NAME!r}: {e}")
            return
    if not os.path.isdir(PARSED_PAPERS_DIR):
        print(f"no parsed papers: {PARSED_PAPERS_DIR!r} is not a directory")
        return
    records = read_parsed_papers(PARSED_PAPERS_DIR)
    if not records:
        print(f"no theorems under {PARSED_PAPERS_DIR!r}: nothing written")
        return
    embeddings = model.encode([r["text_to_embed"] for r in records], convert_to_tensor=True, show_progress_bar=True)
    paths = write_library(OUTPUT_DIR, embeddings, records)
    print(f"{len(records)} theorems embedded -> {paths[0]}, {paths[1]}")


def load_embedding_library(directory):
    """``(embeddings tensor on CPU, theorems_data)`` or ``(None, None)`` (app_showcase_model.py:40-58)."""
    embeddings_path = os.path.join(directory, "corpus_embeddings.pt")
    data_path = os.path.join(directory, "theorems_data.pkl")
    if not os.path.exists(embeddings_path) or not os.path.exists(data_path):
        return None, None
    embeddings = torch.load(embeddings_path, map_location=torch.device("cpu"))
    with open(data_path, "rb") as f:
        theorems_data = pickle.load(f)
    return embeddings, theorems_data


def load_embedding_index(directory, dtype: str = "f32", device: int = 0):
    """The library as a device-resident cosine index plus its metadata."""
    embeddings, theorems_data = load_embedding_library(directory)
    if embeddings is None:
        return None, None
    return TheoremIndex.from_embeddings(embeddings, dtype=dtype, metric="cos", device=device), theorems_data


if __name__ == "__main__":
    create_embedding_library()
