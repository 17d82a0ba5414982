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


def create_embedding_library(model=None):
    """Builds ``OUTPUT_DIR/corpus_embeddings.pt`` and ``OUTPUT_DIR/theorems_data.pkl``."""
    print("Starting the embedding library creation process...")
    print(f"Loading sentence transformer model: '{MODEL_NAME}'...")
    if model is None:
        try:
            model = SentenceEncoder(MODEL_NAME)
        except Exception as e:
            print(f"Error loading model: {e}")
            return
    if not os.path.exists(PARSED_PAPERS_DIR):
        print(f"Error: The directory '{PARSED_PAPERS_DIR}' was not found.")
        return
    json_files = [os.path.join(PARSED_PAPERS_DIR, f) for f in os.listdir(PARSED_PAPERS_DIR) if f.endswith(".json")]
    if not json_files:
        print(f"No parsed JSON files found in '{PARSED_PAPERS_DIR}'.")
        return
    print(f"Found {len(json_files)} parsed paper(s). Loading and preparing data for embedding...")
    all_theorems_data = []
    for file_path in json_files:
        try:
            with open(file_path, "r", encoding="utf-8") as f:
                all_theorems_data.extend(theorem_records(json.load(f)))
        except Exception as e:
            print(f"Warning: Could not process file {file_path}. Error: {e}")
    if not all_theorems_data:
        print("No theorems were extracted from the JSON files. Aborting.")
        return
    print(f"Embedding {len(all_theorems_data)} total theorems. This may take a while...")
    corpus_texts = [item["text_to_embed"] for item in all_theorems_data]
    corpus_embeddings = model.encode(corpus_texts, convert_to_tensor=True, show_progress_bar=True)
    os.makedirs(OUTPUT_DIR, exist_ok=True)
    embeddings_path = os.path.join(OUTPUT_DIR, "corpus_embeddings.pt")
    data_path = os.path.join(OUTPUT_DIR, "theorems_data.pkl")
    print(f"Saving embeddings tensor to '{embeddings_path}'...")
    torch.save(corpus_embeddings, embeddings_path)
    print(f"Saving theorem metadata to '{data_path}'...")
    with open(data_path, "wb") as f:
        pickle.dump(all_theorems_data, f)
    print("\nEmbedding library created successfully!")
    print(f"   - {len(all_theorems_data)} theorems embedded.")
    print(f"   - Files saved in the '{OUTPUT_DIR}' directory.")


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
