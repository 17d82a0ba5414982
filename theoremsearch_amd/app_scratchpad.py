"""Mirror of the data preparation (``app_scratchpad.py:23-63``) and of the search call site of the reference's scratch app
(``app_scratchpad.py:120-154``)::

    query_emb      = model.encode(query, convert_to_tensor=True)
    cosine_scores  = util.cos_sim(query_emb, embeddings_db)[0]
    top_indices    = np.argsort(-cosine_scores.cpu())[:5]

as one fused top-5 search of a device-resident index; what it hands to streamlit (sub-header, one expander per hit
titled ``**Result i | Similarity: s | Type: T**``, paper line, source link, context block quote, statement) is the
reference's.  The streamlit module and the LaTeX clean-up function of the app (UI, out of scope here) are passed in.
"""
from __future__ import annotations

import json

from .index import TheoremIndex


def load_and_prepare_data(paper_files, st=None):
    """Mirror of ``app_scratchpad.py:23-63``: the theorem records of the scratch app from parsed-paper JSON files -
    ``paper_title``, ``paper_url``, ``type``, ``content``, ``global_context`` (the paper's global notations / definitions /
    assumptions under their bold headings) and ``text_to_embed`` = context, blank line, ``**Type:**``, statement: the very
    strings the corpus embeddings are made of.  A missing or undecodable file is reported through ``st.warning`` (when a
    streamlit module is passed) and skipped, as there."""
    all_theorems_data = []
    for file_path in paper_files:
        try:
            with open(file_path, "r", encoding="utf-8") as f:
                data = json.load(f)
        except FileNotFoundError:
            if st is not None:
                st.warning(f"Warning: The data file '{file_path}' was not found.")
            continue
        except json.JSONDecodeError:
            if st is not None:
                st.warning(f"Warning: Could not decode JSON from {file_path}.")
            continue
        parts = []
        for key, title in (("global_notations", "Global Notations"), ("global_definitions", "Global Definitions"),
                           ("global_assumptions", "Global Assumptions")):
            if data.get(key, ""):
                parts.append(f"**{title}:**\n{data[key]}")
        global_context = "\n\n".join(parts)
        for theorem in data.get("theorems", []):
            all_theorems_data.append({
                "paper_title": data.get("title", "N/A"),
                "paper_url": data.get("url", ""),
                "type": theorem["type"],
                "content": theorem["content"],
                "global_context": global_context,
                "text_to_embed": f"{global_context}\n\n**{theorem['type'].capitalize()}:**\n{theorem['content']}",
            })
    return all_theorems_data


def search_theorems(query, model, theorems_data, embeddings_db, st, clean_latex_for_display=lambda text: text):
    """Finds and displays the top 5 most similar theorems.  ``embeddings_db``: a `TheoremIndex` (kept across calls) or
    the ``[N x d]`` matrix / tensor ``load_embedding_library`` returns (indexed for this call)."""
    if not query:
        st.info("Please enter a search query.")
        return
    query_emb = model.encode(query, convert_to_tensor=True)
    own = not isinstance(embeddings_db, TheoremIndex)
    index = TheoremIndex.from_embeddings(embeddings_db, metric="cos") if own else embeddings_db
    try:
        scores, top_indices = index.search(query_emb, min(5, index.n))
    finally:
        if own:
            index.close()
    st.subheader("Top 5 Most Similar Theorems")
    for i, (idx, similarity) in enumerate(zip(top_indices[0], scores[0])):
        if idx < 0:
            continue
        info = theorems_data[int(idx) - index.row_offset]
        expander_title = (
            f"**Result {i+1} | Similarity: {float(similarity):.4f} | "
            f"Type: {info['type'].capitalize()}**"
        )
        with st.expander(expander_title):
            st.markdown(f"**Paper:** *{info['paper_title']}*")
            st.markdown(f"**Source:** [{info['paper_url']}]({info['paper_url']})")
            if info["global_context"]:
                cleaned_ctx = clean_latex_for_display(info["global_context"])
                blockquote_ctx = "> " + cleaned_ctx.replace("\n", "\n> ")
                st.markdown(blockquote_ctx)
                st.write("")
            cleaned_content = clean_latex_for_display(info["content"])
            st.markdown(cleaned_content)
