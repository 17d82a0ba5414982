"""Mirror of the search call site of the reference's scratch app (``app_scratchpad.py:120-154``)::

    query_emb      = model.encode(query, convert_to_tensor=True)
    cosine_scores  = util.cos_sim(query_emb, embeddings_db)[0]
    top_indices    = np.argsort(-cosine_scores.cpu())[:5]

as one fused top-5 search of a device-resident index; what it hands to streamlit (sub-header, one expander per hit
titled ``**Result i | Similarity: s | Type: T**``, paper line, source link, context block quote, statement) is the
reference's.  The streamlit module and the LaTeX clean-up function of the app (UI, out of scope here) are passed in.
"""
from __future__ import annotations

from .index import TheoremIndex


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
