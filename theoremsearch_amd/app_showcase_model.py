"""Mirror of the search function of the reference's showcase app (``app_showcase_model.py:79-156``)::

    query_emb     = model.encode(query, convert_to_tensor=True)
    cosine_scores = util.cos_sim(query_emb, embeddings_db)[0]
    top_indices   = torch.topk(cosine_scores, k=min(200, len(theorems_data)), sorted=True).indices
    for idx in top_indices: ... the sidebar's predicates ... until filters['top_k'] results

as ONE filtered search of a device-resident index (`filters.search_filtered`: the predicates evaluated once to one bit per
row, the bit tested inside the top-k kernel), so the list is the ``top_k`` best rows that pass - the reference's list whenever
its pool of 200 holds that many, and complete where the pool runs dry.  What it hands to streamlit (the two early exits, the
sub-header with the count, one expander per hit titled ``**Result i | Similarity: s | Type: T**``, paper / authors / source /
tag lines, the note, the context block quote, the statement) is the reference's.  The streamlit module and the LaTeX clean-up
function of the app (UI, out of scope here) are passed in.  ``load_embedding_library`` / ``load_embedding_index`` live in
`app_create_embeddings`.
"""
from __future__ import annotations

from . import filters as _filters
from .index import TheoremIndex

MODEL_NAME = "math-similarity/Bert-MLM_arXiv-MP-class_zbMath"      # app_showcase_model.py:10
EMBEDDING_LIBRARY_DIR = "./app_embeds"                               # app_showcase_model.py:11
ALLOWED_TYPES = ["theorem", "lemma", "proposition", "corollary", "definition", "remark", "assumption"]   # :27-29


def load_model(st=None):
    """``app_showcase_model.py:32-38``: the embedding model, or None with the reason shown through ``st.error`` (when a
    streamlit module is passed).  One instance serves every session thread (the app wraps this in ``st.cache_resource``)."""
    from .encoder import SentenceEncoder
    try:
        return SentenceEncoder(MODEL_NAME)
    except Exception as e:                 # noqa: BLE001 - the app shows whatever went wrong and carries on without a model
        if st is not None:
            st.error(f"Error loading embedding model: {e}")
        return None


def load_embedding_library(directory, st=None):
    """``app_showcase_model.py:41-58``: ``(embeddings, theorems_data)`` from ``corpus_embeddings.pt`` + ``theorems_data.pkl``,
    or ``(None, None)`` with the app's messages through ``st``."""
    import os
    from . import app_create_embeddings as ace
    if not os.path.exists(os.path.join(directory, "corpus_embeddings.pt")) or not os.path.exists(os.path.join(directory, "theorems_data.pkl")):
        if st is not None:
            st.error(f"Error: Embedding library not found in '{directory}'.")
            st.info("Please run the `app_create_embeddings.py` script first to generate the necessary files.")
        return None, None
    try:
        return ace.load_embedding_library(directory)
    except Exception as e:                 # noqa: BLE001
        if st is not None:
            st.error(f"Error loading files from the embedding library: {e}")
        return None, None


def search_and_display(query, model, theorems_data, embeddings_db, filters, st, clean_latex_for_display=lambda text: text,
                       mask=None):
    """Performs semantic search, filters the results, and displays them.  ``embeddings_db``: a `TheoremIndex` (kept across
    calls) or the ``[N x d]`` matrix / tensor ``load_embedding_library`` returns (indexed for this call); ``mask``: the
    ``filters.filter_mask`` of this sidebar state when the caller keeps it across queries."""
    if not query:
        st.info("Please enter a search query to begin.")
        return
    if not filters["sources"]:
        st.warning("Please select at least one source from the sidebar to see results.")
        return
    query_emb = model.encode(query, convert_to_tensor=True)
    own = not isinstance(embeddings_db, TheoremIndex)
    index = TheoremIndex.from_embeddings(embeddings_db, metric="cos") if own else embeddings_db
    try:
        filtered_results = _filters.search_filtered(index, query_emb, theorems_data, filters, mask)
    finally:
        if own:
            index.close()
    st.subheader(f"Found {len(filtered_results)} Matching Results")
    if not filtered_results:
        st.warning("No results found matching your query and filter criteria.")
        return
    for i, result in enumerate(filtered_results):
        info = result["info"]
        expander_title = (
            f"**Result {i+1} | Similarity: {result['similarity']:.4f} | "
            f"Type: {info['type'].capitalize()}**"
        )
        with st.expander(expander_title):
            st.markdown(f"**Paper:** *{info['paper_title']}*")
            st.markdown(f"**Authors:** {', '.join(info['authors']) if info['authors'] else 'N/A'}")
            st.markdown(f"**Source:** {info['source']} ([Link]({info['paper_url']}))")
            st.markdown(f"**Math Tag:** `{info['primary_math_tag']}` | **Citations:** {info['citations']} | **Year:** {info.get('year', 'N/A')}")
            st.markdown("*Note: Linking to a specific page within an arXiv PDF is not directly possible.*",
                        help="arXiv links go to the abstract page, not a specific page in the PDF.")
            st.markdown("---")
            if info["global_context"]:
                cleaned_ctx = clean_latex_for_display(info["global_context"])
                st.markdown("> " + cleaned_ctx.replace("\n", "\n> "))
                st.write("")
            cleaned_content = clean_latex_for_display(info["content"])
            st.markdown(cleaned_content)
