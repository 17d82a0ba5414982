"""Synthetic theorem metadata + sidebar filter states shared by the CPU and GPU filter tests."""
import numpy as np

TYPES = ["theorem", "lemma", "proposition", "corollary"]
TAGS = ["math.AG", "math.NT", "math.PR", "math.CO", "math.AP"]
AUTHORS = ["A. Author", "B. Writer", "C. Prover", "D. Lemma", "E. Noether"]
SOURCES = ["arXiv", "Stacks Project", "ProofWiki"]


def make_theorems(n, seed=5):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        src = SOURCES[int(rng.integers(0, 3))]
        item = {
            "type": TYPES[int(rng.integers(0, 4))].capitalize() if i % 3 == 0 else TYPES[int(rng.integers(0, 4))],
            "primary_math_tag": TAGS[int(rng.integers(0, 5))],
            "authors": [AUTHORS[j] for j in rng.choice(5, size=int(rng.integers(1, 4)), replace=False)],
            "source": src,
            "citations": int(rng.integers(0, 300)),
        }
        if src == "arXiv":
            if i % 7:
                item["year"] = int(rng.integers(1995, 2026))
            if i % 5:
                item["journal_published"] = bool(rng.integers(0, 2))
        out.append(item)
    return out


def filter_states(top_k=10):
    base = {"types": [], "tags": [], "authors": [], "sources": list(SOURCES), "citation_range": (0, 10**9),
            "year_range": None, "journal_status": "All", "top_k": top_k}
    return {
        "open": dict(base),
        "types_tags": dict(base, types=["lemma", "theorem"], tags=["math.AG", "math.NT"]),
        "authors": dict(base, authors=["E. Noether", "C. Prover"]),
        "arxiv_years_journal": dict(base, sources=["arXiv"], year_range=(2010, 2020), journal_status="Journal Article"),
        "preprints_cited": dict(base, journal_status="Preprint Only", citation_range=(50, 120)),
        "selective": dict(base, types=["corollary"], tags=["math.AP"], authors=["D. Lemma"], citation_range=(200, 299)),
        "nothing": dict(base, sources=["ProofWiki"], tags=["math.XX"]),
    }
