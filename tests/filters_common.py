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


def make_sql_rows(n, seed=9):
    """Joined (paper, theorem) rows as the production schema holds them, NULLs included."""
    rng = np.random.default_rng(seed)
    cats = ["math.AG", "math.NT", "math.PR", None]
    names = ["Theorem", "Lemma", "Main Theorem", "Corollary", "proposition", None]
    rows = []
    for i in range(n):
        kind = int(rng.integers(0, 10))
        link = None if kind == 0 else (f"https://arxiv.org/abs/{2000 + i % 25}.{i:05d}" if kind < 7
                                       else f"https://stacks.math.columbia.edu/tag/{i:04X}")
        rows.append({
            "link": link,
            "authors": None if i % 11 == 0 else [AUTHORS[j] for j in rng.choice(5, size=int(rng.integers(1, 3)), replace=False)],
            "primary_category": cats[int(rng.integers(0, 4))],
            "year": None if i % 13 == 0 else int(rng.integers(1995, 2026)),
            "journal_ref": None if rng.random() < 0.6 else "J. Test Math. 1 (2020)",
            "title": None if i % 17 == 0 else f"On the {['cohomology', 'Zeta function', 'random walk'][i % 3]} of things {i}",
            "type_name": names[int(rng.integers(0, 6))],
            "citations": None if i % 5 == 0 else int(rng.integers(0, 200)),
        })
    return rows


def sql_filter_states(top_k=10):
    base = {"sources": ["arXiv", "Stacks Project"], "authors": [], "tags": [], "year_range": None, "journal_status": "All",
            "paper_filter": {"ids": set(), "titles": set()}, "types": [], "citation_range": (0, 10**9),
            "include_unknown_citations": True, "citation_weight": 0.0, "top_k": top_k}
    return {
        "open": dict(base),
        "arxiv_only": dict(base, sources=["arXiv"]),
        "stacks_only": dict(base, sources=["Stacks Project"]),
        "authors_tags": dict(base, authors=["E. Noether"], tags=["math.AG", "math.PR"]),
        "years": dict(base, year_range=(2010, 2015)),
        "journal": dict(base, journal_status="Journal Article"),
        "preprint_known_citations": dict(base, journal_status="Preprint Only", include_unknown_citations=False,
                                         citation_range=(10, 150)),
        "paper_filter": dict(base, paper_filter={"ids": {"2003."}, "titles": {"zeta FUNCTION"}}),
        "types": dict(base, types=["theorem", "lemma"]),
        "everything": dict(base, sources=["arXiv"], authors=["A. Author", "C. Prover"], tags=["math.NT"], year_range=(2000, 2024),
                           types=["theorem"], citation_range=(5, 190), include_unknown_citations=False),
    }
