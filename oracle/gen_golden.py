#!/usr/bin/env python3
"""Generate the committed golden fixtures under ``tests/golden/``.

Run in the authoring container only (it reads ``/root/reference`` and needs
torch-CPU); the GPU box never runs it.  Fixtures are data: seeds/inputs and
expected outputs.  Nothing of the reference's source text is written out.

1. ``metrics_*.json`` - the eleven pure-numpy definitions of
   ``/root/reference/compare_embeddings.py:47-371`` (``rank_concepts`` ...
   ``q_measure_at_k``) are taken out of the parsed module with ``ast`` (the
   module itself cannot be imported: ``sentence_transformers`` is absent and its
   notebook cells open a database connection) and executed on seeded inputs.
2. ``search_*.npz`` - the published algorithm of ``sentence_transformers.util.
   cos_sim`` (``F.normalize`` both operands, ``torch.mm``) and the reference's
   selection primitives (``torch.topk(sorted=True)``, ``np.argsort(-s)[:k]``)
   executed with torch-CPU / numpy on seeded inputs, next to fp64 truth.
3. ``adversarial.json`` - what those primitives return on ties, a zero row,
   NaN scores and k > N (score multisets and index sets).
4. ``text_to_embed.json`` - one synthetic paper through the string assembly of
   ``app_create_embeddings.py:48-70`` (the expression is evaluated from the
   parsed reference module, not retyped).
6. ``showcase.json`` - what ``search_and_display`` of the showcase app (``app_showcase_model.py:79-156``) displays for
   seven sidebar states (see ``gen_showcase``).
5. ``callsites.json`` - what the reference's own call-site bodies print / display:
   ``compare_embeddings`` and ``evaluate_retrieval`` (``compare_embeddings.py:14-35,55-92``)
   and ``search_theorems`` (``app_scratchpad.py:120-154``), taken from the parsed files and
   executed around ``util.cos_sim`` = the published torch formula with a seeded stand-in
   model and a recording ``st`` (exact score ties included).
"""
import ast
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

METRIC_FUNCS = [
    "rank_concepts", "precision_at_k", "hit_at_k", "mrr_at_k", "_generate_qrels",
    "_get_rels_for_query", "_dcg_from_rels", "ndcg_at_k", "_get_rels_sparse",
    "err_at_k", "q_measure_at_k",
]


def load_reference_metrics():
    src = open(os.path.join(REF, "compare_embeddings.py"), encoding="utf-8").read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in METRIC_FUNCS]
    assert len(keep) == len(METRIC_FUNCS), [n.name for n in keep]
    ns = {"np": np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "compare_embeddings.py", "exec"), ns)
    return ns


def metric_cases():
    """(name, Q, N, seed, qrels-kind, k)"""
    return [
        ("q4_n10_identity", 4, 10, 0, "identity", 3),
        ("q8_n50_identity", 8, 50, 1, "identity", 5),
        ("q16_n200_graded", 16, 200, 2, "graded", 5),
        ("q16_n200_graded_k10", 16, 200, 3, "graded", 10),
        ("q12_n64_paper", 12, 64, 4, "paper", 5),
    ]


def build_qrels(kind, Q, N, rng, ns):
    if kind == "identity":
        return {q: {q: 1} for q in range(Q)}
    if kind == "graded":
        # exactly one grade-1 document per query, plus 0.5 / 2 / 3 grades
        out = {}
        for q in range(Q):
            docs = rng.choice(N, size=6, replace=False)
            grades = [1, 0.5, 0.5, 2, 3, 0]
            out[q] = {int(d): g for d, g in zip(docs, grades)}
        return out
    if kind == "paper":
        # the reference's own generator (grades 0.5 / 0), then one exact match added
        papers = [f"p{int(i)}" for i in rng.integers(0, 6, size=N)]
        slogans = [(f"s{j}", papers[j]) for j in range(N)]
        queries = [(f"q{i}", papers[int(rng.integers(0, N))]) for i in range(Q)]
        qrels = ns["_generate_qrels"](queries, slogans)
        for q in range(Q):
            same = [j for j, g in qrels[q].items() if g == 0.5]
            qrels[q][same[0] if same else 0] = 1
        return {"_queries": queries, "_slogans": slogans, "qrels": qrels}
    raise ValueError(kind)


def gen_metrics(ns):
    for name, Q, N, seed, kind, k in metric_cases():
        rng = np.random.default_rng(seed)
        sim = rng.standard_normal((Q, N)).astype(np.float32)
        qr = build_qrels(kind, Q, N, rng, ns)
        extra = {}
        if kind == "paper":
            extra = {"queries": qr["_queries"], "slogans": qr["_slogans"]}
            # the un-patched generator output, to pin _generate_qrels itself
            extra["generated_qrels"] = {str(q): {str(d): g for d, g in v.items()}
                                        for q, v in ns["_generate_qrels"](qr["_queries"], qr["_slogans"]).items()}
            qr = qr["qrels"]
        expected = {
            "precision_at_k": ns["precision_at_k"](sim, qr, k=k),
            "precision_at_1": ns["precision_at_k"](sim, qr, k=1),
            "hit_at_k": ns["hit_at_k"](sim, qr, k=k),
            "mrr_at_k": ns["mrr_at_k"](sim, qr, k=k),
            "mrr_all": ns["mrr_at_k"](sim, qr, k=None),
            "ndcg_at_k": ns["ndcg_at_k"](sim, qr, k=k),
            "ndcg_linear": ns["ndcg_at_k"](sim, qr, k=k, gain="linear"),
            "err_at_k": ns["err_at_k"](sim, qr, k=k),
            "err_maxrel4": ns["err_at_k"](sim, qr, k=k, max_rel=4.0),
            "q_measure_at_k": ns["q_measure_at_k"](sim, qr, k=k),
            "rank_row0": [int(i) for i in ns["rank_concepts"](sim)[0]],
        }
        doc = {
            "seed": seed, "Q": Q, "N": N, "k": k, "kind": kind,
            "recipe": "np.random.default_rng(seed).standard_normal((Q,N)).astype(float32)",
            "sim_matrix": sim.tolist(),
            "qrels": {str(q): {str(d): g for d, g in v.items()} for q, v in qr.items()},
            "expected": expected,
        }
        doc.update(extra)
        with open(os.path.join(OUT, f"metrics_{name}.json"), "w") as f:
            json.dump(doc, f)
        print("metrics", name, {k_: v for k_, v in expected.items() if k_ != "rank_row0"})


def st_cos_sim(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Published algorithm of sentence_transformers.util.cos_sim."""
    if a.dim() == 1:
        a = a.unsqueeze(0)
    if b.dim() == 1:
        b = b.unsqueeze(0)
    a_n = torch.nn.functional.normalize(a, p=2, dim=1)
    b_n = torch.nn.functional.normalize(b, p=2, dim=1)
    return torch.mm(a_n, b_n.transpose(0, 1))


def bf16_round(x: np.ndarray) -> np.ndarray:
    return torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()


def search_cases():
    """(name, N, B, d, k, metric, dtype, seed, scale) - corpus rows are NOT pre-normalised when metric=cos."""
    return [
        ("n1000_b1_k5_cos_f32", 1000, 1, 768, 5, "cos", "f32", 11),
        ("n1000_b7_k10_cos_f32", 1000, 7, 768, 10, "cos", "f32", 12),
        ("n4096_b7_k200_cos_f32", 4096, 7, 768, 200, "cos", "f32", 13),
        ("n4096_b256_k10_ip_bf16", 4096, 256, 768, 10, "ip", "bf16", 14),
        ("n65537_b1_k10_cos_f32", 65537, 1, 768, 10, "cos", "f32", 15),
        ("n65537_b7_k1_ip_f32", 65537, 7, 768, 1, "ip", "f32", 16),
        ("n65537_b256_k10_ip_bf16", 65537, 256, 768, 10, "ip", "bf16", 17),
        ("n4096_b3_k5_cos_f32_d1024", 4096, 3, 1024, 5, "cos", "f32", 18),
        ("n20000_b33_k200_cos_bf16", 20000, 33, 768, 200, "cos", "bf16", 19),
    ]


def make_inputs(N, B, d, seed, metric):
    """Seeded inputs; the recipe lives in oracle.golden_inputs (numpy only, elementwise
    operations only, so it regenerates bit-identically on any host)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle.oracle import golden_inputs
    return golden_inputs(N, B, d, seed, metric)


def gen_search():
    for name, N, B, d, k, metric, dtype, seed in search_cases():
        q, c = make_inputs(N, B, d, seed, metric)
        if metric == "cos":
            qn = torch.nn.functional.normalize(torch.from_numpy(q), p=2, dim=1).numpy()
            cn = torch.nn.functional.normalize(torch.from_numpy(c), p=2, dim=1).numpy()
        else:
            qn, cn = q, c
        if dtype == "bf16":
            qn, cn = bf16_round(qn), bf16_round(cn)
        # the reference formulation: normalise (again, for metric=cos on f32 this IS cos_sim) + mm
        if metric == "cos" and dtype == "f32":
            S = st_cos_sim(torch.from_numpy(q), torch.from_numpy(c))
        else:
            S = torch.mm(torch.from_numpy(qn), torch.from_numpy(cn).T)
        kk = min(k, N)
        tk = torch.topk(S, k=kk, dim=1, sorted=True)
        argsort_idx = np.argsort(-S.numpy(), axis=1)[:, :kk]
        truth = qn.astype(np.float64) @ cn.astype(np.float64).T
        order = np.lexsort((np.broadcast_to(np.arange(N), truth.shape), -truth), axis=1)
        ext = order[:, : min(N, kk + 8)]
        np.savez_compressed(
            os.path.join(OUT, f"search_{name}.npz"),
            N=N, B=B, d=d, k=k, metric=metric, dtype=dtype, seed=seed,
            torch_topk_idx=tk.indices.numpy().astype(np.int64),
            torch_topk_scores=tk.values.numpy().astype(np.float32),
            argsort_idx=argsort_idx.astype(np.int64),
            truth_idx=ext.astype(np.int64),
            truth_scores=np.take_along_axis(truth, ext, axis=1),
        )
        print("search", name, "top1", tk.indices[0, 0].item(), float(tk.values[0, 0]))


def gen_adversarial():
    out = {}
    s = np.array([.5, .9, .9, .1, .9, .5], dtype=np.float32)
    out["ties6"] = {
        "scores": s.tolist(), "k": 4,
        "argsort_neg": np.argsort(-s)[:4].tolist(),
        "argsort_rev": s.argsort()[::-1][:4].tolist(),
        "torch_topk": torch.topk(torch.from_numpy(s), 4, sorted=True).indices.tolist(),
    }
    z = np.zeros(1000, dtype=np.float32)
    z[[7, 500, 900]] = 1.0
    out["sparse_ones"] = {
        "n": 1000, "ones_at": [7, 500, 900], "k": 3,
        "argsort_neg": np.argsort(-z)[:3].tolist(),
        "torch_topk": torch.topk(torch.from_numpy(z), 3, sorted=True).indices.tolist(),
    }
    n = np.array([0.3, np.nan, 0.7, 0.1], dtype=np.float32)
    out["nan"] = {
        "scores": [0.3, None, 0.7, 0.1], "k": 2,
        "argsort_neg": np.argsort(-n)[:2].tolist(),
        "argsort_neg_full": np.argsort(-n).tolist(),
        "torch_topk": torch.topk(torch.from_numpy(n), 2, sorted=True).indices.tolist(),
    }
    # zero row through cos_sim: score 0, not NaN
    a = torch.tensor([[1.0, 2.0, 2.0]])
    b = torch.tensor([[0.0, 0.0, 0.0], [2.0, 4.0, 4.0], [-1.0, 0.0, 0.0]])
    out["zero_row"] = {"a": a.tolist(), "b": b.tolist(), "cos_sim": st_cos_sim(a, b).tolist()}
    # duplicates: rows 3 and 11 equal the query direction
    rng = np.random.default_rng(77)
    c = rng.standard_normal((16, 8)).astype(np.float32)
    qv = rng.standard_normal(8).astype(np.float32)
    c[3] = 2.0 * qv
    c[11] = 0.5 * qv
    S = st_cos_sim(torch.from_numpy(qv), torch.from_numpy(c))[0]
    out["duplicates"] = {
        "corpus": c.tolist(), "query": qv.tolist(), "k": 3,
        "cos_sim": S.tolist(),
        "torch_topk": torch.topk(S, 3, sorted=True).indices.tolist(),
        "argsort_neg": np.argsort(-S.numpy())[:3].tolist(),
    }
    with open(os.path.join(OUT, "adversarial.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("adversarial", {k: v.get("torch_topk") for k, v in out.items() if "torch_topk" in v})


def gen_text():
    """Evaluate the reference's own f-string / join expressions on a synthetic paper."""
    src = open(os.path.join(REF, "app_create_embeddings.py"), encoding="utf-8").read()
    tree = ast.parse(src)
    gc_expr = tte_expr = None
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", None) == "global_context":
            gc_expr = node.value
        if isinstance(node, ast.Dict):
            for kx, vx in zip(node.keys, node.values):
                if isinstance(kx, ast.Constant) and kx.value == "text_to_embed":
                    tte_expr = vx
    assert gc_expr is not None and tte_expr is not None
    papers = [
        {"title": "On trees", "url": "http://example.org/1", "authors": ["A. Author"], "citations": 3,
         "primary_math_tag": "math.CO", "year": 2020, "source": "arXiv", "journal_published": True,
         "global_notations": "G denotes a finite graph.", "global_definitions": "A tree is a connected acyclic graph.",
         "global_assumptions": "",
         "theorems": [{"type": "theorem", "content": "A tree on $n$ vertices has $n-1$ edges."},
                      {"type": "lemma", "content": "Every tree with $n\\ge 2$ has a leaf."}]},
        {"title": "No context", "source": "Stacks Project",
         "theorems": [{"type": "proposition", "content": "Let $X$ be a scheme."}]},
    ]
    cases = []
    for data in papers:
        gc = eval(compile(ast.Expression(gc_expr), "gc", "eval"), {"data": data, "filter": filter})
        for theorem in data["theorems"]:
            tte = eval(compile(ast.Expression(tte_expr), "tte", "eval"),
                       {"data": data, "theorem": theorem, "global_context": gc})
            cases.append({"paper": data, "theorem": theorem, "global_context": gc, "text_to_embed": tte})
    with open(os.path.join(OUT, "text_to_embed.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print("text_to_embed", len(cases), repr(cases[0]["text_to_embed"][:60]))


# ----------------------------------------------------------------------------------------------------------------
# 5. call-site bodies: the reference's own functions AROUND util.cos_sim, executed here
# ----------------------------------------------------------------------------------------------------------------
def stub_embedding(text: str, seed: int, d: int) -> np.ndarray:
    """The stand-in model of the call-site fixtures: one seeded vector per distinct text (equal texts -> equal rows
    -> exact score ties).  tests/ rebuild the same model from this rule."""
    import zlib
    return np.random.default_rng([seed, zlib.crc32(text.encode("utf-8"))]).standard_normal(d).astype(np.float32)


class StubModel:
    def __init__(self, seed, d):
        self.seed, self.d = seed, d

    def encode(self, texts, convert_to_tensor=False, **_):
        single = isinstance(texts, str)
        rows = np.stack([stub_embedding(t, self.seed, self.d) for t in ([texts] if single else texts)])
        out = torch.from_numpy(rows[0] if single else rows)
        return out if convert_to_tensor else out.numpy()


class RecordingStreamlit:
    """Records what search_theorems hands to streamlit (st.subheader / st.expander / st.markdown / st.info / st.write)."""

    def __init__(self):
        self.calls = []

    def _rec(self, name):
        def f(*a, **k):
            self.calls.append([name] + [str(x) for x in a])
        return f

    def __getattr__(self, name):
        if name == "expander":
            outer = self

            class Ctx:
                def __init__(self, title):
                    outer.calls.append(["expander", str(title)])

                def __enter__(self):
                    return self

                def __exit__(self, *exc):
                    return False
            return Ctx
        return self._rec(name)


def extract_functions(path, names):
    tree = ast.parse(open(path, encoding="utf-8").read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in keep) == sorted(names), [n.name for n in keep]
    return ast.Module(body=keep, type_ignores=[])


def gen_callsites():
    """compare_embeddings (compare_embeddings.py:14-35: argmax + argsort()[::-1]), evaluate_retrieval (:55-92) and
    search_theorems (app_scratchpad.py:120-154: np.argsort(-scores)[:5]) are pure Python around util.cos_sim.  Their
    function bodies are taken from the parsed reference files and run with util.cos_sim = the published torch formula,
    a model that returns seeded vectors, and a recording streamlit; what they print / display is the fixture."""
    import contextlib
    import io
    import re
    import types
    util = types.SimpleNamespace(cos_sim=st_cos_sim)
    ns = load_reference_metrics()
    ns.update({"util": util, "np": np})
    exec(compile(extract_functions(os.path.join(REF, "compare_embeddings.py"), ["compare_embeddings", "evaluate_retrieval"]),
                 "compare_embeddings.py", "exec"), ns)
    st = RecordingStreamlit()
    ns2 = {"util": util, "np": np, "st": st, "re": re}
    exec(compile(extract_functions(os.path.join(REF, "app_scratchpad.py"), ["search_theorems", "clean_latex_for_display"]),
                 "app_scratchpad.py", "exec"), ns2)
    out = {"recipe": "model.encode(text) = np.random.default_rng([seed, zlib.crc32(text.encode('utf-8'))]).standard_normal(d)"
                     ".astype(float32), one row per text; util.cos_sim = F.normalize(p=2, dim=1) both sides + mm",
           "cases": {}}
    # compare_embeddings: 3 latex tokens vs 9 concepts, two concepts are the same text (an exact tie)
    seed, d = 41, 64
    latex = ["\\alpha", "\\mathbb{R}^n", "G / H"]
    concepts = ["alpha", "real coordinate space", "quotient group", "tree", "quotient group", "scheme", "alpha", "leaf", "edge"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ns["compare_embeddings"](StubModel(seed, d), latex, concepts, top_k=4)
    out["cases"]["compare_embeddings"] = {"seed": seed, "d": d, "latex_texts": latex, "concept_texts": concepts, "top_k": 4,
                                          "stdout": buf.getvalue()}
    # evaluate_retrieval: 40 theorems, 12 queries, graded qrels with one exact match per query
    seed, d = 42, 96
    theorems = [(f"theorem number {j} about object {j % 7}", f"paper{j % 5}") for j in range(40)]
    queries = [(f"query {i} about object {i % 7}", f"paper{i % 5}") for i in range(12)]
    rng = np.random.default_rng(7)
    qrels = {i: {int(j): g for j, g in zip(rng.choice(40, size=4, replace=False), [1, 0.5, 2, 3])} for i in range(12)}
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ns["evaluate_retrieval"](StubModel(seed, d), theorems, queries, qrels, top_k_report=3)
    out["cases"]["evaluate_retrieval"] = {"seed": seed, "d": d, "theorems": theorems, "queries": queries,
                                          "qrels": {str(q): {str(j): g for j, g in v.items()} for q, v in qrels.items()},
                                          "top_k_report": 3, "stdout": buf.getvalue()}
    # search_theorems: 30 theorems (two share a text), one query
    seed, d = 43, 128
    data = [{"type": ["theorem", "lemma", "proposition"][j % 3], "paper_title": f"Paper {j // 3}", "paper_url": f"http://example.org/{j // 3}",
             "global_context": "" if j % 4 else "**Notations:**\nG is a graph.", "content": f"Statement {j}: $x_{{{j}}} = {j}$.",
             "text_to_embed": f"Statement {j % 29}"} for j in range(30)]
    model = StubModel(seed, d)
    db = model.encode([t["text_to_embed"] for t in data], convert_to_tensor=True)
    ns2["search_theorems"]("Statement 3", model, data, db)
    titles = [c[1] for c in st.calls if c[0] == "expander"]
    out["cases"]["search_theorems"] = {"seed": seed, "d": d, "theorems_data": data, "query": "Statement 3",
                                       "expander_titles": titles, "calls": st.calls}
    st2 = RecordingStreamlit()
    ns2["st"] = st2
    ns2["search_theorems"]("", model, data, db)
    out["cases"]["search_theorems"]["empty_query_calls"] = st2.calls
    with open(os.path.join(OUT, "callsites.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("callsites:", out["cases"]["compare_embeddings"]["stdout"].splitlines()[0], "|", titles[0])


def gen_showcase():
    """search_and_display of the showcase app (app_showcase_model.py:79-156: encode -> util.cos_sim -> torch.topk(min(200, N))
    -> the sidebar's predicates -> display) run as the reference wrote it, with util.cos_sim = the published torch formula, the
    seeded stand-in model and the recording streamlit.  Two things are set aside, both outside the hot path: the file's one
    Python-3.12-only f-string (a backslash inside the braces, :149 - this interpreter is 3.10) is back-ported textually to
    the equivalent concatenation before parsing, and ``clean_latex_for_display`` (UI) is the identity, so that every
    recorded call is comparable.  Fixture: theorems, filter states, and what each state displayed."""
    import re
    import types
    src = open(os.path.join(REF, "app_showcase_model.py"), encoding="utf-8").read()
    old = "st.markdown(f\"> {cleaned_ctx.replace('\\n', '\\n> ')}\")"
    assert src.count(old) == 1, "the 3.12-only f-string of app_showcase_model.py:149 was not found"
    src = src.replace(old, "st.markdown('> ' + cleaned_ctx.replace('\\n', '\\n> '))")
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "search_and_display"]
    assert len(keep) == 1
    util = types.SimpleNamespace(cos_sim=st_cos_sim)
    seed, d, n = 47, 96, 60
    rng = np.random.default_rng(5)
    types_, tags = ["theorem", "lemma", "proposition", "corollary"], ["math.AG", "math.NT", "math.PR", "math.CO"]
    authors, sources = ["A. Author", "B. Writer", "C. Prover", "E. Noether"], ["arXiv", "Stacks Project", "ProofWiki"]
    data = []
    for j in range(n):
        srcj = sources[int(rng.integers(0, 3))]
        item = {"type": types_[int(rng.integers(0, 4))].capitalize() if j % 3 == 0 else types_[int(rng.integers(0, 4))],
                "primary_math_tag": tags[int(rng.integers(0, 4))],
                "authors": [authors[i] for i in rng.choice(4, size=int(rng.integers(0, 3)), replace=False)],
                "source": srcj, "citations": int(rng.integers(0, 300)),
                "paper_title": f"Paper {j // 3}", "paper_url": f"http://example.org/{j // 3}",
                "global_context": "" if j % 4 else "**Notations:**\nG is a graph.", "content": f"Statement {j}: $x_{{{j}}} = {j}$.",
                "text_to_embed": f"Statement {j % 57}"}
        if srcj == "arXiv":
            if j % 7:
                item["year"] = int(rng.integers(1995, 2026))
            if j % 5:
                item["journal_published"] = bool(rng.integers(0, 2))
        data.append(item)
    base = {"types": [], "tags": [], "authors": [], "sources": list(sources), "citation_range": (0, 10 ** 6), "year_range": None,
            "journal_status": "All", "top_k": 5}
    states = {"open": dict(base), "types_tags": dict(base, types=["lemma", "theorem"], tags=["math.AG", "math.NT"], top_k=8),
              "authors": dict(base, authors=["E. Noether", "C. Prover"], top_k=3),
              "arxiv_years_journal": dict(base, sources=["arXiv"], year_range=(2005, 2022), journal_status="Journal Article", top_k=20),
              "preprints_cited": dict(base, journal_status="Preprint Only", citation_range=(50, 200), top_k=4),
              "nothing_passes": dict(base, types=["corollary"], tags=["math.CO"], authors=["A. Author"], citation_range=(299, 299)),
              "no_sources": dict(base, sources=[])}
    model = StubModel(seed, d)
    db = model.encode([t["text_to_embed"] for t in data], convert_to_tensor=True)
    out = {"recipe": "as callsites.json; clean_latex_for_display = identity", "seed": seed, "d": d, "theorems_data": data,
           "query": "Statement 11", "states": {}}
    for name, f in states.items():
        st = RecordingStreamlit()
        ns = {"util": util, "torch": torch, "st": st, "re": re, "clean_latex_for_display": lambda text: text}
        exec(compile(ast.Module(body=keep, type_ignores=[]), "app_showcase_model.py", "exec"), ns)
        ns["search_and_display"](out["query"], model, data, db, f)
        out["states"][name] = {"filters": f, "calls": st.calls}
    st = RecordingStreamlit()
    ns = {"util": util, "torch": torch, "st": st, "re": re, "clean_latex_for_display": lambda text: text}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "app_showcase_model.py", "exec"), ns)
    ns["search_and_display"]("", model, data, db, states["open"])
    out["empty_query_calls"] = st.calls
    with open(os.path.join(OUT, "showcase.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("showcase:", {k: len([c for c in v["calls"] if c[0] == "expander"]) for k, v in out["states"].items()})


def gen_paper_filter():
    """parse_paper_filter / extract_arxiv_id / normalize_title of the production app (streamlit_app.py:44-47,118-143), run as the
    reference wrote them (function bodies and the ARXIV_ID_RE assignment taken from the parsed file) on a list of inputs."""
    import re
    tree = ast.parse(open(os.path.join(REF, "streamlit_app.py"), encoding="utf-8").read())
    keep = [n for n in tree.body
            if (isinstance(n, ast.FunctionDef) and n.name in ("extract_arxiv_id", "normalize_title", "parse_paper_filter"))
            or (isinstance(n, ast.Assign) and any(getattr(t, "id", "") == "ARXIV_ID_RE" for t in n.targets))]
    assert len(keep) == 4
    ns = {"re": re}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "streamlit_app.py", "exec"), ns)
    inputs = ["", None, "2401.12345", "https://arxiv.org/abs/2401.12345v2, Optimal Transport", "arXiv.org/pdf/math-ph/0701012",
              " Riemann  Hypothesis ,, STRASSE ", "math/0307245, 0704.0001 , hep-th/9901001x", "1234.567", "Título Ünïcode, ARXIV.ORG/ABS/2101.00001",
              "a,b,a , B", "see https://arxiv.org/abs/1706.03762 and 1810.04805"]
    out = {"inputs": inputs, "parsed": [], "ids": [ns["extract_arxiv_id"](x) for x in inputs], "titles": [ns["normalize_title"](x) for x in inputs]}
    for x in inputs:
        r = ns["parse_paper_filter"](x)
        out["parsed"].append({"ids": sorted(r["ids"]), "titles": sorted(r["titles"])})
    with open(os.path.join(OUT, "paper_filter.json"), "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)
    print("paper_filter:", out["parsed"][3])


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    ns = load_reference_metrics()
    gen_metrics(ns)
    gen_search()
    gen_adversarial()
    gen_text()
    gen_callsites()
    gen_showcase()
    gen_paper_filter()


if __name__ == "__main__":
    sys.exit(main())
