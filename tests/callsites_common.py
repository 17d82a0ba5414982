"""Shared by the CPU and GPU call-site tests: the stand-in model and the recording streamlit of oracle/gen_golden.py's
``gen_callsites`` (rebuilt from the rule stored in tests/golden/callsites.json)."""
import re
import zlib

import numpy as np


def stub_embedding(text, seed, d):
    return np.random.default_rng([seed, zlib.crc32(text.encode("utf-8"))]).standard_normal(d).astype(np.float32)


class StubModel:
    def __init__(self, seed, d):
        self.seed, self.d = seed, d

    def encode(self, texts, convert_to_tensor=False, convert_to_numpy=True, **_):
        single = isinstance(texts, str)
        rows = np.stack([stub_embedding(t, self.seed, self.d) for t in ([texts] if single else texts)])
        return rows[0] if single else rows


class RecordingStreamlit:
    def __init__(self):
        self.calls = []

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

        def f(*a, **k):
            self.calls.append([name] + [str(x) for x in a])
        return f


def run_compare(ce, case, capsys):
    ce.compare_embeddings(StubModel(case["seed"], case["d"]), case["latex_texts"], case["concept_texts"], top_k=case["top_k"])
    return capsys.readouterr().out


def run_evaluate(ce, case, capsys):
    qrels = {int(q): {int(j): g for j, g in v.items()} for q, v in case["qrels"].items()}
    ce.evaluate_retrieval(StubModel(case["seed"], case["d"]), [tuple(t) for t in case["theorems"]],
                          [tuple(t) for t in case["queries"]], qrels, top_k_report=case["top_k_report"])
    return capsys.readouterr().out


def results_from_calls(calls, data):
    """The hits behind the recorded streamlit calls of the reference's display loop, as ``[(row, "0.1234", "Type")]``: one
    expander per hit, titled ``**Result i | Similarity: s | Type: T**``, holding the paper line and (last) the statement -
    together they name the row of ``data``.  The fixtures record what the reference's own function displayed; the mirrors
    return the list that function displays, so this is the comparison at the level of results (rows + ``:.4f`` similarities)."""
    out, i = [], 0
    while i < len(calls):
        if calls[i][0] != "expander":
            i += 1
            continue
        m = re.fullmatch(r"\*\*Result (\d+) \| Similarity: (-?[\d.]+) \| Type: (\w+)\*\*", calls[i][1])
        assert m and int(m.group(1)) == len(out) + 1, calls[i][1]
        j = i + 1
        while j < len(calls) and calls[j][0] != "expander":
            j += 1
        md = [c[1] for c in calls[i + 1:j] if c[0] == "markdown"]
        rows = [r for r, t in enumerate(data) if f"**Paper:** *{t['paper_title']}*" == md[0] and t["content"] == md[-1]]
        assert len(rows) == 1, (calls[i][1], rows)
        out.append((rows[0], m.group(2), m.group(3)))
        i = j
    return out


def results_of(hits, data):
    """The same triples from a mirror's return value ``[{"info", "similarity"}]``."""
    return [(next(r for r, t in enumerate(data) if t is h["info"] or t == h["info"]), f"{h['similarity']:.4f}", h["info"]["type"].capitalize())
            for h in (hits or [])]
