"""Shared by the CPU and GPU call-site tests: the stand-in model and the recording streamlit of oracle/gen_golden.py's
``gen_callsites`` (rebuilt from the rule stored in tests/golden/callsites.json)."""
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
