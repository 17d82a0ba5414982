import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The first `import torch` on a fresh box can take minutes while the image pages in: do it here, outside any test's
    # timeout (pytest.ini: 300 s per test), instead of inside whichever test happens to need it first.
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def golden_path(name):
    return os.path.join(GOLDEN, name)


def load_json(name):
    with open(golden_path(name)) as f:
        return json.load(f)


def load_qrels(doc, key="qrels"):
    """JSON object keys are strings; the metric functions index qrels by int."""
    return {int(q): {int(d): g for d, g in v.items()} for q, v in doc[key].items()}


def search_cases():
    return sorted(f[len("search_"):-len(".npz")] for f in os.listdir(GOLDEN)
                  if f.startswith("search_") and f.endswith(".npz"))


def metric_cases():
    return sorted(f[len("metrics_"):-len(".json")] for f in os.listdir(GOLDEN)
                  if f.startswith("metrics_") and f.endswith(".json"))


def load_search_case(name):
    z = np.load(golden_path(f"search_{name}.npz"))
    case = {k: z[k] for k in z.files}
    for k in ("N", "B", "d", "k", "seed"):
        case[k] = int(case[k])
    for k in ("metric", "dtype"):
        case[k] = str(case[k])
    return case


def gpu_available():
    try:
        from theoremsearch_amd import _ffi
        return _ffi.device_count() > 0
    except Exception:
        return False
