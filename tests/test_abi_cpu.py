"""C-ABI surface checks that need no GPU: the library loads, exports every symbol the header
declares, the ctypes table covers them, and compute entry points fail loudly without a device."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, gpu_available
from theoremsearch_amd import _ffi


def header_functions():
    text = open(os.path.join(ROOT, "include", "tsearch.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ts_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _ffi.load()
    names = header_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in tsearch.h but not exported"
        assert name in _ffi._SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_ffi._SIGNATURES) == names


def test_version_and_error_string():
    lib = _ffi.load()
    assert lib.ts_version() == 100
    assert isinstance(lib.ts_last_error(), bytes)


def test_constants_match_header():
    text = open(os.path.join(ROOT, "include", "tsearch.h")).read()
    defs = dict(re.findall(r"#define\s+(TS_[A-Z0-9_]+)\s+\(?(-?\d+)\)?", text))
    for name in ("TS_F32", "TS_BF16", "TS_METRIC_IP", "TS_METRIC_COS", "TS_ALGO_AUTO", "TS_ALGO_SCAN",
                 "TS_ALGO_MFMA", "TS_MAX_K"):
        assert int(defs[name]) == getattr(_ffi, name)


@pytest.mark.skipif(gpu_available(), reason="checks the no-device behaviour")
def test_compute_fails_loudly_without_device():
    from theoremsearch_amd import TheoremIndex, merge_topk
    assert _ffi.device_count() == 0
    with pytest.raises(_ffi.TSearchError) as e:
        TheoremIndex(16, 8)
    assert e.value.code == -4 and "no CPU path" in str(e.value)
    with pytest.raises(_ffi.TSearchError):
        merge_topk(np.zeros((2, 1, 3), np.float32), np.zeros((2, 1, 3), np.int64), 3)


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "theoremsearch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
