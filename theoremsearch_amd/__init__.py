"""theoremsearch_amd - MI355X-native embedding similarity + exact top-k search for TheoremSearch.

The compute path is libtsearch.so (hand-written HIP kernels for gfx950) behind the C ABI of
include/tsearch.h; see DESIGN.md.  Importing the package does not load the library; the first
use does, and fails loudly when it is missing.
"""
from ._ffi import TSearchError  # noqa: F401
from .index import TheoremIndex, Timer, merge_topk  # noqa: F401

__all__ = ["TheoremIndex", "Timer", "merge_topk", "TSearchError"]
__version__ = "0.1.0"
