"""trinity_amd — MI355X-native execution engine for Trinity's query hot path.

The product is libtrinity_hip.so (hand-written HIP for gfx950 behind the C-ABI of include/trinity_hip.h) plus
the host-side C++ mirror of Trinity's operator surface (csrc/host/).  This Python package is only the thin
ctypes harness used by tests/ and bench.py: it loads the in-tree shared objects and FAILS LOUDLY when they are
missing — there is no CPU fallback anywhere in this package.
"""
from .build import build_all, LIB_HIP, LIB_HOST  # noqa: F401
from .engine import Device, Index, Batch, CollectionBatch, Segment, TrinityError, tok, gen_queries  # noqa: F401
from .engine import OP_TERM, OP_AND, OP_OR, OP_PHRASE, OP_NOT, OP_OPT, OP_SOME, FLAG_DOCUMENTS_ONLY, FLAG_ACCUMULATED_SCORE, FLAG_MATCHED_TERMS, FLAG_HIT_PAYLOADS  # noqa: F401
