"""`SuffixArray` -- the reference's Python class (suffix_array/suffix_array.pyx:110-267, README.md:13-50).

There is ONE implementation: the Cython class of suffix_array.pyx on top of the C seam (include/sa_hip.h).  This
module only makes sure the extension is built (in-tree, `python -m suffixarray_amd.build`) and re-exports it.
"""


def _load():
    try:
        from .suffix_array import SuffixArray as cls
    except ImportError:
        from .build import build_lib, build_cython
        build_lib()
        build_cython()
        from .suffix_array import SuffixArray as cls
    return cls


def __getattr__(name):
    if name == "SuffixArray":
        cls = _load()
        globals()["SuffixArray"] = cls
        return cls
    raise AttributeError(name)


def _ascii_lower(b: bytes) -> bytes:
    """pyx:103-107 lowercase_string: only bytes 65..90 are changed (used by the Python CSV state machine of the tests)."""
    return b.translate(_LOWER)


_LOWER = bytes((c + 32) if 65 <= c <= 90 else c for c in range(256))
