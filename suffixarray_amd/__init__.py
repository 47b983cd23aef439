"""suffixarray_amd -- MI355X-native suffix-array index (construction + batched substring query).

Drop-in for the hot path of jdm365/SuffixArray: `SuffixArray(documents=... | csv_file=...,
search_column=..., max_suffix_length=...)` / `query_records(substring, k)` on top of the C ABI
in include/sa_hip.h (libsa_hip.so, hand-written HIP for gfx950).  No CPU fallback.
"""
from . import _capi
from ._capi import DeviceIndex, SaHipError, PAIR_DTYPE, UINT32_MAX  # noqa: F401


def __getattr__(name):
    # the Cython extension is imported on first use (it is built in-tree: `python -m suffixarray_amd.build`)
    if name == "SuffixArray":
        from . import index
        return index.SuffixArray
    raise AttributeError(name)


__all__ = ["SuffixArray", "DeviceIndex", "SaHipError", "PAIR_DTYPE", "UINT32_MAX"]
