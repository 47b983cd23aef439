"""ctypes bindings for the TEST-INFRASTRUCTURE libraries under oracle/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (suffixarray_amd) never does.

  Oracle  -> oracle/libsa_oracle.so   own CPU restatement (oracle/sa_oracle.c)
  Ref     -> oracle/_ref/libsa_ref.so the reference compiled in place from /root/reference
                                     (libsais.c, libsais64.c, engine.c); see oracle/Makefile
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libsa_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libsa_ref.so")


def usable_threads(cap=16):
    """OpenMP team size that fits this process: affinity mask, cgroup CPU quota, and `cap`.  Letting
    OpenMP default to every core of the machine oversubscribes the GPU box's 16-core share badly."""
    n = max(1, len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def build(quiet=True):
    """Compile the oracle (always) and oracle/_ref (only where /root/reference exists)."""
    subprocess.run(["make", "-C", _HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


class PairU32(C.Structure):
    _fields_ = [("first", C.c_uint32), ("second", C.c_uint32)]


PAIR_DTYPE = np.dtype([("first", "<u4"), ("second", "<u4")])


def _u8(a):
    a = np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a,
                             dtype=np.uint8)
    return a


def pack_patterns(patterns):
    """list[bytes] -> (packed uint8 array, uint64 offsets[Q+1])."""
    off = np.zeros(len(patterns) + 1, dtype=np.uint64)
    if patterns:
        off[1:] = np.cumsum([len(p) for p in patterns], dtype=np.uint64)
    buf = np.frombuffer(b"".join(patterns), dtype=np.uint8).copy() if patterns else np.zeros(0, np.uint8)
    return buf, off


class Oracle:
    """The build's own CPU restatement."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build()
        L = self.lib = C.CDLL(ORACLE_SO)
        vp = C.c_void_p
        L.oracle_sais.restype = C.c_int32
        L.oracle_sais.argtypes = [vp, vp, C.c_int32, C.c_int32, vp]
        L.oracle_sais64.restype = C.c_int64
        L.oracle_sais64.argtypes = [vp, vp, C.c_int64, C.c_int64, vp]
        L.oracle_sa_naive.restype = None
        L.oracle_sa_naive.argtypes = [vp, C.c_uint32, vp]
        L.oracle_truncated_sa.restype = None
        L.oracle_truncated_sa.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
        L.oracle_get_substring_positions.restype = PairU32
        L.oracle_get_substring_positions.argtypes = [vp, C.c_uint64, vp, C.c_uint32, vp, C.c_uint32]
        L.oracle_query_batch.restype = C.c_int
        L.oracle_query_batch.argtypes = [vp, C.c_uint64, vp, C.c_uint32, vp, vp, C.c_uint64, vp, C.c_int]
        L.oracle_sufcheck.restype = C.c_int
        L.oracle_sufcheck.argtypes = [vp, C.c_uint64, vp]

    def sais(self, text, want_freq=False):
        t = _u8(text)
        n = t.size
        sa = np.empty(max(n, 1), dtype=np.int32)
        freq = np.zeros(256, dtype=np.int32)
        rc = self.lib.oracle_sais(t.ctypes.data, sa.ctypes.data, n, 0, freq.ctypes.data if want_freq else None)
        assert rc == 0, rc
        return (sa[:n], freq) if want_freq else sa[:n]

    def sais64(self, text):
        t = _u8(text)
        n = t.size
        sa = np.empty(max(n, 1), dtype=np.int64)
        rc = self.lib.oracle_sais64(t.ctypes.data, sa.ctypes.data, n, 0, None)
        assert rc == 0, rc
        return sa[:n]

    def sa_naive(self, text):
        t = _u8(text)
        sa = np.empty(max(t.size, 1), dtype=np.uint32)
        self.lib.oracle_sa_naive(t.ctypes.data, t.size, sa.ctypes.data)
        return sa[:t.size]

    def truncated_sa(self, text, max_suffix_length):
        t = _u8(text)
        sa = np.empty(max(t.size, 1), dtype=np.uint32)
        self.lib.oracle_truncated_sa(t.ctypes.data, t.size, max_suffix_length, sa.ctypes.data)
        return sa[:t.size]

    def query(self, text, sa, max_suffix_length, pattern):
        t = _u8(text)
        s = np.ascontiguousarray(sa, dtype=np.uint32)
        q = _u8(pattern)
        r = self.lib.oracle_get_substring_positions(t.ctypes.data, t.size, s.ctypes.data,
                                                    max_suffix_length, q.ctypes.data, q.size)
        return (r.first, r.second)

    def query_batch(self, text, sa, max_suffix_length, patterns, threads=None):
        """threads: None -> 1 for small batches, usable_threads() for large ones; 0 -> OpenMP default."""
        t = _u8(text)
        s = np.ascontiguousarray(sa, dtype=np.uint32)
        buf, off = patterns if isinstance(patterns, tuple) else pack_patterns(patterns)
        q = off.size - 1
        if threads is None:
            threads = 1 if q < 20000 else usable_threads()
        out = np.zeros(q, dtype=PAIR_DTYPE)
        buf = np.ascontiguousarray(buf)
        if buf.size == 0:
            buf = np.zeros(1, np.uint8)
        self.threads_used = self.lib.oracle_query_batch(
            t.ctypes.data, t.size, s.ctypes.data, max_suffix_length,
            buf.ctypes.data, off.ctypes.data, q, out.ctypes.data, threads)
        return out

    def sufcheck(self, text, sa):
        t = _u8(text)
        s = np.ascontiguousarray(sa, dtype=np.uint32)
        return self.lib.oracle_sufcheck(t.ctypes.data, t.size, s.ctypes.data)


class RefSuffixArrayStruct(C.Structure):
    """engine.h:123-130 SuffixArray_struct (sizeof == 40)."""
    _fields_ = [("suffix_array", C.c_void_p), ("is_quoted_bitflag", C.c_void_p),
                ("global_byte_start_idx", C.c_uint64), ("global_byte_end_idx", C.c_uint64),
                ("max_suffix_length", C.c_uint32), ("n", C.c_uint32)]


class Ref:
    """The reference itself (libsais 2.8.4 + engine.c), compiled in place by oracle/Makefile."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self):
        L = self.lib = C.CDLL(REF_SO)
        vp = C.c_void_p
        L.libsais.restype = C.c_int32
        L.libsais.argtypes = [vp, vp, C.c_int32, C.c_int32, vp]
        L.libsais_omp.restype = C.c_int32
        L.libsais_omp.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, C.c_int32]
        L.libsais64.restype = C.c_int64
        L.libsais64.argtypes = [vp, vp, C.c_int64, C.c_int64, vp]
        L.libsais64_omp.restype = C.c_int64
        L.libsais64_omp.argtypes = [vp, vp, C.c_int64, C.c_int64, vp, C.c_int64]
        L.get_substring_positions.restype = PairU32
        L.get_substring_positions.argtypes = [vp, C.POINTER(RefSuffixArrayStruct), C.c_char_p]
        L.construct_truncated_suffix_array.restype = None
        L.construct_truncated_suffix_array.argtypes = [vp, C.POINTER(RefSuffixArrayStruct)]

    def libsais(self, text, threads=1, want_freq=False):
        t = _u8(text)
        n = t.size
        sa = np.empty(max(n, 1), dtype=np.int32)
        freq = np.zeros(256, dtype=np.int32)
        fp = freq.ctypes.data if want_freq else None
        if threads == 1:
            rc = self.lib.libsais(t.ctypes.data, sa.ctypes.data, n, 0, fp)
        else:
            rc = self.lib.libsais_omp(t.ctypes.data, sa.ctypes.data, n, 0, fp, threads)
        assert rc == 0, rc
        return (sa[:n], freq) if want_freq else sa[:n]

    def libsais_into(self, t, sa, threads=0):
        """Timed form for bench.py: caller owns T (uint8) and SA (int32, first-touched)."""
        return self.lib.libsais_omp(t.ctypes.data, sa.ctypes.data, t.size, 0, None, threads)

    def libsais64(self, text, threads=1):
        t = _u8(text)
        n = t.size
        sa = np.empty(max(n, 1), dtype=np.int64)
        if threads == 1:
            rc = self.lib.libsais64(t.ctypes.data, sa.ctypes.data, n, 0, None)
        else:
            rc = self.lib.libsais64_omp(t.ctypes.data, sa.ctypes.data, n, 0, None, threads)
        assert rc == 0, rc
        return sa[:n]

    def libsais64_into(self, t, sa, threads=0):
        return self.lib.libsais64_omp(t.ctypes.data, sa.ctypes.data, t.size, 0, None, threads)

    def _struct(self, sa, n, max_suffix_length):
        st = RefSuffixArrayStruct()
        st.suffix_array = sa.ctypes.data
        st.is_quoted_bitflag = None
        st.global_byte_start_idx = 0
        st.global_byte_end_idx = n
        st.max_suffix_length = max_suffix_length
        st.n = n
        return st

    def query(self, text_nul, sa, max_suffix_length, pattern):
        """get_substring_positions (engine.c:869).  text_nul: uint8 text with a trailing NUL.
        Callers must avoid patterns <= the smallest suffix (reference defect)."""
        s = np.ascontiguousarray(sa, dtype=np.uint32)
        st = self._struct(s, s.size, max_suffix_length)
        r = self.lib.get_substring_positions(text_nul.ctypes.data, C.byref(st), bytes(pattern))
        return (r.first, r.second)

    def truncated_sa(self, text_padded, n, max_suffix_length):
        """construct_truncated_suffix_array (engine.c:837).  text_padded must have >= 32 readable
        bytes past n (strncmp_128 reads 16 bytes past the compared position, engine.c:664)."""
        sa = np.zeros(max(n, 1), dtype=np.uint32)
        st = self._struct(sa, n, max_suffix_length)
        self.lib.construct_truncated_suffix_array(text_padded.ctypes.data, C.byref(st))
        return sa[:n]
