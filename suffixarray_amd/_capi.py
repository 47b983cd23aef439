"""ctypes binding of the C ABI in include/sa_hip.h (libsa_hip.so, built in-tree by build.py).

This is the only way Python reaches the device code: no torch types cross the boundary, device
pointers travel as integers.  There is NO CPU fallback -- if the library is missing or no HIP
device is usable, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SA_HIP_LIB", os.path.join(_HERE, "libsa_hip.so"))   # SA_HIP_LIB: A/B builds in tools/

PAIR_DTYPE = np.dtype([("first", "<u4"), ("second", "<u4")])
UINT32_MAX = 0xFFFFFFFF

# every symbol include/sa_hip.h declares (tests check the library exports all of them)
EXPORTS = [
    "sa_hip_libsais", "sa_hip_libsais_omp", "sa_hip_libsais64", "sa_hip_libsais64_omp",
    "sa_hip_libsais64_device", "sa_hip_sufcheck64_device", "sa_hip_index_deep_keys",
    "sa_hip_last_call_breakdown", "sa_hip_release_workspace",
    "sa_hip_construct_truncated_suffix_array", "sa_hip_get_substring_positions",
    "sa_hip_device_count", "sa_hip_index_create", "sa_hip_index_destroy", "sa_hip_index_build",
    "sa_hip_index_build_device", "sa_hip_index_build_device64", "sa_hip_index_load", "sa_hip_index_load_device", "sa_hip_index_n",
    "sa_hip_index_max_suffix_length", "sa_hip_index_text_dev", "sa_hip_index_sa_dev",
    "sa_hip_index_stream", "sa_hip_index_get_sa_u32", "sa_hip_index_get_sa_i64", "sa_hip_index_widen_device",
    "sa_hip_index_get_freq", "sa_hip_query_batch", "sa_hip_query_batch_device", "sa_hip_query_batch_device_fixed",
    "sa_hip_index_get_sa_range", "sa_hip_index_query_hits", "sa_hip_index_sync", "sa_hip_index_verify", "sa_hip_index_build_stats",
    "sa_hip_index_set_rows", "sa_hip_index_query_rows", "sa_hip_index_query_rows_batch", "sa_hip_index_rows_for_range", "sa_hip_csv_index_copy_rows", "sa_hip_index_get_text", "sa_hip_csv_index_create", "sa_hip_csv_index_adopt",
    "sa_hip_get_matching_row_spans_file", "sa_hip_csv_index_destroy", "sa_hip_csv_index_create_partitioned", "sa_hip_csv_index_free_parts", "sa_hip_csv_index_handle", "sa_hip_csv_index_num_rows", "sa_hip_csv_index_num_columns",
    "sa_hip_csv_index_column_index", "sa_hip_csv_index_column_name", "sa_hip_csv_index_row_tables",
    "sa_hip_get_substring_positions_file", "sa_hip_get_matching_records_file", "sa_hip_get_matching_records", "sa_hip_free_records",
    "sa_hip_init_suffix_array_byte_idxs", "sa_hip_free_suffix_array", "sa_hip_write_suffix_array", "sa_hip_read_suffix_array",
    "sa_hip_comm_unique_id", "sa_hip_comm_create", "sa_hip_comm_destroy", "sa_hip_comm_rank", "sa_hip_comm_size",
    "sa_hip_comm_replicate_index", "sa_hip_comm_allgather_ranges",
    "sa_hip_index_replica_layout", "sa_hip_index_replica_buffers", "sa_hip_index_replica_reserve", "sa_hip_index_replica_commit",
    "sa_hip_index_query_stats", "sa_hip_csv_extract_column", "sa_hip_csv_free", "sa_hip_synth_csv", "sa_hip_sort_pairs", "sa_hip_synth_uniform27", "sa_hip_last_error", "sa_hip_version",
]


class PairU32(C.Structure):
    _fields_ = [("first", C.c_uint32), ("second", C.c_uint32)]


class SuffixArrayStruct(C.Structure):
    """engine.h:123-130 layout (sa_hip_SuffixArray_struct)."""
    _fields_ = [("suffix_array", C.c_void_p), ("is_quoted_bitflag", C.c_void_p),
                ("global_byte_start_idx", C.c_uint64), ("global_byte_end_idx", C.c_uint64),
                ("max_suffix_length", C.c_uint32), ("n", C.c_uint32)]


class BuildStats(C.Structure):
    _fields_ = [("n", C.c_uint64), ("sigma", C.c_uint32), ("bits_per_symbol", C.c_uint32),
                ("initial_chars", C.c_uint32), ("rounds", C.c_uint32), ("chunk_rounds", C.c_uint32),
                ("doubling_rounds", C.c_uint32), ("final_depth", C.c_uint32), ("radix_passes", C.c_uint32),
                ("radix_records", C.c_uint64), ("radix_bytes", C.c_uint64), ("active_total", C.c_uint64),
                ("tiny_resolved", C.c_uint64),
                ("radix_ms", C.c_double), ("total_ms", C.c_double),
                ("pass_ms", C.c_double * 4), ("pass_bytes", C.c_uint64 * 4), ("pass_launches", C.c_uint32 * 4),
                ("text_top_pass", C.c_uint32), ("narrow_k", C.c_uint32), ("widen_ms", C.c_double),
                ("finisher_records", C.c_uint64), ("finisher_resolved", C.c_uint64), ("finisher_runs", C.c_uint32),
                ("widen_fused", C.c_uint32), ("narrow48", C.c_uint32), ("lite_flags", C.c_uint32),
                ("period_resolved", C.c_uint64), ("split_plan", C.c_uint32), ("split_max", C.c_uint32)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if k.startswith("pass_") else getattr(self, k)) for k, _ in self._fields_}


class CsvColumn(C.Structure):
    _fields_ = [("text", C.c_void_p), ("text_len", C.c_uint64), ("row_text_starts", C.c_void_p),
                ("row_file_offsets", C.c_void_p), ("num_rows", C.c_uint64), ("column_names", C.c_void_p),
                ("num_columns", C.c_uint32), ("column_index", C.c_uint32)]


class QueryStats(C.Structure):
    _fields_ = [("q", C.c_uint64), ("kernel_ms", C.c_double), ("kernel_ms_sum", C.c_double), ("launches", C.c_uint32),
                ("pad_", C.c_uint32)]


class BigStats(C.Structure):
    """sa_hip_big_stats: the 64-bit-index build (texts beyond 2^32 - 2 bytes)."""
    _fields_ = [("sigma", C.c_uint32), ("bits_per_symbol", C.c_uint32), ("initial_chars", C.c_uint32), ("sort_passes", C.c_uint32),
                ("rounds", C.c_uint32), ("pad_", C.c_uint32), ("tied_after_sort", C.c_uint64), ("tied_total", C.c_uint64),
                ("total_ms", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "pad_"}


class CallBreakdown(C.Structure):
    _fields_ = [("n", C.c_uint64), ("workspace_reused", C.c_uint32), ("pad_", C.c_uint32), ("total_ms", C.c_double),
                ("workspace_ms", C.c_double), ("upload_ms", C.c_double), ("build_ms", C.c_double), ("build_device_ms", C.c_double),
                ("download_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "pad_"}


class ReplicaLayout(C.Structure):
    """sa_hip_replica_layout: what a replica must know about the index it copies (travels as bytes)."""
    _fields_ = [("n", C.c_uint64), ("max_suffix_length", C.c_uint32), ("key_bytes", C.c_uint32), ("bits_per_symbol", C.c_uint32),
                ("initial_chars", C.c_uint32), ("dir_bits", C.c_uint32), ("lo_shift", C.c_int32), ("dir_entries", C.c_uint64),
                ("code", C.c_uint16 * 256), ("freq", C.c_uint64 * 256)]


class ReplicaBuffers(C.Structure):
    _fields_ = [("text", C.c_void_p), ("sa", C.c_void_p), ("keys", C.c_void_p), ("dir", C.c_void_p),
                ("text_bytes", C.c_uint64), ("sa_bytes", C.c_uint64), ("keys_bytes", C.c_uint64), ("dir_bytes", C.c_uint64)]

    def items(self):
        """(device pointer, bytes) of every buffer that has to travel, in a fixed order"""
        return [(getattr(self, k) or 0, int(getattr(self, k + "_bytes"))) for k in ("text", "sa", "keys", "dir")]


class SaHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsa_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load libsa_hip.so (raises if it has not been built: there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: run `python -m suffixarray_amd.build` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, i32, i64 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_int64
    L.sa_hip_libsais.restype = i32
    L.sa_hip_libsais.argtypes = [vp, vp, i32, i32, vp]
    L.sa_hip_libsais_omp.restype = i32
    L.sa_hip_libsais_omp.argtypes = [vp, vp, i32, i32, vp, i32]
    L.sa_hip_libsais64.restype = i64
    L.sa_hip_libsais64.argtypes = [vp, vp, i64, i64, vp]
    L.sa_hip_libsais64_omp.restype = i64
    L.sa_hip_libsais64_omp.argtypes = [vp, vp, i64, i64, vp, i64]
    L.sa_hip_libsais64_device.restype = C.c_int
    L.sa_hip_libsais64_device.argtypes = [vp, vp, i64, C.c_int, C.POINTER(BigStats)]
    L.sa_hip_sufcheck64_device.restype = C.c_int
    L.sa_hip_sufcheck64_device.argtypes = [vp, vp, i64, C.c_int, C.POINTER(u64)]
    L.sa_hip_index_deep_keys.restype = C.c_int
    L.sa_hip_index_deep_keys.argtypes = [vp, C.c_int]
    L.sa_hip_last_call_breakdown.restype = C.c_int
    L.sa_hip_last_call_breakdown.argtypes = [C.POINTER(CallBreakdown)]
    L.sa_hip_release_workspace.restype = None
    L.sa_hip_release_workspace.argtypes = []
    L.sa_hip_construct_truncated_suffix_array.restype = C.c_int
    L.sa_hip_construct_truncated_suffix_array.argtypes = [vp, C.POINTER(SuffixArrayStruct)]
    L.sa_hip_get_substring_positions.restype = PairU32
    L.sa_hip_get_substring_positions.argtypes = [vp, C.POINTER(SuffixArrayStruct), C.c_char_p]
    L.sa_hip_device_count.restype = C.c_int
    L.sa_hip_device_count.argtypes = []
    L.sa_hip_index_create.restype = C.c_int
    L.sa_hip_index_create.argtypes = [C.POINTER(vp), u64, C.c_int]
    L.sa_hip_index_destroy.restype = None
    L.sa_hip_index_destroy.argtypes = [vp]
    L.sa_hip_index_build.restype = C.c_int
    L.sa_hip_index_build.argtypes = [vp, vp, u64, u32]
    L.sa_hip_index_build_device.restype = C.c_int
    L.sa_hip_index_build_device.argtypes = [vp, vp, u64, u32]
    L.sa_hip_index_build_device64.restype = C.c_int
    L.sa_hip_index_build_device64.argtypes = [vp, vp, u64, u32, vp]
    L.sa_hip_index_load.restype = C.c_int
    L.sa_hip_index_load.argtypes = [vp, vp, vp, u64, u32]
    L.sa_hip_index_load_device.restype = C.c_int
    L.sa_hip_index_load_device.argtypes = [vp, vp, vp, u64, u32]
    L.sa_hip_comm_unique_id.restype = C.c_int
    L.sa_hip_comm_unique_id.argtypes = [vp]
    L.sa_hip_comm_create.restype = C.c_int
    L.sa_hip_comm_create.argtypes = [C.POINTER(vp), vp, C.c_int, C.c_int, C.c_int]
    L.sa_hip_comm_destroy.restype = None
    L.sa_hip_comm_destroy.argtypes = [vp]
    L.sa_hip_comm_rank.restype = C.c_int
    L.sa_hip_comm_rank.argtypes = [vp]
    L.sa_hip_comm_size.restype = C.c_int
    L.sa_hip_comm_size.argtypes = [vp]
    L.sa_hip_comm_replicate_index.restype = C.c_int
    L.sa_hip_comm_replicate_index.argtypes = [vp, vp, C.c_int, C.POINTER(u64)]
    L.sa_hip_comm_allgather_ranges.restype = C.c_int
    L.sa_hip_comm_allgather_ranges.argtypes = [vp, vp, vp, u64, vp]
    L.sa_hip_index_replica_layout.restype = C.c_int
    L.sa_hip_index_replica_layout.argtypes = [vp, C.POINTER(ReplicaLayout)]
    L.sa_hip_index_replica_buffers.restype = C.c_int
    L.sa_hip_index_replica_buffers.argtypes = [vp, C.POINTER(ReplicaBuffers)]
    L.sa_hip_index_replica_reserve.restype = C.c_int
    L.sa_hip_index_replica_reserve.argtypes = [vp, C.POINTER(ReplicaLayout), C.POINTER(ReplicaBuffers)]
    L.sa_hip_index_replica_commit.restype = C.c_int
    L.sa_hip_index_replica_commit.argtypes = [vp]
    L.sa_hip_index_n.restype = u64
    L.sa_hip_index_n.argtypes = [vp]
    L.sa_hip_index_max_suffix_length.restype = u32
    L.sa_hip_index_max_suffix_length.argtypes = [vp]
    L.sa_hip_index_text_dev.restype = vp
    L.sa_hip_index_text_dev.argtypes = [vp]
    L.sa_hip_index_sa_dev.restype = vp
    L.sa_hip_index_sa_dev.argtypes = [vp]
    L.sa_hip_index_stream.restype = vp
    L.sa_hip_index_stream.argtypes = [vp]
    L.sa_hip_index_get_sa_u32.restype = C.c_int
    L.sa_hip_index_get_sa_u32.argtypes = [vp, vp]
    L.sa_hip_index_get_sa_i64.restype = C.c_int
    L.sa_hip_index_get_sa_i64.argtypes = [vp, vp]
    L.sa_hip_index_widen_device.restype = C.c_int
    L.sa_hip_index_widen_device.argtypes = [vp, vp]
    L.sa_hip_index_get_freq.restype = C.c_int
    L.sa_hip_index_get_freq.argtypes = [vp, vp]
    L.sa_hip_query_batch.restype = C.c_int
    L.sa_hip_query_batch.argtypes = [vp, vp, vp, u64, vp]
    L.sa_hip_query_batch_device.restype = C.c_int
    L.sa_hip_query_batch_device.argtypes = [vp, vp, vp, u64, vp]
    L.sa_hip_query_batch_device_fixed.restype = C.c_int
    L.sa_hip_query_batch_device_fixed.argtypes = [vp, vp, u64, u64, vp]
    L.sa_hip_index_get_sa_range.restype = C.c_int
    L.sa_hip_index_get_sa_range.argtypes = [vp, u64, u64, vp]
    L.sa_hip_index_query_hits.restype = C.c_int
    L.sa_hip_index_query_hits.argtypes = [vp, C.c_char_p, u64, C.c_uint32, C.POINTER(PairU32), vp, C.POINTER(C.c_uint32)]
    L.sa_hip_index_sync.restype = C.c_int
    L.sa_hip_index_sync.argtypes = [vp]
    L.sa_hip_index_verify.restype = C.c_int
    L.sa_hip_index_verify.argtypes = [vp, C.POINTER(u64)]
    L.sa_hip_index_build_stats.restype = C.c_int
    L.sa_hip_index_build_stats.argtypes = [vp, C.POINTER(BuildStats)]
    L.sa_hip_index_query_stats.restype = C.c_int
    L.sa_hip_index_query_stats.argtypes = [vp, C.POINTER(QueryStats)]
    L.sa_hip_csv_extract_column.restype = C.c_int
    L.sa_hip_csv_extract_column.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(CsvColumn)]
    L.sa_hip_csv_free.restype = None
    L.sa_hip_csv_free.argtypes = [C.POINTER(CsvColumn)]
    L.sa_hip_synth_csv.restype = C.c_int
    L.sa_hip_synth_csv.argtypes = [C.c_char_p, u64, u64]
    L.sa_hip_index_set_rows.restype = C.c_int
    L.sa_hip_index_set_rows.argtypes = [vp, vp, u64]
    L.sa_hip_index_query_rows.restype = C.c_int
    L.sa_hip_index_query_rows.argtypes = [vp, C.c_char_p, u64, u32, vp, C.POINTER(u32), C.POINTER(PairU32)]
    L.sa_hip_index_query_rows_batch.restype = C.c_int
    L.sa_hip_index_query_rows_batch.argtypes = [vp, vp, vp, u64, u32, vp, vp, vp]
    L.sa_hip_index_rows_for_range.restype = C.c_int
    L.sa_hip_index_rows_for_range.argtypes = [vp, PairU32, u32, vp, C.POINTER(u32)]
    L.sa_hip_index_get_text.restype = C.c_int
    L.sa_hip_index_get_text.argtypes = [vp, vp]
    L.sa_hip_csv_index_create.restype = C.c_int
    L.sa_hip_csv_index_create.argtypes = [C.POINTER(vp), C.c_char_p, C.c_char_p, u32, C.c_int]
    L.sa_hip_csv_index_adopt.restype = C.c_int
    L.sa_hip_csv_index_adopt.argtypes = [C.POINTER(vp), C.c_char_p, vp, vp, u64, vp, vp, u64, C.c_char_p, u32, u32, u32, C.c_int]
    L.sa_hip_csv_index_destroy.restype = None
    L.sa_hip_csv_index_destroy.argtypes = [vp]
    L.sa_hip_get_matching_row_spans_file.restype = C.c_int
    L.sa_hip_get_matching_row_spans_file.argtypes = [vp, C.c_char_p, u32, C.POINTER(C.c_char_p), C.POINTER(u32), C.POINTER(u32)]
    L.sa_hip_csv_index_create_partitioned.restype = C.c_int
    L.sa_hip_csv_index_create_partitioned.argtypes = [C.POINTER(C.POINTER(vp)), C.POINTER(u32), C.c_char_p, C.c_char_p, u32, C.c_int, u64]
    L.sa_hip_csv_index_free_parts.restype = None
    L.sa_hip_csv_index_free_parts.argtypes = [C.POINTER(vp)]
    L.sa_hip_csv_index_handle.restype = vp
    L.sa_hip_csv_index_handle.argtypes = [vp]
    L.sa_hip_csv_index_num_rows.restype = u64
    L.sa_hip_csv_index_num_rows.argtypes = [vp]
    L.sa_hip_csv_index_num_columns.restype = u32
    L.sa_hip_csv_index_num_columns.argtypes = [vp]
    L.sa_hip_csv_index_column_index.restype = u32
    L.sa_hip_csv_index_column_index.argtypes = [vp]
    L.sa_hip_csv_index_column_name.restype = C.c_char_p
    L.sa_hip_csv_index_column_name.argtypes = [vp, u32]
    L.sa_hip_csv_index_row_tables.restype = C.c_int
    L.sa_hip_csv_index_row_tables.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.sa_hip_csv_index_copy_rows.restype = C.c_int
    L.sa_hip_csv_index_copy_rows.argtypes = [vp, vp, u32, C.POINTER(vp)]
    L.sa_hip_get_substring_positions_file.restype = PairU32
    L.sa_hip_get_substring_positions_file.argtypes = [vp, C.c_char_p]
    L.sa_hip_get_matching_records_file.restype = C.c_int
    L.sa_hip_get_matching_records_file.argtypes = [vp, C.c_char_p, u32, C.POINTER(vp), C.POINTER(u32)]
    L.sa_hip_get_matching_records.restype = u32
    L.sa_hip_get_matching_records.argtypes = [vp, C.POINTER(SuffixArrayStruct), C.c_char_p, u32, C.POINTER(vp)]
    L.sa_hip_free_records.restype = None
    L.sa_hip_free_records.argtypes = [C.POINTER(vp), u32]
    L.sa_hip_init_suffix_array_byte_idxs.restype = C.c_int
    L.sa_hip_init_suffix_array_byte_idxs.argtypes = [C.POINTER(SuffixArrayStruct), u32, u64, u64, u32]
    L.sa_hip_free_suffix_array.restype = None
    L.sa_hip_free_suffix_array.argtypes = [C.POINTER(SuffixArrayStruct)]
    L.sa_hip_write_suffix_array.restype = C.c_int
    L.sa_hip_write_suffix_array.argtypes = [C.POINTER(SuffixArrayStruct), C.c_char_p, C.c_char_p]
    L.sa_hip_read_suffix_array.restype = C.c_int
    L.sa_hip_read_suffix_array.argtypes = [C.POINTER(SuffixArrayStruct), C.c_char_p]
    L.sa_hip_sort_pairs.restype = C.c_int
    L.sa_hip_sort_pairs.argtypes = [vp, vp, u64, C.c_int, C.c_int, C.c_int]
    L.sa_hip_synth_uniform27.restype = None
    L.sa_hip_synth_uniform27.argtypes = [vp, u64, u64]
    L.sa_hip_last_error.restype = C.c_char_p
    L.sa_hip_last_error.argtypes = []
    L.sa_hip_version.restype = C.c_char_p
    L.sa_hip_version.argtypes = []
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise SaHipError(rc, lib().sa_hip_last_error().decode("utf-8", "replace"))


def as_u8(a):
    if isinstance(a, (bytes, bytearray, memoryview)):
        return np.frombuffer(a, dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


def pack_patterns(patterns):
    """list[bytes] -> (packed uint8 array, uint64 offsets[Q+1])"""
    off = np.zeros(len(patterns) + 1, dtype=np.uint64)
    if len(patterns):
        off[1:] = np.cumsum([len(p) for p in patterns], dtype=np.uint64)
    buf = np.frombuffer(b"".join(patterns), dtype=np.uint8) if len(patterns) else np.zeros(0, np.uint8)
    return buf, off


class DeviceIndex:
    """Handle API: text + suffix array resident in HBM (sa_hip_index)."""

    def __init__(self, n_max, device=0):
        self._h = C.c_void_p()
        self._lib = lib()
        check(self._lib.sa_hip_index_create(C.byref(self._h), int(n_max), int(device)))

    @classmethod
    def from_handle(cls, handle, owner=None):
        """Non-owning wrapper of an existing sa_hip_index* (an integer); `owner` is kept alive with it."""
        self = cls.__new__(cls)
        self._h = C.c_void_p(int(handle))
        self._lib = lib()
        self._borrowed = True
        self._owner = owner
        return self

    def close(self):
        if getattr(self, "_borrowed", False):
            self._h = C.c_void_p()
            return
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.sa_hip_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- construction -------------------------------------------------------------------------
    def build(self, text, max_suffix_length=0):
        t = as_u8(text)
        check(self._lib.sa_hip_index_build(self._h, t.ctypes.data if t.size else None, t.size, max_suffix_length))
        return self

    def build_device(self, text_dev_ptr, n, max_suffix_length=0):
        check(self._lib.sa_hip_index_build_device(self._h, text_dev_ptr, n, max_suffix_length))
        return self

    def build_device64(self, text_dev_ptr, n, sa64_dev_ptr, max_suffix_length=0):
        """Device build that also leaves the suffix array in libsais64 layout (int64[n]) in a device buffer."""
        check(self._lib.sa_hip_index_build_device64(self._h, text_dev_ptr, n, max_suffix_length, sa64_dev_ptr))
        return self

    def load(self, text, sa, max_suffix_length=0):
        t = as_u8(text)
        s = np.ascontiguousarray(sa, dtype=np.uint32)
        assert s.size == t.size
        check(self._lib.sa_hip_index_load(self._h, t.ctypes.data if t.size else None,
                                          s.ctypes.data if s.size else None, t.size, max_suffix_length))
        return self

    def load_device(self, text_dev_ptr, sa_dev_ptr, n, max_suffix_length=0):
        check(self._lib.sa_hip_index_load_device(self._h, text_dev_ptr, sa_dev_ptr, n, max_suffix_length))
        return self

    # -- replicas: the query structures travel as they are (no rebuilding on the receiving side) ------------
    def replica_layout(self):
        lay = ReplicaLayout()
        check(self._lib.sa_hip_index_replica_layout(self._h, C.byref(lay)))
        return lay

    def replica_buffers(self):
        """Device buffers of this (built) index: the source of a replication."""
        b = ReplicaBuffers()
        check(self._lib.sa_hip_index_replica_buffers(self._h, C.byref(b)))
        return b

    def replica_reserve(self, layout):
        """Allocate buffers for `layout`; returns where the caller has to put the data (then replica_commit)."""
        b = ReplicaBuffers()
        check(self._lib.sa_hip_index_replica_reserve(self._h, C.byref(layout), C.byref(b)))
        return b

    def replica_commit(self):
        check(self._lib.sa_hip_index_replica_commit(self._h))

    @property
    def stream(self):
        """The index's hipStream_t as an integer (torch.cuda.ExternalStream wraps it for event ordering)."""
        return self._lib.sa_hip_index_stream(self._h)

    # -- accessors ----------------------------------------------------------------------------
    @property
    def n(self):
        return int(self._lib.sa_hip_index_n(self._h))

    @property
    def max_suffix_length(self):
        return int(self._lib.sa_hip_index_max_suffix_length(self._h))

    @property
    def text_dev(self):
        return self._lib.sa_hip_index_text_dev(self._h)

    @property
    def sa_dev(self):
        return self._lib.sa_hip_index_sa_dev(self._h)

    def text(self):
        """The indexed text (n bytes) copied back to the host (CSV mode: the extracted column)."""
        out = np.empty(max(self.n, 1), dtype=np.uint8)
        check(self._lib.sa_hip_index_get_text(self._h, out.ctypes.data))
        return out[:self.n]

    def sa_u32(self):
        out = np.empty(max(self.n, 1), dtype=np.uint32)
        check(self._lib.sa_hip_index_get_sa_u32(self._h, out.ctypes.data))
        return out[:self.n]

    def sa_i64(self):
        out = np.empty(max(self.n, 1), dtype=np.int64)
        check(self._lib.sa_hip_index_get_sa_i64(self._h, out.ctypes.data))
        return out[:self.n]

    def widen_device(self, out_dev_ptr):
        """int64[n] libsais64-layout copy of the suffix array into a device buffer (asynchronous)."""
        check(self._lib.sa_hip_index_widen_device(self._h, out_dev_ptr))

    def sa_range(self, first, count):
        out = np.empty(max(count, 1), dtype=np.uint32)
        check(self._lib.sa_hip_index_get_sa_range(self._h, first, count, out.ctypes.data))
        return out[:count]

    def query_hits(self, pattern: bytes, max_hits: int = 4096):
        """One query and its first hits in one call: ((first, second), SA[first .. first + nhits))."""
        rng = PairU32()
        hits = np.empty(max(min(max_hits, 4096), 1), dtype=np.uint32)
        nh = C.c_uint32(0)
        check(self._lib.sa_hip_index_query_hits(self._h, pattern, len(pattern), min(max_hits, 4096), C.byref(rng),
                                                hits.ctypes.data, C.byref(nh)))
        return (rng.first, rng.second), hits[:nh.value]

    def freq(self):
        out = np.zeros(256, dtype=np.uint64)
        check(self._lib.sa_hip_index_get_freq(self._h, out.ctypes.data))
        return out

    def sync(self):
        check(self._lib.sa_hip_index_sync(self._h))

    def deep_keys(self, mode=2):
        """Second-level keys for patterns longer than the key (sa_hip_index_deep_keys): 2 = build now, 1 = large batches build
        them (default of a handle), 0 = drop and never build.  True when the index has them afterwards."""
        rc = self._lib.sa_hip_index_deep_keys(self._h, mode)
        if rc < 0:
            check(rc)
        return rc == 1

    def prepare_deep_keys(self):
        return self.deep_keys(2)

    def verify(self):
        """Number of violations of the suffix-array property found on the device (0 = verified)."""
        v = C.c_uint64(0)
        check(self._lib.sa_hip_index_verify(self._h, C.byref(v)))
        return int(v.value)

    def build_stats(self):
        st = BuildStats()
        check(self._lib.sa_hip_index_build_stats(self._h, C.byref(st)))
        return st.as_dict()

    def query_stats(self):
        st = QueryStats()
        check(self._lib.sa_hip_index_query_stats(self._h, C.byref(st)))
        return {"q": st.q, "kernel_ms": st.kernel_ms, "kernel_ms_sum": st.kernel_ms_sum, "launches": st.launches}

    # -- query ----------------------------------------------------------------------------------
    def query_batch(self, patterns):
        """patterns: list[bytes] or (packed uint8, uint64 offsets).  -> structured array (first, second)."""
        buf, off = patterns if isinstance(patterns, tuple) else pack_patterns(patterns)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        q = off.size - 1
        out = np.zeros(max(q, 1), dtype=PAIR_DTYPE)
        if q:
            check(self._lib.sa_hip_query_batch(self._h, buf.ctypes.data if buf.size else None, off.ctypes.data, q,
                                               out.ctypes.data))
        return out[:q]

    def set_rows(self, row_text_starts):
        r = np.ascontiguousarray(row_text_starts, dtype=np.uint64)
        check(self._lib.sa_hip_index_set_rows(self._h, r.ctypes.data if r.size else None, r.size))

    def query_rows(self, pattern: bytes, k):
        """ONE query -> (row ids in SA order of their first hit, (first, second))."""
        rows = np.empty(max(k, 1), dtype=np.uint64)
        n = C.c_uint32(0)
        rng = PairU32()
        check(self._lib.sa_hip_index_query_rows(self._h, pattern, len(pattern), k, rows.ctypes.data, C.byref(n), C.byref(rng)))
        return rows[:n.value].copy(), (rng.first, rng.second)

    def query_rows_batch(self, patterns, k):
        """A batch -> (list of row-id arrays, structured ranges): one search launch + one rows launch."""
        buf, off = patterns if isinstance(patterns, tuple) else pack_patterns(patterns)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        q = off.size - 1
        rows = np.empty((max(q, 1), max(k, 1)), dtype=np.uint64)
        counts = np.zeros(max(q, 1), dtype=np.uint32)
        ranges = np.zeros(max(q, 1), dtype=PAIR_DTYPE)
        if q:
            check(self._lib.sa_hip_index_query_rows_batch(self._h, buf.ctypes.data if buf.size else None, off.ctypes.data, q, k,
                                                          rows.ctypes.data, counts.ctypes.data, ranges.ctypes.data))
        return [rows[i, :counts[i]].copy() for i in range(q)], ranges[:q]

    def query_rows_batch_raw(self, patterns, k, out=None):
        """query_rows_batch without the per-query Python list: ((row_ids uint64[Q, k], counts uint32[Q]), ranges).
        out: (rows, counts, ranges) of an earlier call with the same Q and k, to be overwritten -- the C entry point fills
        caller-provided arrays (as engine.c:1326 does), and a loop that answers batch after batch keeps its arrays instead of
        mapping and unmapping Q x k x 8 bytes per call (1.28 GB at Q = 1e7, k = 16: ~60 ms of page-table work per step)."""
        buf, off = patterns if isinstance(patterns, tuple) else pack_patterns(patterns)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        q = off.size - 1
        if out is not None:
            rows, counts, ranges = out
            if (rows.shape != (max(q, 1), max(k, 1)) or rows.dtype != np.uint64 or counts.shape != (max(q, 1),) or counts.dtype != np.uint32
                    or ranges.shape != (max(q, 1),) or ranges.dtype != PAIR_DTYPE
                    or not (rows.flags.c_contiguous and counts.flags.c_contiguous and ranges.flags.c_contiguous)):
                raise ValueError("query_rows_batch_raw: `out` does not fit this batch")
        else:
            rows = np.empty((max(q, 1), max(k, 1)), dtype=np.uint64)
            counts = np.zeros(max(q, 1), dtype=np.uint32)
            ranges = np.zeros(max(q, 1), dtype=PAIR_DTYPE)
        if q:
            check(self._lib.sa_hip_index_query_rows_batch(self._h, buf.ctypes.data if buf.size else None, off.ctypes.data, q, k,
                                                          rows.ctypes.data, counts.ctypes.data, ranges.ctypes.data))
        return (rows[:q], counts[:q]), ranges[:q]

    def query_batch_device(self, patterns_dev_ptr, offsets_dev_ptr, q, out_dev_ptr):
        check(self._lib.sa_hip_query_batch_device(self._h, patterns_dev_ptr, offsets_dev_ptr, q, out_dev_ptr))

    def query_batch_device_fixed(self, patterns_dev_ptr, pattern_len, q, out_dev_ptr):
        """q patterns of pattern_len bytes each, packed back to back in device memory (no offsets array)."""
        check(self._lib.sa_hip_query_batch_device_fixed(self._h, patterns_dev_ptr, pattern_len, q, out_dev_ptr))


class Comm:
    """sa_hip_comm: RCCL communicator of the C ABI (one process per GPU; no torch involved)."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(lib().sa_hip_comm_unique_id(buf))
        return buf.raw

    def __init__(self, unique_id: bytes, nranks, rank, device=0):
        self._lib = lib()
        self._h = C.c_void_p()
        check(self._lib.sa_hip_comm_create(C.byref(self._h), unique_id, nranks, rank, device))

    @property
    def rank(self):
        return self._lib.sa_hip_comm_rank(self._h)

    @property
    def size(self):
        return self._lib.sa_hip_comm_size(self._h)

    def replicate_index(self, idx, root=0):
        n = C.c_uint64(0)
        check(self._lib.sa_hip_comm_replicate_index(self._h, idx._h, root, C.byref(n)))
        return int(n.value)

    def allgather_ranges(self, idx, send_dev_ptr, pairs_per_rank, recv_dev_ptr):
        check(self._lib.sa_hip_comm_allgather_ranges(self._h, idx._h, send_dev_ptr, pairs_per_rank, recv_dev_ptr))

    def close(self):
        if self._h:
            self._lib.sa_hip_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# -- libsais- / engine-compatible one-shot wrappers ----------------------------------------------

def last_call_breakdown():
    """Where the time of the last libsais-compatible one-shot call of this process went (sa_hip_call_breakdown)."""
    b = CallBreakdown()
    check(lib().sa_hip_last_call_breakdown(C.byref(b)))
    return b.as_dict()


def release_workspace():
    lib().sa_hip_release_workspace()


def libsais(text, want_freq=False):
    t = as_u8(text)
    sa = np.empty(max(t.size, 1), dtype=np.int32)
    freq = np.zeros(256, dtype=np.int32)
    rc = lib().sa_hip_libsais(t.ctypes.data, sa.ctypes.data, t.size, 0, freq.ctypes.data if want_freq else None)
    check(rc)
    return (sa[:t.size], freq) if want_freq else sa[:t.size]


def libsais64(text, want_freq=False):
    t = as_u8(text)
    sa = np.empty(max(t.size, 1), dtype=np.int64)
    freq = np.zeros(256, dtype=np.int64)
    rc = lib().sa_hip_libsais64(t.ctypes.data, sa.ctypes.data, t.size, 0, freq.ctypes.data if want_freq else None)
    check(int(rc))
    return (sa[:t.size], freq) if want_freq else sa[:t.size]


def libsais64_device(text_ptr, sa_ptr, n, device=0):
    """The 64-bit-index build (csrc/big_build.hpp) on device buffers: text_ptr = n bytes, sa_ptr = n int64 entries.  Returns its stats."""
    st = BigStats()
    check(lib().sa_hip_libsais64_device(text_ptr, sa_ptr, n, device, C.byref(st)))
    return st.as_dict()


def sufcheck64_device(text_ptr, sa_ptr, n, device=0):
    """Slots at which the int64 array on the device is not the suffix array of the text (0 = it is)."""
    v = C.c_uint64(0)
    check(lib().sa_hip_sufcheck64_device(text_ptr, sa_ptr, n, device, C.byref(v)))
    return int(v.value)


def construct_truncated_suffix_array(text, max_suffix_length):
    t = as_u8(text)
    sa = np.zeros(max(t.size, 1), dtype=np.uint32)
    st = SuffixArrayStruct()
    st.suffix_array = sa.ctypes.data
    st.max_suffix_length = max_suffix_length
    st.n = t.size
    st.global_byte_end_idx = t.size
    check(lib().sa_hip_construct_truncated_suffix_array(t.ctypes.data, C.byref(st)))
    return sa[:t.size]


def get_substring_positions(text, sa, max_suffix_length, substring):
    t = as_u8(text)
    s = np.ascontiguousarray(sa, dtype=np.uint32)
    st = SuffixArrayStruct()
    st.suffix_array = s.ctypes.data
    st.max_suffix_length = max_suffix_length
    st.n = t.size
    r = lib().sa_hip_get_substring_positions(t.ctypes.data, C.byref(st), bytes(substring))
    return (r.first, r.second)


class CsvIndex:
    """sa_hip_csv_index through ctypes: the C seam of CSV mode exactly as a C caller sees it (the tests of the record
    retrieval entry points go through this; the Python class binds the same functions from Cython)."""

    def __init__(self, csv_file, search_column, max_suffix_length=32, device=0):
        self._lib = lib()
        self._h = C.c_void_p()
        check(self._lib.sa_hip_csv_index_create(C.byref(self._h), os.fsencode(csv_file), search_column.encode("utf-8"),
                                                int(max_suffix_length), int(device)))

    def close(self):
        if self._h:
            self._lib.sa_hip_csv_index_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def index(self):
        return DeviceIndex.from_handle(self._lib.sa_hip_csv_index_handle(self._h), self)

    @property
    def num_rows(self):
        return int(self._lib.sa_hip_csv_index_num_rows(self._h))

    @property
    def columns(self):
        return [self._lib.sa_hip_csv_index_column_name(self._h, i).decode("utf-8") for i in range(self._lib.sa_hip_csv_index_num_columns(self._h))]

    def get_substring_positions_file(self, substring: bytes):
        r = self._lib.sa_hip_get_substring_positions_file(self._h, substring)
        return (r.first, r.second)

    def get_matching_records_file(self, substring: bytes, k, already=()):
        """-> list of row bytes; `already`: rows that occupy the front of the table (num_matches starts at their count)."""
        table = (C.c_void_p * max(k, 1))()
        num = C.c_uint32(len(already))
        check(self._lib.sa_hip_get_matching_records_file(self._h, substring, k, table, C.byref(num)))
        out = [C.string_at(table[i]) for i in range(len(already), num.value)]
        tail = (C.c_void_p * max(num.value - len(already), 1))(*[table[i] for i in range(len(already), num.value)])
        self._lib.sa_hip_free_records(tail, num.value - len(already))
        return out, num.value


def get_matching_records(text, sa, max_suffix_length, substring: bytes, k):
    """sa_hip_get_matching_records: host text + host SA, engine.c:1168-1215's calling convention."""
    t = as_u8(text)
    s = np.ascontiguousarray(sa, dtype=np.uint32)
    st = SuffixArrayStruct()
    st.suffix_array = s.ctypes.data
    st.max_suffix_length = max_suffix_length
    st.n = t.size
    table = (C.c_void_p * max(k, 1))()
    n = lib().sa_hip_get_matching_records(t.ctypes.data, C.byref(st), substring, k, table)
    out = [C.string_at(table[i]) for i in range(n)]
    lib().sa_hip_free_records(table, n)
    return out


class _CsvOwner:
    """Keeps the malloc'ed arrays of one sa_hip_csv_column alive for the numpy views made of them."""

    def __init__(self, col):
        self.col = col

    def __del__(self):
        try:
            lib().sa_hip_csv_free(C.byref(self.col))
        except Exception:
            pass


def _view(owner, ptr, count, ctype, dtype):
    if not count:
        return np.zeros(0, dtype)
    buf = (ctype * count).from_address(C.cast(ptr, C.c_void_p).value)
    buf._owner = owner   # the view's base is this ctypes array: the owner lives as long as any view does
    return np.frombuffer(buf, dtype=dtype)


def csv_extract_column(path, column, copy=True):
    """Native RFC-4180 column extractor -> (columns, text, row_text_starts, row_file_offsets).
    copy=True: text as bytes, the offsets as numpy arrays of their own.  copy=False: text as a uint8 array and the
    offsets as views of the arrays the extractor allocated (freed when the last view goes): no second copy of the
    ~2 bytes + 16 bytes per row that a 50M-row file yields (0.3 s of the 0.9 s the copying form takes)."""
    col = CsvColumn()
    check(lib().sa_hip_csv_extract_column(os.fsencode(path), column.encode("utf-8"), C.byref(col)))
    owner = _CsvOwner(col)
    names, p = [], col.column_names
    for _ in range(col.num_columns):
        s = C.string_at(p)
        names.append(s.decode("utf-8"))
        p += len(s) + 1
    text = _view(owner, col.text, col.text_len, C.c_uint8, np.uint8)
    starts = _view(owner, col.row_text_starts, col.num_rows, C.c_int64, np.int64)   # offsets are < 2^63: viewed as int64
    offs = _view(owner, col.row_file_offsets, col.num_rows + 1, C.c_int64, np.int64)
    if copy:
        return names, text.tobytes(), starts.copy(), offs.copy()
    return names, text, starts, offs


def synth_csv(path, rows, seed=1):
    check(lib().sa_hip_synth_csv(os.fsencode(path), rows, seed))


def sort_pairs(keys, values=None, begin_bit=0, end_bit=64, device=0):
    """In-place stable device sort of (u64 key, u32 value) records; returns (keys, values)."""
    k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
    v = None if values is None else np.ascontiguousarray(values, dtype=np.uint32).copy()
    check(lib().sa_hip_sort_pairs(k.ctypes.data if k.size else None, None if v is None else v.ctypes.data,
                                  k.size, begin_bit, end_bit, device))
    return k, v


def synth_uniform27(n, seed=88172645463325252):
    out = np.empty(n, dtype=np.uint8)
    lib().sa_hip_synth_uniform27(out.ctypes.data, n, seed)
    return out
