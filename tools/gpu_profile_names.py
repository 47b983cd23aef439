"""Kernel-level view of a refinement-heavy build: company-name column (config-5 shape), L = 32 and full.
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_names --output-format csv -- python3 tools/gpu_profile_names.py [rows]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi
from suffixarray_amd.csv_ingest import extract_column

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
path = "/tmp/companies_%d.csv" % rows
_capi.synth_csv(path, rows, 1)
col = extract_column(path, "company_name")
os.remove(path)
import numpy as np
t = np.frombuffer(col.text, dtype=np.uint8)
for L in (32, 0):
    with _capi.DeviceIndex(t.size, 0) as idx:
        idx.build(t, L)
        idx.build(t, L)
        st = idx.build_stats()
        print("names rows %d n=%d L=%d total %.2f ms (%.2f Gchars/s) radix %.2f ms passes %d k0=%d rounds %d (chunk %d dbl %d) active_total %d tiny %d verify=%d" % (
            rows, t.size, L, st["total_ms"], t.size / st["total_ms"] / 1e6, st["radix_ms"], st["radix_passes"], st["initial_chars"], st["rounds"],
            st["chunk_rounds"], st["doubling_rounds"], st["active_total"], st.get("tiny_resolved", -1), idx.verify()), flush=True)
