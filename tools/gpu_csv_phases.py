"""Where the time of SuffixArray(csv_file=...) goes on the GPU box's host (config 5 shape): native extractor phases
(SA_HIP_CSV_TIMING), the Python wrapper around it, index creation + first build."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SA_HIP_CSV_TIMING"] = "1"
import ctypes as C
from suffixarray_amd import _capi, SuffixArray
from suffixarray_amd.csv_ingest import extract_column
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
path = "/tmp/companies_%d.csv" % rows
t0 = time.time(); _capi.synth_csv(path, rows, 1); print("gen %.1fs, %.2f GB" % (time.time() - t0, os.path.getsize(path) / 1e9), flush=True)
# first thing in the process, as a user would see it (HIP runtime cold, no device memory held yet)
t0 = time.time(); s = SuffixArray(csv_file=path, search_column="company_name", max_suffix_length=32); print("SuffixArray(csv_file), fresh process %.3fs" % (time.time() - t0), flush=True)
s.close(); del s
lib = _capi.lib()
for rep in range(2):
    col = _capi.CsvColumn()
    t0 = time.time(); lib.sa_hip_csv_extract_column(path.encode(), b"company_name", C.byref(col)); t1 = time.time() - t0
    t0 = time.time(); lib.sa_hip_csv_free(C.byref(col)); t2 = time.time() - t0
    print("C call %.3fs free %.3fs" % (t1, t2), flush=True)
for rep in range(2):
    t0 = time.time(); c = extract_column(path, "company_name"); print("extract_column (wrapper) %.3fs" % (time.time() - t0), flush=True)
n = len(c.text)
t0 = time.time(); idx = _capi.DeviceIndex(n, 0); t1 = time.time() - t0
t0 = time.time(); idx.build(c.text, 32); t2 = time.time() - t0
t0 = time.time(); idx.build(c.text, 32); t3 = time.time() - t0
print("DeviceIndex create %.3fs, first build (host call) %.3fs, second %.3fs, device build %.1f ms" % (t1, t2, t3, idx.build_stats()["total_ms"]), flush=True)
idx.close()
t0 = time.time(); s = SuffixArray(csv_file=path, search_column="company_name", max_suffix_length=32); print("SuffixArray(csv_file) again (24 GB of device memory just freed) %.3fs" % (time.time() - t0), flush=True)
os.remove(path)
