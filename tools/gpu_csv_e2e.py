"""End-to-end phases of SuffixArray(csv_file=...) at config-5 size: SA_HIP_DIAG=1 SA_HIP_CSV_TIMING=1 python3 tools/gpu_csv_e2e.py [rows]"""
import os, sys, time
import torch   # first: its HIP runtime is the one this process uses (libsa_hip.so loaded before torch would bring the system's copy)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import SuffixArray, _capi
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
path = "/tmp/e2e_%d.csv" % rows
t0 = time.perf_counter(); _capi.synth_csv(path, rows, 1); print("synth %.2f s" % (time.perf_counter() - t0), flush=True)
torch.cuda.init(); torch.zeros(1, device="cuda")   # the HIP runtime is up, as in a serving process
for rep in range(2):
    t0 = time.perf_counter()
    sa = SuffixArray(csv_file=path, search_column="company_name", max_suffix_length=32)
    t1 = time.perf_counter()
    st = sa._index.build_stats()
    print("rep %d: end to end %.3f s (device build %.1f ms)" % (rep, t1 - t0, st["total_ms"]), flush=True)
    sa.close()
os.remove(path)
