"""Where does the first build of a process spend its time? (handle creation, first build, second build)"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
from suffixarray_amd import _capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
text = synth.d1_uniform27(n)
t1 = time.perf_counter()
print("import + text %.2fs" % (t1 - t0), flush=True)
t1 = time.perf_counter(); idx = _capi.DeviceIndex(n, 0); t2 = time.perf_counter()
print("DeviceIndex(n) %.1f ms" % ((t2 - t1) * 1e3), flush=True)
for i in range(3):
    t2 = time.perf_counter(); idx.build(text); idx.sync() if hasattr(idx, "sync") else None; t3 = time.perf_counter()
    print("build %d wall %.1f ms (device %.1f ms)" % (i, (t3 - t2) * 1e3, idx.build_stats()["total_ms"]), flush=True)
idx.close()
t1 = time.perf_counter(); idx = _capi.DeviceIndex(n, 0); t2 = time.perf_counter()
print("second DeviceIndex(n) %.1f ms" % ((t2 - t1) * 1e3), flush=True)
t2 = time.perf_counter(); idx.build(text); t3 = time.perf_counter()
print("its first build wall %.1f ms (device %.1f ms)" % ((t3 - t2) * 1e3, idx.build_stats()["total_ms"]), flush=True)
