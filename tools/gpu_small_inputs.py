"""Build latency of small inputs (launch-bound): device time and host wall time of the second build on a warm handle."""
import os, sys, time
os.environ.setdefault("SA_HIP_DIAG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from suffixarray_amd import _capi
texts = cases.small_texts()
readme = np.frombuffer("\n".join(["the quick brown fox jumps over the lazy dog", "i am going to the store to buy some milk",
                                  "uhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhhh"]).encode(), np.uint8)
sel = [("readme_3docs", readme, 32), ("mississippi", texts["mississippi"], 0), ("r27_65536", texts["r27_65536"], 0),
       ("d1_300k", texts["d1_300k"], 0), ("d2_300k", texts["d2_300k"], 0), ("fib_46k", texts["fib"], 0), ("repeat_block_60k", texts["repeat_block"], 0)]
for mode in ("1", "0"):
    os.environ["SA_HIP_LOCAL_ROUNDS"] = mode
    print("SA_HIP_LOCAL_ROUNDS=" + mode)
    for name, t, L in sel:
        with _capi.DeviceIndex(max(t.size, 1), 0) as idx:
            idx.build(t, L)
            best_dev, best_wall = 1e9, 1e9
            for _ in range(5):
                t0 = time.perf_counter(); idx.build(t, L); w = (time.perf_counter() - t0) * 1e3
                st = idx.build_stats()
                best_dev = min(best_dev, st["total_ms"]); best_wall = min(best_wall, w)
            print("  %-18s n=%7d L=%2d rounds %2d: device %.3f ms, host call %.3f ms" % (name, t.size, L, st["rounds"], best_dev, best_wall), flush=True)
