"""Per-record time of the narrow sort passes as a function of n: does a ping-pong that fits the 256 MiB Infinity Cache run
faster than one streamed from HBM?  (the question behind cache-blocked passes): python3 tools/gpu_pass_scaling.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi, synth
for n in (8_000_000, 12_000_000, 16_000_000, 24_000_000, 32_000_000, 64_000_000, 128_000_000, 512_000_000, 1_000_000_000):
    t = synth.d1_uniform27(n)
    with _capi.DeviceIndex(n, 0) as idx:
        idx.build(t)
        best = None
        for _ in range(5):
            idx.build_device(idx.text_dev, n, 0)
            st = idx.build_stats()
            cur = [st["pass_ms"][k] / max(st["pass_launches"][k], 1) for k in range(4)]
            best = cur if best is None else [min(a, b) for a, b in zip(best, cur)]
        print("n=%-11d ping-pong %6.0f MB: top pass %.3f ms (%.2f ps/rec)  narrow pass %.3f ms (%.2f ps/rec, %.0f GB/s)  last %.3f ms  build %.2f ms" % (
            n, 16.0 * n / 1e6, best[1], best[1] * 1e9 / n, best[2], best[2] * 1e9 / n, 16.0 * n / best[2] / 1e6, best[3], st["total_ms"]), flush=True)
