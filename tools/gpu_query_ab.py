"""1M-query batch at N = 1e9: K windows of 64 / 32 bytes, directory bits (one box): python3 tools/gpu_query_ab.py"""
import os, subprocess, sys, json
os.environ.setdefault("SA_HIP_DIAG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for env in ({"SA_HIP_SECTOR_SEARCH": "2", "SA_HIP_DIR_BITS": "27"}, {"SA_HIP_SECTOR_SEARCH": "2", "SA_HIP_DIR_BITS": "28"}, {"SA_HIP_SECTOR_SEARCH": "1", "SA_HIP_DIR_BITS": "28"},
            {"SA_HIP_SECTOR_SEARCH": "2", "SA_HIP_DIR_BITS": "26"}, {"SA_HIP_SECTOR_SEARCH": "2", "SA_HIP_DIR_BITS": "27"}, {"SA_HIP_SECTOR_SEARCH": "2", "SA_HIP_DIR_BITS": "28"}):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-secondary"], env=e, capture_output=True, text=True, timeout=900)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print(env, "query_ms %.4f (%.2f G q/s) build_ms %.2f gate %s" % (j["query_ms"], j["queries_per_s"] / 1e9, j["build_ms"], j["gate"]), flush=True)
    except Exception as ex:
        print(env, "FAILED", r.stderr[-500:], flush=True)
