"""FIN_COUNT_MAX after the counting ranks were unrolled (one box): python3 tools/gpu_fin_sweep5.py"""
import os, subprocess, sys
os.environ.setdefault("SA_HIP_DIAG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for kind, n, L in (("names", "400000000", "32"), ("words", "100000000", "0")):
    for cm in ("96", "128", "192", "256", "384", "96"):
        e = dict(os.environ); e.update(SA_HIP_FIN_COUNT_MAX=cm)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_profile_text.py"), kind, n, L, "3"], env=e, capture_output=True, text=True, timeout=600)
        print("count_max=%-4s %s" % (cm, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1][:110]), flush=True)
