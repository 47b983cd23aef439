"""Finisher knobs on the config-5 column (names, L = 32): python3 tools/gpu_fin_sweep3.py [n]"""
import os, subprocess, sys
os.environ.setdefault("SA_HIP_DIAG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = sys.argv[1] if len(sys.argv) > 1 else "400000000"
for cm, rc in (("96", "0"), ("256", "0"), ("512", "0"), ("1024", "0"), ("4096", "0"), ("96", "4"), ("96", "5"), ("256", "5"), ("512", "4"), ("96", "0")):
    e = dict(os.environ); e.update(SA_HIP_FIN_COUNT_MAX=cm, SA_HIP_FIN_RADIX_CHARS=rc)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_profile_text.py"), "names", n, "32", "3"], env=e, capture_output=True, text=True, timeout=600)
    print("count_max=%-4s radix_chars=%s  %s" % (cm, rc, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1][:110]), flush=True)
