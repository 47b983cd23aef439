"""Sweep of the group finisher's knobs (one box): python3 tools/gpu_fin_sweep.py"""
import os, subprocess, sys
os.environ.setdefault("SA_HIP_DIAG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = sys.argv[1] if len(sys.argv) > 1 else "100000000"
for pilot in ("0", "1"):
    for cm in ("0", "48", "96", "160"):
        for rc in ("0", "2", "3", "4"):
            e = dict(os.environ); e.update(SA_HIP_PILOT=pilot, SA_HIP_FIN_COUNT_MAX=cm, SA_HIP_FIN_RADIX_CHARS=rc)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_profile_text.py"), "words", n, "0", "4"], env=e, capture_output=True, text=True, timeout=600)
            print("pilot=%s count_max=%-3s radix_chars=%s  %s" % (pilot, cm, rc, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1][:120]), flush=True)
