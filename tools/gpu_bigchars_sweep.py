"""Characters per global round while the finisher is at work (SA_HIP_BIG_ROUND_CHARS): python3 tools/gpu_bigchars_sweep.py"""
import os, subprocess, sys
os.environ.setdefault("SA_HIP_DIAG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for kind, n, L in (("names", "900000000", "32"), ("words", "100000000", "0"), ("names", "100000000", "0")):
    for pilot in ("1", "0"):
        for bc in ("0", "3", "4", "5", "6", "7"):
            e = dict(os.environ); e.update(SA_HIP_BIG_ROUND_CHARS=bc, SA_HIP_PILOT=pilot)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_profile_text.py"), kind, n, L, "3"], env=e, capture_output=True, text=True, timeout=900)
            print("pilot=%s big_round_chars=%s %s" % (pilot, bc, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1][:175]), flush=True)
