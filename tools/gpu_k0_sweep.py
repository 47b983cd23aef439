"""Initial key length on refinement-heavy text at config-5 scale (one box): python3 tools/gpu_k0_sweep.py [kind] [n] [L]"""
import os, subprocess, sys
os.environ.setdefault("SA_HIP_DIAG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kind = sys.argv[1] if len(sys.argv) > 1 else "names"
n = sys.argv[2] if len(sys.argv) > 2 else "900000000"
L = sys.argv[3] if len(sys.argv) > 3 else "32"
for k0 in ("8", "9", "10", "11", "12"):
    e = dict(os.environ); e.update(SA_HIP_INITIAL_CHARS=k0)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_profile_text.py"), kind, n, L, "3"], env=e, capture_output=True, text=True, timeout=900)
    print("k0=%-2s %s" % (k0, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1][:170]), flush=True)
