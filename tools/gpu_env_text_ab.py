"""One diagnostic switch on refinement-heavy text (one box): python3 tools/gpu_env_text_ab.py VAR v1,v2 [kind n L] [rounds]"""
import os, subprocess, sys
os.environ.setdefault("SA_HIP_DIAG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
var, vals = sys.argv[1], sys.argv[2].split(",")
kind, n, L = (sys.argv[3:6] + ["words", "100000000", "0"][len(sys.argv[3:6]):])
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 3
for _ in range(rounds):
    for v in vals:
        e = dict(os.environ); e[var] = v
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_profile_text.py"), kind, n, L, "3"], env=e, capture_output=True, text=True, timeout=900)
        print("%s=%-3s %s" % (var, v, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1][:120]), flush=True)
