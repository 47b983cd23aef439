"""A/B of the in-LDS group finisher and of the initial key length on refinement-heavy texts (one box, one process):
    python3 tools/gpu_finisher_ab.py [n]"""
import os
os.environ.setdefault("SA_HIP_DIAG", "1")
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = sys.argv[1] if len(sys.argv) > 1 else "100000000"
cases = [("words", "0"), ("names", "32"), ("names", "0"), ("d1", "0"), ("repeat", "0")]
variants = [("finish=1", {"SA_HIP_GROUP_FINISH": "1"}), ("finish=1 nopilot", {"SA_HIP_GROUP_FINISH": "1", "SA_HIP_PILOT": "0"}),
            ("finish=0", {"SA_HIP_GROUP_FINISH": "0"}), ("finish=0 nopilot", {"SA_HIP_GROUP_FINISH": "0", "SA_HIP_PILOT": "0"})]
for kind, L in cases:
    for label, env in variants:
        if kind in ("d1", "repeat") and "nopilot" in label:
            continue
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_profile_text.py"), kind, n if kind != "repeat" else "50000000", L, "4"],
                           env=e, capture_output=True, text=True, timeout=600)
        print("%-18s %s" % (label, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
