"""Pass 0 of the 12-byte-record sort from the text vs from a key array (SA_HIP_WIDE_TEXT_PASS): python3 tools/gpu_widetext_ab.py"""
import os, subprocess, sys
os.environ.setdefault("SA_HIP_DIAG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for kind, n, L in (("words", "100000000", "0"), ("names", "100000000", "32"), ("names", "900000000", "32")):
    for mode in ("1", "0", "1", "0"):
        e = dict(os.environ); e.update(SA_HIP_WIDE_TEXT_PASS=mode)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_profile_text.py"), kind, n, L, "4"], env=e, capture_output=True, text=True, timeout=900)
        print("wide_text_pass=%s %s" % (mode, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1][:150]), flush=True)
