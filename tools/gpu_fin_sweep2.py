"""Tile capacity / workgroup size of the group finisher (libraries built with -DSA_FIN_CAP / -DSA_FIN_ITEMS): python3 tools/gpu_fin_sweep2.py"""
import os, subprocess, sys, glob
os.environ.setdefault("SA_HIP_DIAG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [("default", os.path.join(ROOT, "suffixarray_amd", "libsa_hip.so"))] + [(os.path.basename(p), p) for p in sorted(glob.glob(os.path.join(ROOT, "tools", "libsa_fin_*.so")))]
for kind, L in (("words", "0"), ("names", "32")):
    for pilot in ("0", "1"):
        for name, lib in libs:
            e = dict(os.environ); e.update(SA_HIP_PILOT=pilot, SA_HIP_LIB=lib, SA_HIP_FIN_COUNT_MAX="96")
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_profile_text.py"), kind, "100000000", L, "4"], env=e, capture_output=True, text=True, timeout=600)
            print("%-5s pilot=%s %-24s %s" % (kind, pilot, name, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1][:150]), flush=True)
