"""Alternate builds of the library on refinement-heavy text (one box): python3 tools/gpu_lib_ab.py lib1.so,lib2.so,... [kind n L] [rounds]"""
import os, subprocess, sys
os.environ.setdefault("SA_HIP_DIAG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1].split(",")
kind, n, L = (sys.argv[2:5] + ["names", "400000000", "32"][len(sys.argv[2:5]):])
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 2
for _ in range(rounds):
    for lib in libs:
        e = dict(os.environ); e.update(SA_HIP_LIB=os.path.join(ROOT, lib))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_profile_text.py"), kind, n, L, "3"], env=e, capture_output=True, text=True, timeout=900)
        print("%-40s %s" % (lib, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1][:130]), flush=True)
