import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi, synth
kind = sys.argv[1]; N = int(sys.argv[2]); L = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t = {"d1": synth.d1_uniform27, "d2": synth.d2_words}[kind](N)
with _capi.DeviceIndex(t.size, 0) as idx:
    for _ in range(3):
        idx.build(t, L)
    print(idx.build_stats())
