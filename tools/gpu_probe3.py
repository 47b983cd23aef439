import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi
rng = np.random.default_rng(1)
blk = rng.integers(97, 123, 1 << 20, dtype=np.uint8)
t = np.tile(blk, 64)
with _capi.DeviceIndex(t.size, 0) as idx:
    try:
        idx.build(t)
        print("ok verify", idx.verify())
    except Exception as e:
        print("FAIL", e)
