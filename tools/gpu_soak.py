"""Soak: repeated builds with on-device verification (rare-race hunting).  The D1 builds are 64-bit builds
(sa_hip_index_build_device64: int64 copy out of the sort's last pass + patch) and the int64 array is compared with the
u32 one on the device after every build.  python3 tools/gpu_soak.py [reps of the N = 1e9 case]"""
import sys, os, time
import numpy as np
import torch   # before libsa_hip.so: the process must end up with ONE HIP runtime (torch's), whichever library is loaded first decides
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi, synth
rng = np.random.default_rng(1)
D1_REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 25
cases = [("d1_1e9", synth.d1_uniform27(1_000_000_000), 0, D1_REPS),
         ("repeat_2^26", np.tile(rng.integers(97, 123, 1 << 20, dtype=np.uint8), 64), 0, 8),
         ("d2_2e8", synth.d2_words(200_000_000), 0, 8),
         ("d2_2e8_L32", None, 32, 8)]
prev = None
for name, t, L, reps in cases:
    if t is None: t = prev
    prev = t
    bad_total = 0
    with _capi.DeviceIndex(t.size, 0) as idx:
        t0 = time.time()
        idx.build(t, L)
        sa64 = torch.empty(t.size, dtype=torch.int64, device="cuda:0")
        from suffixarray_amd.distributed import device_view
        for r in range(reps):
            idx.build_device64(idx.text_dev, t.size, sa64.data_ptr(), L)
            idx.sync()
            v = idx.verify()
            sa32 = device_view(idx.sa_dev, t.size, torch.int32, torch.device("cuda:0"))
            # (values >= 2^31 do not occur at these sizes: the int32 view equals the u32 array)
            mism = 0
            for o in range(0, t.size, 1 << 28):
                mism += int((sa64[o:o + (1 << 28)] != sa32[o:o + (1 << 28)].to(torch.int64)).sum().item())
            v += mism
            bad_total += v
            if v: print("  VIOLATIONS", name, "rep", r, v, idx.build_stats(), flush=True)
        print("%-12s n=%d L=%d reps=%d violations=%d (%.1fs) last build %.1f ms" % (name, t.size, L, reps, bad_total, time.time() - t0, idx.build_stats()["total_ms"]), flush=True)
