"""Kernel-level view of a refinement-heavy build on one text distribution.
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_words --output-format csv -- python3 tools/gpu_profile_text.py words 100000000 [L]
kinds: words (D2), names (company_name column of the synthetic config-5 CSV), d1, repeat (1 MiB block repeated)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi, synth  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "words"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 0
builds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
if kind == "words":
    t = synth.d2_words(n)
elif kind == "d1":
    t = synth.d1_uniform27(n)
elif kind == "repeat":
    rng = np.random.default_rng(1)
    t = np.tile(rng.integers(97, 123, 1 << 20, dtype=np.uint8), n // (1 << 20) + 1)[:n].copy()
elif kind == "names":
    rows = n // 18   # ~18.3 column characters per row
    path = "/tmp/companies_%d.csv" % rows
    _capi.synth_csv(path, rows, 1)
    t = np.array(_capi.csv_extract_column(path, "company_name", copy=False)[1])
    os.remove(path)
else:
    raise SystemExit("unknown kind " + kind)
with _capi.DeviceIndex(t.size, 0) as idx:
    idx.build(t, L)
    ms = []
    for _ in range(builds - 1):
        idx.build_device(idx.text_dev, t.size, L)
        ms.append(idx.build_stats()["total_ms"])
    st = idx.build_stats()
    print("%s n=%d L=%d builds=%d: best %.2f ms median %.2f ms (%.2f Gchars/s) radix %.2f ms passes %d k0=%d rounds %d (chunk %d dbl %d) active_total %d tiny %d verify=%d split rb=%d max=%d flags=%d" % (
        kind, t.size, L, builds, min(ms), sorted(ms)[len(ms) // 2], t.size / min(ms) / 1e6, st["radix_ms"], st["radix_passes"], st["initial_chars"], st["rounds"],
        st["chunk_rounds"], st["doubling_rounds"], st["active_total"], st.get("tiny_resolved", -1), idx.verify(), st["split_plan"], st["split_max"], st["lite_flags"]), flush=True)
