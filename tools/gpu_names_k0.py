"""Initial key length on word-like text: build time of the company-name column (config-5 shape) and of D2 words for
SA_HIP_INITIAL_CHARS = default (pilot: 12), 11, 10 -- 60 bits take 8 passes of 12-byte records, 55 and 50 bits take 7."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from suffixarray_amd import _capi, synth
from suffixarray_amd.csv_ingest import extract_column
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
path = "/tmp/companies_%d.csv" % rows
_capi.synth_csv(path, rows, 1)
names = extract_column(path, "company_name").text_array.copy()
os.remove(path)
words = synth.d2_words(200_000_000)
for label, t, L in (("names L=32", names, 32), ("names full", names, 0), ("words full", words, 0)):
    for k0 in ("", "11", "10"):
        if k0: os.environ["SA_HIP_INITIAL_CHARS"] = k0
        else: os.environ.pop("SA_HIP_INITIAL_CHARS", None)
        with _capi.DeviceIndex(t.size, 0) as idx:
            ms = []
            for _ in range(3):
                idx.build(t, L)
                ms.append(idx.build_stats()["total_ms"])
            st = idx.build_stats()
            print("%-11s n=%d k0=%2d passes %2d rounds %d active_total %d: %.2f ms (best of 3; %.2f Gchars/s) verify=%d" % (
                label, t.size, st["initial_chars"], st["radix_passes"], st["rounds"], st["active_total"], min(ms), t.size / min(ms) / 1e6, idx.verify()), flush=True)
