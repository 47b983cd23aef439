import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi, synth
from suffixarray_amd.csv_ingest import extract_column
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
p = "/tmp/c_%d.csv" % rows
if not os.path.exists(p): _capi.synth_csv(p, rows, 1)
names = np.frombuffer(extract_column(p, "company_name").text, np.uint8)
cases = [("names", names, 0), ("names_L32", names, 32), ("d2_words", synth.d2_words(names.size), 0)]
for name, t, L in cases:
    with _capi.DeviceIndex(t.size, 0) as idx:
        idx.build(t, L); idx.build(t, L)
        st = idx.build_stats()
        print("%-10s n=%d L=%d total %8.2f ms radix %8.2f ms passes %3d k0=%d rounds %d (chunk %d dbl %d) active_total %d verify=%d" % (
            name, t.size, L, st["total_ms"], st["radix_ms"], st["radix_passes"], st["initial_chars"], st["rounds"], st["chunk_rounds"], st["doubling_rounds"], st["active_total"], idx.verify()), flush=True)
