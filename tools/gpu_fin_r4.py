"""Round 4: the finisher's counting rounds on the block-striped layout x FIN_COUNT_MAX, names (config-5 column) and words, one box, one process
(the texts are generated once).  python3 tools/gpu_fin_r4.py [names_chars] [words_chars] [settings: striped:count_max,...]"""
import os, sys, time
os.environ.setdefault("SA_HIP_DIAG", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi, synth  # noqa: E402

n_names = int(sys.argv[1]) if len(sys.argv) > 1 else 916_000_000
n_words = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
settings = [tuple(x.split(":")) for x in (sys.argv[3] if len(sys.argv) > 3 else "0:96,1:96,1:256,1:512,1:1024,0:96,1:512").split(",")]
extra = dict(kv.split("=") for kv in sys.argv[4].split(",")) if len(sys.argv) > 4 else {}
texts = []
if n_names:
    rows = n_names // 18
    path = "/tmp/companies_%d.csv" % rows
    t0 = time.time()
    _capi.synth_csv(path, rows, 1)
    names_text = np.array(_capi.csv_extract_column(path, "company_name", copy=False)[1])
    os.remove(path)
    texts.append(("names", names_text, 32))
    print("names text: %d chars (%.1f s)" % (texts[-1][1].size, time.time() - t0), flush=True)
if n_words:
    texts.append(("words", synth.d2_words(n_words), 0))
for kind, t, L in texts:
    first = True
    for striped, cm in settings:
        os.environ.update(SA_HIP_FIN_STRIPED=striped, SA_HIP_FIN_COUNT_MAX=cm, **extra)
        if first:
            os.environ["SA_HIP_DEBUG_ROUNDS"] = "1"
        with _capi.DeviceIndex(t.size, 0) as idx:
            idx.build(t, L)
            os.environ.pop("SA_HIP_DEBUG_ROUNDS", None)
            ms = []
            if first:   # the debug build is slow: fresh handle for timing
                pass
            for _ in range(3):
                idx.build_device(idx.text_dev, t.size, L)
                ms.append(idx.build_stats()["total_ms"])
            st = idx.build_stats()
            bad = idx.verify() if first else -1
        if first:
            # timing on a handle that was created without the debug switch
            with _capi.DeviceIndex(t.size, 0) as idx:
                idx.build(t, L)
                ms = []
                for _ in range(3):
                    idx.build_device(idx.text_dev, t.size, L)
                    ms.append(idx.build_stats()["total_ms"])
                st = idx.build_stats()
        first = False
        print("%s n=%d L=%d striped=%s count_max=%-5s %s: best %.2f ms (%.2f Gchars/s) k0=%d rounds %d active_total %d finisher_resolved %d verify=%d" % (
            kind, t.size, L, striped, cm, extra, min(ms), t.size / min(ms) / 1e6, st["initial_chars"], st["rounds"], st["active_total"],
            st["finisher_resolved"], bad), flush=True)
