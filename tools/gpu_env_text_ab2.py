"""A/B of one diagnostic switch on refinement-heavy text, one box, one process (the texts are generated once):
    python3 tools/gpu_env_text_ab2.py VAR v1,v2 [names_chars] [words_chars] [rounds] [compare]
Per value: a fresh handle, one upload build, three timed device builds; the first round also verifies on the device and
(compare = 1) checks that every value leaves the same suffix array; SA_HIP_DEBUG_ROUNDS=1 once per value (stderr)."""
import os, sys, time
os.environ.setdefault("SA_HIP_DIAG", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suffixarray_amd import _capi, synth  # noqa: E402

var = sys.argv[1]
vals = sys.argv[2].split(",")
n_names = int(sys.argv[3]) if len(sys.argv) > 3 else 400_000_000
n_words = int(sys.argv[4]) if len(sys.argv) > 4 else 100_000_000
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 2
compare = int(sys.argv[6]) if len(sys.argv) > 6 else 1
texts = []
if n_names:
    rows = n_names // 18
    path = "/tmp/companies_%d.csv" % rows
    t0 = time.time()
    _capi.synth_csv(path, rows, 1)
    names_text = np.array(_capi.csv_extract_column(path, "company_name", copy=False)[1])
    os.remove(path)
    texts.append(("names", names_text, 32))
    print("names text: %d chars (%.1f s)" % (names_text.size, time.time() - t0), flush=True)
if n_words:
    texts.append(("words", synth.d2_words(n_words), 0))
bad = 0
for kind, t, L in texts:
    ref = None
    for r in range(rounds):
        for v in vals:
            os.environ[var] = v
            if r == 0:
                os.environ["SA_HIP_DEBUG_ROUNDS"] = "1"
                with _capi.DeviceIndex(t.size, 0) as idx:
                    idx.build(t, L)
                    viol = idx.verify()
                    if compare:
                        sa = idx.sa_u32().copy()
                        if ref is None:
                            ref = sa
                        elif not np.array_equal(ref, sa):
                            bad += 1
                            print("MISMATCH %s %s=%s: %d slots differ" % (kind, var, v, int((ref != sa).sum())), flush=True)
                        del sa
                    if viol:
                        bad += 1
                os.environ.pop("SA_HIP_DEBUG_ROUNDS", None)
            else:
                viol = -1
            with _capi.DeviceIndex(t.size, 0) as idx:
                idx.build(t, L)
                ms = []
                for _ in range(3):
                    idx.build_device(idx.text_dev, t.size, L)
                    ms.append(idx.build_stats()["total_ms"])
                st = idx.build_stats()
            print("%s n=%d L=%d %s=%s: best %.2f ms (%.2f Gchars/s) k0=%d rounds %d active_total %d finisher runs %d looked at %d resolved %d verify=%d" % (
                kind, t.size, L, var, v, min(ms), t.size / min(ms) / 1e6, st["initial_chars"], st["rounds"], st["active_total"], st["finisher_runs"],
                st["finisher_records"], st["finisher_resolved"], viol), flush=True)
print("FAILED %d" % bad if bad else "ALL OK")
sys.exit(1 if bad else 0)
