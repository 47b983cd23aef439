"""Summarise the rocprofv3 --pmc runs of tools/gpu_profiles.sh <tag> (FETCH_SIZE, WRITE_SIZE and the raw L2
memory-side request counters, separate passes over `bench.py --steps 1 --warmup 0`) into

    profiles/pmc_onesweep.json            dominant sort kernel: HBM bytes per launch
    profiles/pmc_query.json               query_kernel: HBM bytes + 64-byte requests per 1M-query launch
    profiles/<tag>_pmc_{fetch,write,raw}.csv   the per-dispatch counters of the kernels of a build + the query batch
    profiles/<tag>_pmc_calibration.md     FETCH_SIZE on random 4-/8-byte reads (tools/gatherbench), if that pass ran

    python tools/pmc_summary.py <tag> [n_chars] [queries]

Corrections (MI355X_MICROARCH.md, HBM section; profiles/r01_c_pmc_calibration.md; <tag>_pmc_calibration.md):
counters are in KiB; FETCH_SIZE = requests x 64 B: a wide coalesced streaming read goes out as 128-byte requests and
is under-counted by exactly 2 (x2 for the streaming sort kernels), a RANDOM narrow read is one 64-byte request and is
counted exactly (x1 for query_kernel; the raw pass shows TCC_EA0_RDREQ_32B = TCC_BUBBLE = 0 and
TCC_EA0_RDREQ x 64 B = FETCH_SIZE for it); WRITE_SIZE is exact."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SORT_KERNEL = "seg_onesweep_kernel<512, 24, false, true>"
KEEP = ("onesweep", "text_top_pass", "query_kernel", "flags_kernel", "seg_hist", "top_hist", "byte_hist", "compact", "widen", "gather_kernel",
        "loc_sort", "chunk_keys", "tiny_groups", "seg_split", "local_finish", "split_hist", "flags_lite", "lite_gather")


def counter_file(tag, what):
    g = glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{what}", "*", "*_counter_collection.csv"))
    return g[0] if g else None


def rows(path):
    out = []
    if not path:
        return out
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void sa::", "").replace("void ", "")
        out.append((int(r["Dispatch_Id"]), name, int(r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"])))
    return sorted(out)


def keep_csv(tag, what, data):
    if not data:
        return
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_{what}.csv"), "w") as o:
        o.write("Dispatch_Id,Kernel_Name,Grid_Size,Counter_Name,Counter_Value\n")
        for d, k, g, c, v in data:
            if any(s in k for s in KEEP):
                o.write("%d,\"%s\",%d,%s,%f\n" % (d, k, g, c, v))


def main():
    tag = sys.argv[1]
    n_chars = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
    queries = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
    fetch, write, raw = rows(counter_file(tag, "fetch")), rows(counter_file(tag, "write")), rows(counter_file(tag, "raw"))
    for what, data in (("fetch", fetch), ("write", write), ("raw", raw)):
        keep_csv(tag, what, data)
    light = "--no-cpu-baseline --no-secondary"
    cmds = [f"rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 {light}",
            f"rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 {light}"]

    # ---- dominant sort kernel ----------------------------------------------------------------------------
    # round 4: with the three-pass plan (radix_split.hpp) the dominant kernel of a D1 build is local_finish_kernel (8 + 8 bytes per
    # record in, 8 + 8 out, + 8 for the int64 copy of bench.py's 64-bit build); the split pass is reported beside it
    sort_kernel, alg_bytes, workload = SORT_KERNEL, 16.0 * n_chars, \
        "bench.py default: D1 uniform27 N=1e9, k0=8 (40-bit keys): top-digit pass + 3 passes of this kernel + last pass per build"
    if any("local_finish_kernel" in k for _, k, g, c, v in fetch):
        sort_kernel, alg_bytes, workload = "local_finish_kernel", 24.0 * n_chars, \
            "bench.py default: D1 uniform27 N=1e9, k0=8 (40-bit keys), three-pass plan: top-digit pass + split pass + ONE launch of this kernel per build (int64 copy fused)"
    f = [(g, v) for _, k, g, c, v in fetch if c == "FETCH_SIZE" and sort_kernel in k]
    w = [(g, v) for _, k, g, c, v in write if c == "WRITE_SIZE" and sort_kernel in k]
    if f and len(f) == len(w):
        big = max(g for g, _ in f)
        fb = [v * 1024.0 * 2.0 for g, v in f if g == big]
        wb = [v * 1024.0 for g, v in w if g == big]
        if sort_kernel == "local_finish_kernel" and len(wb) > 1 and max(wb) > 1.2 * min(wb):
            # the process runs one u32 build (no int64 copy: 8 bytes less written per record) before the timed 64-bit one:
            # the line's figure belongs to the 64-bit launch, the one with the larger write count
            i = max(range(len(wb)), key=lambda k: wb[k])
            fb, wb = [fb[i]], [wb[i]]
        j = {"kernel": sort_kernel, "n_chars": n_chars,
             "workload": workload,
             "launches": len(fb), "fetch_bytes_total": sum(fb), "write_bytes_total": sum(wb),
             "traffic_bytes_per_launch": (sum(fb) + sum(wb)) / len(fb), "algorithmic_bytes_per_launch": alg_bytes,
             "corrections": "FETCH_SIZE x2 (wide coalesced reads leave the L2 as 128-byte requests tallied at 64 B; calibrated with 8- and "
                            "4-byte-per-lane copy kernels, tools/sortbench.hip), WRITE_SIZE exact, counters in KiB",
             "commands": cmds, "source": tag}
        sf = [v * 1024.0 * 2.0 for _, k, g, c, v in fetch if c == "FETCH_SIZE" and "seg_split_kernel" in k]
        sw = [v * 1024.0 for _, k, g, c, v in write if c == "WRITE_SIZE" and "seg_split_kernel" in k]
        if sf and len(sf) == len(sw):
            j["split_pass"] = {"kernel": "seg_split_kernel<512, 28>", "launches": len(sf), "traffic_bytes_per_launch": (sum(sf) + sum(sw)) / len(sf),
                               "fetch_bytes_per_launch": sum(sf) / len(sf), "write_bytes_per_launch": sum(sw) / len(sw),
                               "algorithmic_bytes_per_launch": 16.0 * n_chars}
        json.dump(j, open(os.path.join(ROOT, "profiles", "pmc_onesweep.json"), "w"), indent=1)
        print(json.dumps(j, indent=1))

    # ---- query kernel: one launch per bench step -------------------------------------------------------------------
    # launches of query_kernel in `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary`, in order: [0] the timed batch
    # (what the line's rate belongs to), [1] the 8Q classification batch of query_populations, [2..7] six launches of the
    # hits-only batch, [8..13] six of the misses-only batch (round 4; before that the batch was the only / last launch)
    qf = [v for _, k, g, c, v in fetch if c == "FETCH_SIZE" and "query_kernel" in k]
    qw = [v for _, k, g, c, v in write if c == "WRITE_SIZE" and "query_kernel" in k]
    qname = next((k for _, k, g, c, v in fetch if "query_kernel" in k), None)
    main_i = 0 if len(qf) >= 14 else -1
    if qf and qw:
        fb, wb = qf[main_i] * 1024.0, qw[main_i] * 1024.0
        j = {"kernel": qname, "n_chars": n_chars, "queries": queries,
             "workload": f"bench.py default: one batch of {queries:,} 16-byte patterns over the N={n_chars:,} index (50 % text windows, 50 % random)",
             "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "traffic_bytes_per_launch": fb + wb,
             "bytes_per_query": (fb + wb) / queries,
             "corrections": "FETCH_SIZE x1: a random narrow read is ONE 64-byte request and is tallied exactly (tools/gatherbench: 16.8 M random "
                            "4-byte reads of a 4 GiB table = 16.76 M requests = 64.0 B per read by FETCH_SIZE; no 32-byte, no 128-byte requests); "
                            "WRITE_SIZE exact; counters in KiB",
             "commands": cmds, "source": tag}
        rq = [v for _, k, g, c, v in raw if "query_kernel" in k and c == "TCC_EA0_RDREQ_sum"]
        r32 = [v for _, k, g, c, v in raw if "query_kernel" in k and c == "TCC_EA0_RDREQ_32B_sum"]
        if rq:
            j["raw_requests"] = {"TCC_EA0_RDREQ_sum": rq[main_i], "TCC_EA0_RDREQ_32B_sum": r32[main_i] if r32 else None}
            j["read_requests_per_query"] = rq[main_i] / queries
            if len(rq) >= 14:   # the two sub-populations of the batch on their own (bench.py: query_populations)
                j["populations"] = {
                    "hits_only": {"read_requests_per_query": sum(rq[2:8]) / 6 / queries, "fetch_bytes_per_query": sum(qf[2:8]) / 6 * 1024.0 / queries,
                                  "what": "directory (2 adjacent entries) + key window(s) + SA[lo] + the text at it, pattern longer than the key"},
                    "misses_only": {"read_requests_per_query": sum(rq[8:14]) / 6 / queries, "fetch_bytes_per_query": sum(qf[8:14]) / 6 * 1024.0 / queries,
                                    "what": "directory + key window(s); most patterns die in the key array"}}
        # the request-rate ceiling of the memory system for random reads: a CURVE since round 3 (tools/gatherbench sweep ->
        # profiles/r03_gather_sweep.log: footprint x request size x resident lanes).  The figure the bench line compares with
        # is the rate of random 4-byte reads over a 4 GiB footprint (the K and SA arrays are 4 GB each at N = 1e9).
        ceiling, curve = None, {}
        sweep = os.path.join(ROOT, "profiles", "r03_gather_sweep.log")
        if os.path.exists(sweep):
            for line in open(sweep):
                p_ = line.split()
                if len(p_) == 6 and p_[0].isdigit():
                    key = f"{p_[1]}B_requests_at_{int(p_[0])}MiB"
                    curve[key] = max(curve.get(key, 0.0), float(p_[4]) * 1e9)
            ceiling = curve.get("4B_requests_at_4096MiB")
        if ceiling is None:
            try:
                ceiling = json.load(open(os.path.join(ROOT, "profiles", "pmc_query.json"))).get("random_read_requests_per_s_ceiling")
            except Exception:
                pass
        j["random_read_requests_per_s_ceiling"] = ceiling
        j["random_read_curve_requests_per_s"] = curve
        j["ceiling_note"] = ("tools/gatherbench sweep (profiles/r03_gather_sweep.log; 2^26 random reads, four independent requests in flight per lane, "
                             "best of 3): random reads beyond the Infinity Cache are served at ~1.65 TB/s in 32-byte granules -- 4-byte reads 51 G/s, "
                             "64-byte reads 26 G/s, 128-byte reads 13 G/s at a 4 GiB footprint (56 G/s for every size up to 1 GiB).  Round 2's 37.4 G/s came "
                             "from ONE request per lane over 2^24 lanes, a launch too short to reach the rate.  TCC_EA0_RDREQ counts a 32-byte "
                             "window read as one request like a 64-byte one: the counter did not move (3.05 -> 3.03) when the K windows were "
                             "halved, the kernel time did (0.089 -> 0.080 ms)")
        json.dump(j, open(os.path.join(ROOT, "profiles", "pmc_query.json"), "w"), indent=1)
        print(json.dumps(j, indent=1))

    # ---- instruction counters per kernel of a build ---------------------------------------------------------------------
    valu = rows(counter_file(tag, "valu"))
    if valu:
        per = {}
        for d, k, g, c, v in valu:
            if g < (1 << 22) or not any(s_ in k for s_ in KEEP):   # the launches over the whole input only
                continue
            per.setdefault((d, k), {})[c] = v
        agg = {}
        for (d, k), cs in per.items():
            a = agg.setdefault(k, {"launches": 0})
            a["launches"] += 1
            for c, v in cs.items():
                a[c] = a.get(c, 0.0) + v
        lines = [f"# Instruction counters of a D1 build at N = {n_chars:,} ({tag})", "",
                 "`rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace",
                 "--output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary`; per launch, launches over the whole input.",
                 "SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles (MI355X_MICROARCH.md); busy = active cycles of the unit / (GRBM_GUI_ACTIVE / 8 XCDs x 4 SIMDs x 256 CUs / 8).", "",
                 "| kernel | launches | waves | VALU instr / wave | LDS instr / wave | VALU active (quad-cycles) / wave | LDS active / wave | wave cycles / wave |",
                 "|---|---|---|---|---|---|---|---|"]
        for k, a in sorted(agg.items()):
            w = a.get("SQ_WAVES", 0.0)
            if w <= 0:
                continue
            lines.append("| `%s` | %d | %.0f | %.0f | %.0f | %.0f | %.0f | %.0f |" % (
                k, a["launches"], w / a["launches"], a.get("SQ_INSTS_VALU", 0) / w, a.get("SQ_INSTS_LDS", 0) / w,
                a.get("SQ_ACTIVE_INST_VALU", 0) / w, a.get("SQ_ACTIVE_INST_LDS", 0) / w, a.get("SQ_WAVE_CYCLES", 0) / w))
        open(os.path.join(ROOT, "profiles", f"{tag}_pmc_valu.md"), "w").write("\n".join(lines) + "\n")
        keep_csv(tag, "valu", valu)
        print("\n".join(lines))

    # ---- calibration on random reads ------------------------------------------------------------------------------------
    cf, cr = rows(counter_file(tag, "calib_fetch")), rows(counter_file(tag, "calib_raw"))
    if cf:
        reads = 1 << 24
        lines = [f"# FETCH_SIZE on random narrow reads ({tag})", "",
                 "`rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- tools/gatherbench` and the same with",
                 "`--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum`: 2^24 lanes read one element each at a hashed index of a",
                 "4 GiB table (every read a distinct sector, far beyond the 256 MiB Infinity Cache).", "",
                 "| kernel | counter | per dispatch | per read |", "|---|---|---|---|"]
        for _, k, g, c, v in cf + cr:
            if "gather_kernel" in k:
                per = v * 1024.0 / reads if c == "FETCH_SIZE" else v / reads
                lines.append(f"| `{k}` | {c} | {v:.1f}{' KiB' if c == 'FETCH_SIZE' else ''} | {per:.3f}{' B' if c == 'FETCH_SIZE' else ' requests'} |")
        lines += ["", "A random 4- or 8-byte read leaves the L2 as ONE 64-byte request and FETCH_SIZE tallies exactly those 64 bytes: no x2",
                  "correction for `query_kernel` (the x2 of the streaming kernels comes from their 128-byte requests being tallied at 64).",
                  "Rate of the same kernel without the profiler (`tools/gatherbench`): 35-36 G random reads/s = 2.3 TB/s of 64-byte",
                  "sectors -- the request-rate ceiling of the memory system for this access pattern."]
        open(os.path.join(ROOT, "profiles", f"{tag}_pmc_calibration.md"), "w").write("\n".join(lines) + "\n")
        print("\n".join(lines))


if __name__ == "__main__":
    main()
