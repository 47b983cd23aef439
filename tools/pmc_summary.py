"""Summarise two rocprofv3 --pmc runs of bench.py (FETCH_SIZE and WRITE_SIZE, separate passes) into
profiles/pmc_onesweep.json + the per-dispatch CSVs kept beside it.

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <tag> [kernel name substring]

Corrections (profiles/r01_c_pmc_calibration.md): counters are in KiB; on gfx950 FETCH_SIZE counts 64 B
per 128-B request, i.e. half of the bytes (re-calibrated with 8- and 4-byte-per-lane copy kernels);
WRITE_SIZE is exact."""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = sys.argv[4] if len(sys.argv) > 4 else "seg_onesweep_kernel<512, 24, false, true>"   # substring of the rocprofv3 kernel name


def rows(path, counter):
    out = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and KERNEL in r["Kernel_Name"]:
            out.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"])))
    return sorted(out)


def main():
    fetch_csv, write_csv, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    f = rows(fetch_csv, "FETCH_SIZE")
    w = rows(write_csv, "WRITE_SIZE")
    assert f and len(f) == len(w), (len(f), len(w))
    big = max(g for _, _, g, _ in f)
    fb = [v * 1024.0 * 2.0 for _, _, g, v in f if g == big]
    wb = [v * 1024.0 for _, _, g, v in w if g == big]
    n = len(fb)
    for name, data, cname in (("fetch", f, "FETCH_SIZE"), ("write", w, "WRITE_SIZE")):
        with open(os.path.join(ROOT, "profiles", "%s_pmc_%s_onesweep_n1e9.csv" % (tag, name)), "w") as o:
            o.write("Dispatch_Id,Kernel_Name,Grid_Size,Counter_Name,Counter_Value_KB\n")
            for d, k, g, v in data:
                o.write("%d,%s,%d,%s,%f\n" % (d, k.split("(")[0].replace("void sa::", ""), g, cname, v))
    j = {
        "kernel": KERNEL,
        "workload": "bench.py default: D1 uniform27 N=1e9, k0=8 (40-bit keys): top-digit pass + 3 passes of this kernel + last pass per build",
        "launches": n,
        "fetch_bytes_total": sum(fb),
        "write_bytes_total": sum(wb),
        "traffic_bytes_per_launch": (sum(fb) + sum(wb)) / n,
        "algorithmic_bytes_per_launch": 16000000000.0 if "seg_onesweep_kernel<512, 24, false, true>" in KERNEL else None,
        "corrections": "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request; calibrated with 8-byte and 4-byte-per-lane copy "
                       "kernels in tools/sortbench.hip), WRITE_SIZE exact, counters in KiB",
        "commands": [
            "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline",
            "rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline",
        ],
        "source": [os.path.basename(fetch_csv), os.path.basename(write_csv), tag],
    }
    json.dump(j, open(os.path.join(ROOT, "profiles", "pmc_onesweep.json"), "w"), indent=1)
    print(json.dumps(j, indent=1))


if __name__ == "__main__":
    main()
