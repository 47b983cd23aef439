#!/bin/bash
# Instruction / wait counters of the group finisher (v1 and v2) on name text: tools/gpu_fin_pmc.sh <tag> [n]
tag=$1; n=${2:-400000000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SA_HIP_DIAG=1
for v in 1 0; do
  export SA_HIP_FIN_V2=$v
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_fin${v}_a -- python3 tools/gpu_profile_text.py names $n 32 2 > gpurun_out/pmc_${tag}_fin${v}_a.log 2>&1 || echo "pass a failed"
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_fin${v}_b -- python3 tools/gpu_profile_text.py names $n 32 2 > gpurun_out/pmc_${tag}_fin${v}_b.log 2>&1 || echo "pass b failed"
  timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_fin${v}_c -- python3 tools/gpu_profile_text.py names $n 32 2 > gpurun_out/pmc_${tag}_fin${v}_c.log 2>&1 || echo "pass c failed"
done
python3 - <<PY
import csv, glob, collections
for v in (1, 0):
    for ps in "abc":
        fs = glob.glob("gpurun_out/pmc_${tag}_fin%d_%s/**/*counter_collection.csv" % (v, ps), recursive=True)
        if not fs:
            print("no csv", v, ps); continue
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(fs[0])):
            k = r["Kernel_Name"]
            if "group_finish" not in k and "seg48_onesweep_kernel<512, 14, false, false>" not in k: continue
            k = k.split("(")[0][-45:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            if r["Counter_Name"] in ("SQ_WAVES", "SQ_INSTS_SALU", "TCC_REQ_sum"): cnt[k] += 1
        for k in acc:
            print("FIN_V2=%d %s launches %d: %s" % (v, k, cnt[k], {c: "%.4g" % x for c, x in acc[k].items()}))
PY
