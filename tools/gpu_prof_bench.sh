#!/bin/bash
# rocprofv3 kernel stats of the bench (2 timed steps): tools/gpu_prof_bench.sh <tag>  -> gpurun_out/prof_<tag>/
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1
rc=$?
python - <<PY
import csv, glob
f = glob.glob('gpurun_out/prof_$tag/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r['Name'][:64].ljust(64), r['Calls'], round(float(r['AverageNs'])/1e6,3), r['Percentage'])
PY
exit $rc
