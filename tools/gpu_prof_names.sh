#!/bin/bash
# rocprofv3 kernel stats of the config-5-shaped build (company names, L = 32 then full): tools/gpu_prof_names.sh [rows]
rows=${1:-25000000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_names -- python3 tools/gpu_profile_names.py $rows > gpurun_out/prof_names.log 2>&1
rc=$?
grep "^names" gpurun_out/prof_names.log
python - <<PY
import csv, glob
f = glob.glob('gpurun_out/prof_names/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), ('%.3f' % (float(r['AverageNs'])/1e6)).rjust(8), ('%.1f' % (float(r['TotalDurationNs'])/1e6)).rjust(8), r['Percentage'])
PY
exit $rc
