#!/bin/bash
# per-kernel times of the headline build for several builds of the library (one box): tools/gpu_lib_kernels.sh lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  export SA_HIP_LIB=$GRAFT_REPO_ROOT/$lib
  tag=$(basename $lib .so)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lk_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $BENCH_ARGS > gpurun_out/lk_$tag.log 2>&1 || exit 1
  python3 - "$tag" <<'P'
import csv, glob, sys
f = glob.glob('gpurun_out/prof_lk_%s/*/*_kernel_stats.csv' % sys.argv[1])[0]
out = []
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('flags_kernel', 'compact_kernel', 'seg_hist', 'top_hist', 'byte_hist', 'text_top', 'seg_onesweep', 'query_kernel')):
        out.append('%s %.3f' % (r['Name'].split('(')[0].replace('void sa::', '').replace('sa::', '')[:34], float(r['AverageNs']) / 1e6))
print(sys.argv[1], '|', ' | '.join(out))
P
done
