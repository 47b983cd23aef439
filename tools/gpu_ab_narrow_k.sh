#!/bin/bash
# A/B on one box: narrow K on / off, bench (steps 5) + rocprofv3 kernel stats of the default
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for m in 1 0 1 0; do
  SA_HIP_NARROW_K=$m timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('NARROW_K=$m build_ms', d['build_ms'], 'query_ms', d['query_ms'], {k: round(v['avg_launch_ms'],3) for k, v in d['sort_passes']['by_kernel'].items()})
" || exit 1
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nk -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_nk.log 2>&1
