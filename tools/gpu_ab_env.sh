#!/bin/bash
# A/B of one diagnostic switch on one box: tools/gpu_ab_env.sh VAR v1,v2,... [steps] [rounds]
# alternates bench runs (no CPU baseline) with VAR set to each value and prints build / query / per-kernel pass times
var=$1; vals=$(echo ${2:-"1,0"} | tr "," " "); steps=${3:-5}; rounds=${4:-2}
export TMPDIR=/tmp SA_HIP_DIAG=1
for r in $(seq $rounds); do for m in $vals; do
  env $var=$m timeout -k 10 200 python bench.py --steps $steps --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$var=$m build_ms %.3f query_ms %.4f' % (d['build_ms'], d['query_ms']), {k: round(v['avg_launch_ms'],3) for k, v in d['sort_passes']['by_kernel'].items()})
" || exit 1
done; done
