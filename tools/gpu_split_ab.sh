#!/bin/bash
# A/B of the three-pass plan's knobs on one box: bench lines (no CPU baseline) per setting, build / pass times
export TMPDIR=/tmp SA_HIP_DIAG=1
run() {
  env "$@" timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$*', 'build_ms %.3f' % d['build_ms'], {k: round(v['avg_launch_ms'],3) for k, v in d['sort_passes']['by_kernel'].items()}, d['gate'].get('verify_violations'))
" || echo "$* FAILED"
}
for r in 1 2; do
  run SA_HIP_SPLIT_FLAGS=1
  run SA_HIP_SPLIT_FLAGS=0
done
for r in; do
  run SA_HIP_SPLIT_ITEMS=24 SA_HIP_LOCAL_BINS=11
  run SA_HIP_SPLIT_ITEMS=28 SA_HIP_LOCAL_BINS=12
  run SA_HIP_SPLIT_ITEMS=32 SA_HIP_LOCAL_BINS=12
  run SA_HIP_SPLIT_ITEMS=24 SA_HIP_LOCAL_BINS=12
done
