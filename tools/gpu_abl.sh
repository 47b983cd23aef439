#!/bin/bash
# A/B of alternative builds of the library (tools/abl/*.so, hipcc -D...) on one box: pass times per variant
export TMPDIR=/tmp SA_HIP_DIAG=1
for r in 1 2; do for m in ${1:-product ti12 ti20 ti24}; do
  lib=$GRAFT_REPO_ROOT/tools/abl/libsa_hip_$m.so; [ $m = product ] && lib=$GRAFT_REPO_ROOT/suffixarray_amd/libsa_hip.so
  SA_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$m build_ms %.3f' % d['build_ms'], {k: round(v['avg_launch_ms'],3) for k, v in d['sort_passes']['by_kernel'].items()}, d['gate'].get('verify_violations'))
" || echo "$m FAILED"
done; done
