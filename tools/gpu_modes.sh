#!/bin/bash
# process-to-process spread of the claim-form passes, by setting (diagnostic; one box): tools/gpu_modes.sh "SET1 SET2 ..." runs
export TMPDIR=/tmp SA_HIP_DIAG=1 SA_HIP_DEBUG_ADDR=1
sets=${1:-"SA_HIP_CURSOR_PAD=1 SA_HIP_CURSOR_PAD=0"}; runs=${2:-5}
for r in $(seq $runs); do for st in $sets; do
  env $(echo $st | tr "," " ") timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/tmp/modes.err | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$st build_ms %.3f' % d['build_ms'], {k.split('<')[0]: round(v['avg_launch_ms'],3) for k, v in d['sort_passes']['by_kernel'].items()}, d['gate'].get('verify_violations'))
" || echo "$st FAILED"
  grep "atomic probe" /tmp/modes.err | sort | uniq -c | head -3
done; done
