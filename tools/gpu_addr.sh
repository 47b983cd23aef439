#!/bin/bash
# do the pass times of a build move with where its buffers lie?  (diagnostic; one box)
export TMPDIR=/tmp SA_HIP_DIAG=1 SA_HIP_DEBUG_ADDR=1
for r in 1 2 3 4 5 6 7 8; do
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2> /tmp/addr.err | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('run $r build_ms %.3f' % d['build_ms'], {k: round(v['avg_launch_ms'],3) for k, v in d['sort_passes']['by_kernel'].items()})
"
  grep "addr" /tmp/addr.err | sort | uniq -c
done
