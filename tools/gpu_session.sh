#!/bin/bash
# Runs GPU steps one after another; stops at the first step that times out (never start another
# GPU step after a kill).  Usage: tools/gpu_session.sh "<timeout_s> <logname> <cmd...>" ...
mkdir -p gpurun_out
export TMPDIR=/tmp
for spec in "$@"; do
  set -- $spec
  t=$1; log=$2; shift 2
  echo "=== [$log] timeout ${t}s: $*" | tee -a gpurun_out/session.log
  timeout -k 10 "$t" "$@" > "gpurun_out/$log.log" 2>&1
  rc=$?
  echo "=== [$log] exit $rc" | tee -a gpurun_out/session.log
  tail -n 15 "gpurun_out/$log.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping session"; exit $rc; fi
done
exit 0
