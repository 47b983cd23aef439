#!/bin/bash
# tools/gpu_workloads.py with SA_HIP_LOCAL_ROUNDS = 1 / 0 (rounds sorted in LDS, round_sort.hpp, against the global sort)
for m in 1 0 1 0; do echo "SA_HIP_LOCAL_ROUNDS=$m"; SA_HIP_LOCAL_ROUNDS=$m timeout -k 10 200 python tools/gpu_workloads.py | grep -v "d1_\|dna\|bytes" || exit 1; done
