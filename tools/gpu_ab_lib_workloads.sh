#!/bin/bash
# tools/gpu_workloads.py through two builds of the library: tools/gpu_ab_lib_workloads.sh <libA> <libB>
for r in 1 2; do for l in "$1" "$2"; do echo "SA_HIP_LIB=$l"; SA_HIP_LIB=$l timeout -k 10 200 python tools/gpu_workloads.py | grep -v "d1_\|dna\|bytes\|all_a" || exit 1; done; done
