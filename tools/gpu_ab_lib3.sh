#!/bin/bash
for r in 1 2; do for l in "$@"; do echo "SA_HIP_LIB=$l"; SA_HIP_LIB=$l timeout -k 10 200 python tools/gpu_workloads.py | grep "names_full\|d2_words\|repeat" || exit 1; done; done
