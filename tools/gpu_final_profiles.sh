#!/bin/bash
# The judged evidence of one state of the code, all on one box: tools/gpu_final_profiles.sh <tag>
#   gpurun_out/<tag>_bench.json            python bench.py (default flags, CPU baseline included)
#   gpurun_out/prof_<tag>/                 rocprofv3 --kernel-trace --stats of bench.py --steps 2 --warmup 1 --no-cpu-baseline
#   gpurun_out/pmc_<tag>_fetch|write/      rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --steps 1 --warmup 0
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
tail -c 600 gpurun_out/${tag}_bench.json; echo
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1 || exit 2
echo stats done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_${tag}_fetch.log 2>&1 || exit 3
echo fetch done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_${tag}_write.log 2>&1 || exit 4
echo write done
ls gpurun_out/pmc_${tag}_fetch/*/ gpurun_out/pmc_${tag}_write/*/
