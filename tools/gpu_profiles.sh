#!/bin/bash
# The judged evidence of one state of the code, all on one box: tools/gpu_profiles.sh <tag> [stages]
#   stages (default "bench stats pmc raw valu calib"):
#   bench  gpurun_out/<tag>_bench.json         python bench.py (default flags: CPU baseline + secondaries)
#   stats  gpurun_out/prof_<tag>/              rocprofv3 --kernel-trace --stats of bench.py --steps 2 --warmup 1 (no CPU legs)
#   pmc    gpurun_out/pmc_<tag>_fetch|write/   rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --steps 1 --warmup 0
#   raw    gpurun_out/pmc_<tag>_raw/           the L2's memory-side request counters behind FETCH_SIZE (requests by size)
#   valu   gpurun_out/pmc_<tag>_valu/          instruction counters of every kernel of a build (vector ALU, LDS, wave cycles)
#   calib  gpurun_out/pmc_<tag>_calib_*/       the same counters on tools/gatherbench (random 4- / 8-byte reads of a 4 GiB table)
# The program stands directly behind `--` (no env / bash -c hop: the profiler's library has the GPU initialised by then).
tag=$1
stages=${2:-"bench stats pmc raw valu calib"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LIGHT="--no-cpu-baseline --no-secondary"
for s in $stages; do
case $s in
bench)
  timeout -k 10 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
  tail -c 400 gpurun_out/${tag}_bench.json; echo ;;
stats)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 2 --warmup 1 $LIGHT > gpurun_out/prof_$tag.log 2>&1 || exit 2
  echo stats done ;;
pmc)
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_fetch -- python3 bench.py --steps 1 --warmup 0 $LIGHT > gpurun_out/pmc_${tag}_fetch.log 2>&1 || exit 3
  echo fetch done
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_write -- python3 bench.py --steps 1 --warmup 0 $LIGHT > gpurun_out/pmc_${tag}_write.log 2>&1 || exit 4
  echo write done ;;
raw)
  rocprofv3 -L > gpurun_out/${tag}_counters.txt 2>&1
  timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_raw -- python3 bench.py --steps 1 --warmup 0 $LIGHT > gpurun_out/pmc_${tag}_raw.log 2>&1 || echo "raw pass failed (see log)"
  echo raw done ;;
valu)
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_valu -- python3 bench.py --steps 1 --warmup 0 $LIGHT > gpurun_out/pmc_${tag}_valu.log 2>&1 || echo "valu pass failed (see log)"
  echo valu done ;;
calib)
  timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_calib_fetch -- tools/gatherbench > gpurun_out/pmc_${tag}_calib_fetch.log 2>&1 || exit 6
  timeout -k 10 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_calib_raw -- tools/gatherbench > gpurun_out/pmc_${tag}_calib_raw.log 2>&1 || echo "calib raw pass failed (see log)"
  echo calib done ;;
esac
done
ls gpurun_out/pmc_${tag}_*/*/ 2>/dev/null | head -40
