#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 timing + PMC passes of bench.py, summarised into gpurun_out/profiles_<tag>/ .
# usage: bash tools/collect_profiles.sh <tag> [bench args...]      e.g.  bash tools/collect_profiles.sh r01
#        bash tools/collect_profiles.sh r01_n20209 --n 20209 --nsteps 2
# PMC counters are collected in their own passes (no trace domains), the program directly after `--`.
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles_$tag; mkdir -p $out
rm -rf gpurun_out/prof_X gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_cal_f gpurun_out/pmc_cal_w
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_X -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/bench_under_rocprof_$tag.json 2> $out/err1.log
echo "timing pass done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> $out/err2.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> $out/err3.log
echo "pmc passes done"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_cal_f -- python tools/pmc_calibrate.py > /dev/null 2> $out/err4.log
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_cal_w -- python tools/pmc_calibrate.py > /dev/null 2> $out/err5.log
echo "calibration passes done"
python tools/profile_summarize.py $tag gpurun_out/prof_X $out
rm -rf gpurun_out/prof_X gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_cal_f gpurun_out/pmc_cal_w
ls -la $out
