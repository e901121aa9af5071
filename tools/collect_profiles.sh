#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 timing pass + separate PMC passes of bench.py, summarised into gpurun_out/profiles_<tag>/ .
# usage: bash tools/collect_profiles.sh <tag> [bench args...]      e.g.  bash tools/collect_profiles.sh r02_n371
#        bash tools/collect_profiles.sh r02_n20209 --n 20209 --nsteps 2
# Counters are collected in their own passes (no trace domains; the program directly after `--`), as MI355X_MICROARCH.md prescribes:
#   FETCH_SIZE (x2 on gfx950: 128-B requests tallied at 64 B) and WRITE_SIZE (x1), both in KB, both L2-fabric traffic INCLUDING Infinity-Cache hits;
#   SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE (matrix-core utilisation) and SQ_INSTS_VALU_MFMA_MOPS_F64 (x512 = f64 MFMA flops).
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles_$tag; mkdir -p $out
rm -rf gpurun_out/prof_X gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_m gpurun_out/pmc_o
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_X -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-general-path "$@" > $out/bench_under_rocprof_$tag.json 2> $out/err1.log
echo "timing pass done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-general-path "$@" > /dev/null 2> $out/err2.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-general-path "$@" > /dev/null 2> $out/err3.log
echo "traffic passes done"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_m -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-general-path "$@" > /dev/null 2> $out/err4.log
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d gpurun_out/pmc_o -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-general-path "$@" > /dev/null 2> $out/err5.log
echo "mfma passes done"
python tools/profile_summarize.py $tag gpurun_out/prof_X $out
rm -rf gpurun_out/prof_X gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_m gpurun_out/pmc_o
ls -la $out
