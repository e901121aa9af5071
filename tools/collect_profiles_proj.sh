#!/bin/bash
# configs[2] (SteelProfile(1357) Ros2, default ADI() = Projection(2)): rocprofv3 timing pass + separate PMC passes of tools/time_proj1357.py,
# summarised like tools/collect_profiles.sh.   usage: bash tools/collect_profiles_proj.sh <tag>     (run on the GPU box)
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles_$tag; mkdir -p $out
rm -rf gpurun_out/prof_X gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_m gpurun_out/pmc_o
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_X -- python tools/time_proj1357.py 1 > $out/run_under_rocprof_$tag.log 2> $out/err1.log
echo "timing pass done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python tools/time_proj1357.py 1 > /dev/null 2> $out/err2.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python tools/time_proj1357.py 1 > /dev/null 2> $out/err3.log
echo "traffic passes done"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_m -- python tools/time_proj1357.py 1 > /dev/null 2> $out/err4.log
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d gpurun_out/pmc_o -- python tools/time_proj1357.py 1 > /dev/null 2> $out/err5.log
echo "mfma passes done"
python tools/profile_summarize.py $tag gpurun_out/prof_X $out
rm -rf gpurun_out/prof_X gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_m gpurun_out/pmc_o
ls -la $out
