#!/bin/bash
# The training step under rocprofv3:  bash scripts/profile_train.sh r05 [train_bench flags]
#   kernel trace + stats, then separate counter passes (never combined with sys/hip tracing); scripts/summarize_train_profile.py <tag>
#   reduces them to profiles/<tag>_train_kernel_stats.csv and profiles/<tag>_train_pmc.csv
tag=${1:-r05}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_train_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
T="python $GRAFT_REPO_ROOT/scripts/train_bench.py $@"
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o trace --output-format csv -- $T --steps 5 --warmup 2 > $out/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o pmc --output-format csv -- $T --steps 1 --warmup 1 > $out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o pmc --output-format csv -- $T --steps 1 --warmup 1 > $out/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/pmc_sq -o pmc --output-format csv -- $T --steps 1 --warmup 1 > $out/pmc_sq.log 2>&1
find $out -name '*.csv' -size +30M -delete
python $GRAFT_REPO_ROOT/scripts/summarize_train_profile.py $tag
ls $out
