#!/bin/bash
# Collect the tracked profiles of a round on the GPU box:  bash scripts/profile.sh r02
#   gpurun_out/prof_<tag>/trace      rocprofv3 --kernel-trace --stats of the headline bench command
#   gpurun_out/prof_<tag>/pmc_*      counter passes (one group per run; never combined with sys/hip tracing)
# then scripts/summarize_profiles.py <tag> turns them into profiles/<tag>_kernel_stats.csv, <tag>_pmc.csv, pmc_dominant.json
tag=${1:-r03}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# PROF_CONFIG: another bench configuration (relative to the repository), e.g. configs/PSMNet/kitti_2015.py with tag r04kitti
# PROF_BATCH: pairs per step (default 4); with 1 the latency legs stay off
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-latency ${PROF_CONFIG:+--config $GRAFT_REPO_ROOT/$PROF_CONFIG} ${PROF_BATCH:+--batch $PROF_BATCH}"
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o trace --output-format csv -- $B --steps 5 --warmup 2 > $out/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o pmc --output-format csv -- $B --steps 1 --warmup 1 > $out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o pmc --output-format csv -- $B --steps 1 --warmup 1 > $out/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $out/pmc_sq -o pmc --output-format csv -- $B --steps 1 --warmup 1 > $out/pmc_sq.log 2>&1
[ -n "$PROF_CONFIG$PROF_BATCH" ] && { ls $out; exit 0; }
# the group-wise correlation volume of BASELINE configs[2] (not part of the PSMNet step): same three passes on its own case
for g in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  d=$(echo $g | cut -d' ' -f1 | sed 's/FETCH_SIZE/pmc_fetch/; s/WRITE_SIZE/pmc_write/; s/SQ_VALU_MFMA_BUSY_CYCLES/pmc_sq/')
  timeout 200 rocprofv3 --kernel-trace --pmc $g -d $out/${d}_gwc -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/scripts/kcase.py gwc 3 > $out/${d}_gwc.log 2>&1
done
ls $out
