#!/bin/bash
# rocprofv3 passes for one round: kernel trace + stats, then PMC counters in their own runs (never combined with
# sys/hip/hsa tracing).  Usage on the GPU box:  bash scripts/profile.sh r01
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH --steps 5 --warmup 2 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCH --steps 1 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $BENCH --steps 1 --warmup 1 > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o pmc -- $BENCH --steps 1 --warmup 1 > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -30
du -sh $OUT
