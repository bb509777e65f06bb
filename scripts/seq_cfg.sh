#!/bin/bash
# Development aid: the kernels of one step in launch order (scripts/step_sequence.py) for one configuration at one batch.
#   scripts/seq_cfg.sh <tag> <config relative to configs/> <batch>
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace_$1 -o trace --output-format csv -- python $R/bench.py --config $R/configs/$2 --batch $3 --no-cpu-baseline --no-extras --no-latency --steps 3 --warmup 2 > $O/trace_$1.log 2>&1
python $R/scripts/step_sequence.py $(find $O/trace_$1 -name 'trace_kernel_trace.csv' | head -1) > $O/step_sequence_$1.log 2>&1
find $O/trace_$1 -name '*.csv' -size +20M -delete
