#!/bin/bash
# Development aid: the kernels of one training step in launch order.   scripts/train_seq.sh <tag> [train_bench flags]
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace_train_$tag -o trace --output-format csv -- python $R/scripts/train_bench.py --steps 4 --warmup 2 "$@" > $O/trace_train_$tag.log 2>&1
python $R/scripts/train_step_sequence.py $(find $O/trace_train_$tag -name 'trace_kernel_trace.csv' | head -1) > $O/train_step_sequence_$tag.log 2>&1
find $O/trace_train_$tag -name '*.csv' -size +20M -delete
python $R/scripts/train_bench.py --steps 10 --warmup 3 "$@" > $O/train_bench_$tag.log 2>&1
tail -2 $O/train_bench_$tag.log
