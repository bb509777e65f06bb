#!/bin/bash
# round 5, first GPU call: the GPU suite on the new code, the batch-1 lines, per-kernel traces of a batch-1 step
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 900 bash scripts/bench_b1.sh > $O/b1.log 2> $O/b1.err
cat $O/b1.log | cut -c1-1500
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
for c in PSMNet/baseline_cfg0_256x512_d64.py PSMNet/scene_flow.py; do
  t=$(basename $c .py)
  timeout 300 rocprofv3 --kernel-trace -d $O/trace_$t -o trace --output-format csv -- python $R/bench.py --config $R/configs/$c --batch 1 --no-cpu-baseline --no-extras --no-latency --steps 3 --warmup 2 > $O/trace_$t.log 2>&1
  python $R/scripts/step_sequence.py $(find $O/trace_$t -name 'trace_kernel_trace.csv' | head -1) > $O/step_sequence_b1_$t.log 2>&1
  find $O/trace_$t -name '*.csv' -size +20M -delete
done
ls -la $O
