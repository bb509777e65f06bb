#!/bin/bash
# round 5, collection of the tracked evidence on the final code (part 1: suite, configs; part 2: profiles)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
if [ "$1" = "1" ]; then
  timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
  tail -4 $O/pytest.log
  timeout 1500 bash scripts/bench_configs.sh > $O/configs.log 2> $O/configs.err
  wc -l $O/configs.log
  timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
  tail -c 600 $O/bench.json
else
  bash scripts/profile.sh r05 > $O/profile_r05.log 2>&1
  PROF_CONFIG=configs/PSMNet/kitti_2015.py bash scripts/profile.sh r05kitti > $O/profile_r05kitti.log 2>&1
  PROF_BATCH=1 bash scripts/profile.sh r05b1 > $O/profile_r05b1.log 2>&1
  bash scripts/profile_train.sh r05 > $O/profile_train.log 2>&1
  for t in r05 r05kitti r05b1; do
    f=$(find $R/gpurun_out/prof_$t/trace -name 'trace_kernel_trace.csv' | head -1)
    python scripts/step_sequence.py $f > $O/step_sequence_$t.log 2>&1
  done
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace -d $O/trace_cfg0 -o trace --output-format csv -- python $R/bench.py --config $R/configs/PSMNet/baseline_cfg0_256x512_d64.py --batch 1 --no-cpu-baseline --no-extras --no-latency --steps 3 --warmup 2 > $O/trace_cfg0.log 2>&1
  python $R/scripts/step_sequence.py $(find $O/trace_cfg0 -name 'trace_kernel_trace.csv' | head -1) > $O/step_sequence_cfg0_b1.log 2>&1
  cd $R
  python scripts/kbench_hg.py > $O/kbench_hg.log 2>&1
  find $R/gpurun_out -name '*.csv' -size +30M -delete
  ls $O
fi
