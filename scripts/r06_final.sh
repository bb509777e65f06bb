#!/bin/bash
# round 6, collection of the tracked evidence on the final code.  bash scripts/r06_final.sh 1 : suite, configurations, bench line;
# bash scripts/r06_final.sh 2 : profiles (kernel stats, counters, step sequences), probes
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06f
mkdir -p $O
cd $R
if [ "$1" = "1" ]; then
  timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
  tail -4 $O/pytest.log
  timeout 1800 bash scripts/bench_configs.sh > $O/configs.log 2> $O/configs.err
  wc -l $O/configs.log
  timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
  tail -c 400 $O/bench.json
else
  bash scripts/profile.sh r06 > $O/profile_r06.log 2>&1
  PROF_CONFIG=configs/PSMNet/kitti_2015.py bash scripts/profile.sh r06kitti > $O/profile_r06kitti.log 2>&1
  PROF_BATCH=1 bash scripts/profile.sh r06b1 > $O/profile_r06b1.log 2>&1
  for t in r06 r06kitti r06b1; do
    f=$(find $R/gpurun_out/prof_$t/trace -name 'trace_kernel_trace.csv' | head -1)
    python scripts/step_sequence.py $f > $O/step_sequence_$t.log 2>&1
  done
  bash scripts/seq_cfg.sh b1_cfg0 PSMNet/baseline_cfg0_256x512_d64.py 1
  bash scripts/seq_cfg.sh b1_kitti PSMNet/kitti_2015.py 1
  cp $R/gpurun_out/r06/step_sequence_b1_cfg0.log $R/gpurun_out/r06/step_sequence_b1_kitti.log $O/
  cd $R
  python scripts/binding_overhead.py > $O/binding_overhead.log 2>&1
  find $R/gpurun_out -name '*.csv' -size +30M -delete
  ls $O
fi
