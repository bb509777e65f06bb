#!/bin/bash
# The batch-1 / latency regime (the reference publishes and serves one pair per call: ResultOfPSMNet.md:15-19, apis/inference.py:191-225):
# BASELINE configs[0] (256x512, max_disp 64), the headline size, the KITTI operating point, AcfNet at KITTI -- each line carries
# "latency" (path and images -> disparity, ms per pair, eager and HIP-graph replay) next to roofline and cpu_baseline.
R=$GRAFT_REPO_ROOT
cd $R
for c in PSMNet/baseline_cfg0_256x512_d64.py PSMNet/scene_flow.py PSMNet/kitti_2015.py AcfNet/kitti_2015_adaptive.py; do
  python bench.py --config $R/configs/$c --batch 1 --steps 40 --warmup 10 --no-extras 2>/dev/null | tail -1
done
