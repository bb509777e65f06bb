#!/bin/bash
# Every BASELINE configuration through bench.py (one JSON line each, with parity_vs_cpu and cpu_baseline):  bash scripts/bench_configs.sh > profiles/rNN_configs.log
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1
# the reference's published operating point for PSMNet / AcfNet: KITTI, 384x1248 (configs/PSMNet/kitti_2015.py:113, ResultOfPSMNet.md:15-19)
python bench.py --config $R/configs/PSMNet/kitti_2015.py --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1
python bench.py --config $R/configs/AcfNet/kitti_2015_adaptive.py --steps 10 --warmup 3 --no-extras 2>/dev/null | tail -1
python bench.py --config $R/configs/GwcNet/scene_flow.py --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1
python bench.py --config $R/configs/AcfNet/scene_flow_adaptive.py --steps 10 --warmup 3 --no-extras 2>/dev/null | tail -1
python bench.py --config $R/configs/AcfNet/scene_flow_uniform.py --steps 10 --warmup 3 --no-extras 2>/dev/null | tail -1
python bench.py --config $R/configs/StereoNet/scene_flow_8x_2stage.py --batch 8 --steps 300 --warmup 50 --no-extras 2>/dev/null | tail -1
python bench.py --config $R/configs/GCNet/scene_flow.py --batch 1 --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | tail -1
bash $R/scripts/bench_b1.sh
