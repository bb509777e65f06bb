# PSMNet at the reference's published KITTI operating point: 375x1242 padded to 384x1248, max_disp 192
# (reference configs/PSMNet/kitti_2015.py:113,122,129; ResultOfPSMNet.md:15-19).  Same model block as scene_flow.py.
import os, runpy
_c = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "_common.py"))
task = 'stereo'
max_disp = 192
model = dict(
    meta_architecture="GeneralizedStereoModel",
    max_disp=max_disp,
    batch_norm=True,
    backbone=dict(type="PSMNet", in_planes=3),
    cost_processor=dict(
        type='Concatenation',
        cost_computation=_c['volume']("default", max_disp, 4),
        cost_aggregator=dict(type="PSMNet", max_disp=max_disp, in_planes=64),
    ),
    disp_predictor=_c['predictor']('FASTER', max_disp),
    losses=dict(l1_loss=dict(max_disp=max_disp, weights=(1.0, 0.7, 0.5), weight=1.0)),
    eval=_c['evaluation'](max_disp),
)
data = dict(sparse=True, eval=dict(input_shape=[384, 1248], original_shape=[375, 1242], mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375]))
eval_disparity_id = [0, 1, 2]
dist_params = dict(backend='nccl')
