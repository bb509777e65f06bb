# GC-Net: concatenation volume at 1/2 resolution, 3-D encoder/decoder aggregator, soft-argmin at full resolution.
import os, runpy
_c = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "_common.py"))
task = 'stereo'
max_disp = 192
model = dict(
    meta_architecture="GeneralizedStereoModel",
    max_disp=max_disp,
    batch_norm=True,
    backbone=dict(type="GCNet", in_planes=3),
    cost_processor=dict(
        type='Concatenation',
        cost_computation=_c['volume']("default", max_disp, 2),
        cost_aggregator=dict(type="GCNet", max_disp=max_disp, in_planes=64),
    ),
    disp_predictor=_c['predictor']('FASTER', max_disp),
    losses=dict(l1_loss=dict(max_disp=max_disp, weights=(1.0,), weight=1.0)),
    eval=_c['evaluation'](max_disp),
)
data = dict(sparse=False, eval=dict(input_shape=[544, 960], original_shape=[540, 960], mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375]))
eval_disparity_id = [0]
dist_params = dict(backend='nccl')
