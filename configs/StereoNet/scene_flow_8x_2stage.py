# StereoNet-8x cost path: difference volume at 1/8 resolution, 4 conv units + 1-channel head, soft-argmin at 1/8.
import os, runpy
_c = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "_common.py"))
task = 'stereo'
max_disp = 192
model = dict(
    meta_architecture="GeneralizedStereoModel",
    max_disp=max_disp,
    batch_norm=True,
    cost_processor=dict(
        type='Difference',
        cost_computation=_c['volume']("default", max_disp, 8),
        cost_aggregator=dict(type="StereoNet", max_disp=max_disp, in_planes=32),
    ),
    disp_predictor=_c['predictor']('FASTER', max_disp // 8),
    eval=_c['evaluation'](max_disp),
)
data = dict(sparse=True, eval=dict(input_shape=[384, 1248], original_shape=[375, 1242], mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375]))
eval_disparity_id = [0]
dist_params = dict(backend='nccl')
