# GwcNet-style volume (40-group correlation of 320-ch features + concat of 2x12-ch features = 64 channels) into the
# PSMNet aggregator.  The reference ships no GwcNet; spec in SURVEY.md section 8-a4.
import os, runpy
_c = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "_common.py"))
task = 'stereo'
max_disp = 192
model = dict(
    meta_architecture="GeneralizedStereoModel",
    max_disp=max_disp,
    batch_norm=True,
    cost_processor=dict(
        type='Correlation',
        cost_computation=dict(_c['volume']("gwc_cat", max_disp, 4), num_groups=40),
        cost_aggregator=dict(type="GwcNet", max_disp=max_disp, in_planes=64),
    ),
    disp_predictor=_c['predictor']('FASTER', max_disp),
    eval=_c['evaluation'](max_disp),
)
data = dict(sparse=False, eval=dict(input_shape=[544, 960], original_shape=[540, 960], mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375]))
eval_disparity_id = [0, 1, 2]
dist_params = dict(backend='nccl')
