"""Shared pieces of the model configs of the HIP path.  Each config file is a plain Python file defining
``task``, ``max_disp`` and ``model`` with exactly the keys the reference's builders read (SURVEY.md section 5,
"Config / flags"), so the reference's own config files load unchanged as well (Config.fromfile)."""


def volume(kind, max_disp, scale):
    """cost_computation block: the search range is expressed at the feature resolution (1/scale)."""
    return dict(type=kind, max_disp=int(max_disp // scale), start_disp=0, dilation=1)


def predictor(kind, max_disp, **extra):
    return dict(type=kind, max_disp=max_disp, start_disp=0, dilation=1, alpha=1.0, normalize=True, **extra)


def evaluation(max_disp):
    return dict(lower_bound=0, upper_bound=max_disp, eval_occlusion=False, is_cost_return=False, is_cost_to_cpu=False)
