"""Group-wise correlation volumes (GwcNet).  The reference has no implementation (README.md:16 names the model
only); these fill the ``COR_FUNCS`` slot of cost_processors/utils/correlation1d_cost.py:29-31 with the same
call signature as the other builders.  Spec: SURVEY.md section 8-a4."""
import torch

from ..... import ops


def gwc_fms(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1, disp_sample=None, num_groups=40):
    """[B, C, H, W] x 2 -> [B, G, D, H, W]: per-group mean of L[c, y, x] * R[c, y, x - d]."""
    idx = ops.disp_index_list(max_disp, start_disp, dilation)
    return ops.gwc_fms(reference_fm.float(), target_fm.float(), idx, num_groups)


def gwc_cat_fms(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1, disp_sample=None, num_groups=40):
    """GwcNet "gwc + concat" volume.  ``reference_fm``/``target_fm`` are (correlation_features,
    concat_features) pairs: [B, 320, H, W] and [B, 12, H, W]; output [B, G + 2*Cc, D, H, W] -- 64 channels for
    G=40, Cc=12, i.e. what PSMAggregator(in_planes=64) consumes.  Both parts are written in place into one
    tensor (no torch.cat pass)."""
    (lg, lc), (rg, rc) = reference_fm, target_fm
    idx = ops.disp_index_list(max_disp, start_disp, dilation)
    B, _, H, W = lg.shape
    Cc = lc.shape[1]
    out = torch.empty((B, num_groups + 2 * Cc, len(idx), H, W), dtype=torch.float32, device=lg.device)
    ops.gwc_fms(lg.float(), rg.float(), idx, num_groups, out=out, out_ch_offset=0)
    ops.cat_fms_into(lc.float(), rc.float(), idx, out, num_groups)
    return out


COR_FUNCS = dict(
    default=gwc_fms,
    gwc=gwc_fms,
    gwc_cat=gwc_cat_fms,
)
