"""``calc_mean_std`` / ``adaptive_instance_normalization`` on the gfx950 kernels.

Same names, argument meaning, return shapes and assertions as the reference
(Style_3DGS/AdaIN/function.py:4-23).  Inputs are GPU tensors shaped [N,C,H,W]; both memory
formats are served without a copy: standard contiguous (NCHW in memory) and ``channels_last``
(NHWC in memory — what this package's encoder returns).  No CPU fallback.

``coral`` (function.py:26-67, colour preservation; host-side 3x3 linear algebra on CPU tensors,
only reached with ``preserve_color=True``, which no caller in the reference sets) is restated in
torch as in the reference because it is outside the GPU hot path.
"""
import torch

from .. import runtime as rt


def _layout(feat):
    """Returns (tensor_view, nhwc) where tensor_view is contiguous in the chosen layout."""
    if feat.is_contiguous():
        return feat, False
    if feat.is_contiguous(memory_format=torch.channels_last):
        return feat.permute(0, 2, 3, 1), True
    return feat.contiguous(), False


def _like(out_view, nhwc):
    return out_view.permute(0, 3, 1, 2) if nhwc else out_view


def calc_mean_std(feat, eps=1e-5):
    # eps is a small value added to the variance to avoid divide-by-zero (function.py:5)
    size = feat.size()
    assert (len(size) == 4)
    N, C = size[:2]
    view, nhwc = _layout(feat.float() if feat.dtype != torch.float32 else feat)
    mean, std = rt.mean_std(view, nhwc, eps)
    return mean.view(N, C, 1, 1), std.view(N, C, 1, 1)


def adaptive_instance_normalization(content_feat, style_feat):
    assert (content_feat.size()[:2] == style_feat.size()[:2])
    N, C = content_feat.size()[:2]
    s_mean, s_std = calc_mean_std(style_feat)
    c_mean, c_std = calc_mean_std(content_feat)
    view, nhwc = _layout(content_feat)
    out = rt.blend_alpha(view, nhwc, c_mean.view(N, C), c_std.view(N, C), s_mean.view(N, C), s_std.view(N, C), 1.0)
    return _like(out, nhwc)


# ---- colour preservation (host side, CPU tensors, as in the reference) ------------------------------
def _calc_feat_flatten_mean_std(feat):
    assert (feat.size()[0] == 3)
    assert (isinstance(feat, torch.FloatTensor))
    flat = feat.view(3, -1)
    return flat, flat.mean(dim=-1, keepdim=True), flat.std(dim=-1, keepdim=True)


def _mat_sqrt(x):
    U, D, Vh = torch.linalg.svd(x)
    return U @ torch.diag(D.pow(0.5)) @ Vh


def coral(source, target):
    """CORAL colour transfer of ``source`` [3,H,W] towards ``target`` [3,H,W] (function.py:41-67)."""
    s_f, s_mean, s_std = _calc_feat_flatten_mean_std(source)
    s_norm = (s_f - s_mean) / s_std
    s_cov = s_norm @ s_norm.t() + torch.eye(3)
    t_f, t_mean, t_std = _calc_feat_flatten_mean_std(target)
    t_norm = (t_f - t_mean) / t_std
    t_cov = t_norm @ t_norm.t() + torch.eye(3)
    transfer = _mat_sqrt(t_cov) @ (torch.inverse(_mat_sqrt(s_cov)) @ s_norm)
    return (transfer * t_std + t_mean).view(source.size())
