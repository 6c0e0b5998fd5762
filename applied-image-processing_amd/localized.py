"""Localized (background-only) style transfer with foreground colour harmonisation — the caller of the AdaIN path in the
reference's Style_3DGS/localized_style_transfer.py: the background of the content image is stylised through
``adain_inference(content_mask=background_mask, alpha=1)`` (:207-217), the untouched foreground is then colour-matched to the
stylised background in Reinhard's l-alpha-beta space along each region's first principal axis (:128-168), and the two are
composited (:232-243).

The AdaIN call runs on the MI355X kernels; everything after it is the reference's host-side numpy arithmetic on two uint8
images, restated here with the same names, argument meaning and return types.  Two dependencies of the reference are not
needed: scikit-learn's ``PCA(n_components=1)`` is restated in numpy (covariance eigen-decomposition with scikit-learn >= 1.5's
deterministic sign convention) and the DeepLabV3 background segmentation (:171-188, a network download) is a pluggable
provider (``set_mask_provider``) or a precomputed ``background_mask``.
"""
from pathlib import Path

import numpy as np
from PIL import Image

# Reinhard et al. 2001, "Color Transfer between Images": RGB -> LMS cone space, log10, then the decorrelating l-alpha-beta
# rotation diag(1/sqrt3, 1/sqrt6, 1/sqrt2) @ [[1,1,1],[1,1,-2],[1,-1,0]]  (localized_style_transfer.py:11-19)
RGB_TO_LMS = np.array([[0.3811, 0.5783, 0.0402], [0.1967, 0.7244, 0.0782], [0.0241, 0.1288, 0.8444]])
LMS_TO_LAB = np.diag([1 / np.sqrt(3), 1 / np.sqrt(6), 1 / np.sqrt(2)]) @ np.array([[1, 1, 1], [1, 1, -2], [1, -1, 0]])
LAB_TO_LMS = np.linalg.inv(LMS_TO_LAB)
LMS_TO_RGB = np.linalg.inv(RGB_TO_LMS)


def rgb_to_lab_pixels(pixels_uint8):
    """[N,3] uint8 RGB -> [N,3] float64 l-alpha-beta (:67-77): /255 in float32, LMS clamped at 1e-6 before the log10."""
    lms = np.dot(pixels_uint8.astype(np.float32) / 255.0, RGB_TO_LMS.T)
    return np.dot(np.log10(np.maximum(lms, 1e-6)), LMS_TO_LAB.T)


def lab_to_rgb_pixels(lab_pixels):
    """[N,3] l-alpha-beta -> [N,3] uint8 RGB (:80-89): 10**x, back to RGB, clip to [0,1], x255 and TRUNCATE."""
    rgb = np.dot(np.power(10, np.dot(lab_pixels, LAB_TO_LMS.T)), LMS_TO_RGB.T)
    return (np.clip(rgb, 0, 1) * 255).astype(np.uint8)


def rgb_to_lab_image(image_uint8):
    """[H,W,3] uint8 -> [H,W,3] l-alpha-beta (:22-41)."""
    h, w, _ = image_uint8.shape
    return rgb_to_lab_pixels(image_uint8.reshape(-1, 3)).reshape(h, w, 3)


def lab_to_rgb_image(lab):
    """[H,W,3] l-alpha-beta -> [H,W,3] uint8 (:44-61)."""
    h, w, _ = lab.shape
    return lab_to_rgb_pixels(lab.reshape(-1, 3)).reshape(h, w, 3)


class PCA1:
    """``sklearn.decomposition.PCA(n_components=1)`` as the reference uses it (:92-96), for tall [N,3] data: mean, covariance
    (X^T X - N mu mu^T) / (N - 1), symmetric eigen-decomposition, leading eigenvector with the sign that makes its
    largest-magnitude loading positive (scikit-learn >= 1.5, ``svd_flip(u_based_decision=False)``).  The sign matters here:
    the projections of two regions are CDF-matched against each other."""

    def fit(self, X):
        X = np.asarray(X, dtype=np.float64)
        n = X.shape[0]
        self.mean_ = X.mean(axis=0)
        cov = X.T @ X
        cov -= n * np.outer(self.mean_, self.mean_)
        cov /= n - 1
        vals, vecs = np.linalg.eigh(cov)
        v = vecs[:, np.argmax(vals)]
        if v[np.argmax(np.abs(v))] < 0:
            v = -v
        self.components_ = v[None, :]
        self.explained_variance_ = np.array([max(float(vals.max()), 0.0)])
        return self

    def transform(self, X):
        return np.asarray(X, dtype=np.float64) @ self.components_.T - self.mean_[None, :] @ self.components_.T

    def fit_transform(self, X):
        return self.fit(X).transform(X)

    def inverse_transform(self, P):
        return np.asarray(P, dtype=np.float64) @ self.components_ + self.mean_


def apply_pca(lab_data):
    """Projection of the pixels onto their predominant colour axis (:92-96) -> (projection [N,1], fitted model)."""
    pca = PCA1()
    return pca.fit_transform(lab_data), pca


def match_cdf(target_proj, source_proj):
    """Histogram matching of the 1-D ``target_proj`` to ``source_proj`` (:99-125): both sorted, the shorter quantile function
    resampled to the longer one's length on a uniform [0,1] grid, then target values mapped through t-quantile -> s-quantile."""
    t_sorted = np.sort(target_proj, axis=0).flatten()
    s_sorted = np.sort(source_proj, axis=0).flatten()
    nt, ns = len(t_sorted), len(s_sorted)
    if nt > ns:
        s_sorted = np.interp(np.linspace(0, 1, nt), np.linspace(0, 1, ns), s_sorted)
    elif ns > nt:
        t_sorted = np.interp(np.linspace(0, 1, ns), np.linspace(0, 1, nt), t_sorted)
    return np.interp(target_proj.flatten(), t_sorted, s_sorted).reshape(-1, 1)


def color_transfer_foreground(foreground_img, background_img):
    """Colour-harmonises the non-black pixels of ``foreground_img`` [H,W,3] uint8 with those of ``background_img`` (:128-168):
    both pixel sets go to l-alpha-beta, each is projected on its own first principal axis, the foreground projection is
    CDF-matched to the background's and mapped back through the FOREGROUND's axis (so every adjusted pixel lies on that
    axis).  Black pixels (sum == 0) mark "not in this region" and are left alone; an empty region returns a copy."""
    fg_mask = foreground_img.sum(axis=-1) > 0
    bg_mask = background_img.sum(axis=-1) > 0
    fg_pixels, bg_pixels = foreground_img[fg_mask], background_img[bg_mask]
    if fg_pixels.size == 0:
        print("Warning: No foreground pixels found.")
        return foreground_img.copy()
    if bg_pixels.size == 0:
        print("Warning: No background pixels found for color transfer.")
        return foreground_img.copy()
    fg_proj, fg_pca = apply_pca(rgb_to_lab_pixels(fg_pixels))
    bg_proj, _ = apply_pca(rgb_to_lab_pixels(bg_pixels))
    adjusted = lab_to_rgb_pixels(fg_pca.inverse_transform(match_cdf(fg_proj, bg_proj)))
    out = foreground_img.copy()
    out[fg_mask] = adjusted
    return out


# ---- the pipeline (run_localized_style_transfer, :191-245) -------------------------------------------------------------------------
_mask_provider = None


def set_mask_provider(fn):
    """``fn(PIL.Image RGB) -> np.ndarray [1,H,W] uint8`` background mask (1 = background), the role of the reference's
    ``extract_foreground_deeplab`` (:171-188; torchvision's pretrained DeepLabV3 is a network download).  None clears it."""
    global _mask_provider
    _mask_provider = fn


def extract_foreground_deeplab(content_img, threshold=0.5):
    if _mask_provider is None:
        raise RuntimeError("no background-mask provider: call set_mask_provider(fn) or pass background_mask= "
                           "(the reference downloads torchvision's DeepLabV3 here, which needs network access)")
    return _mask_provider(content_img)


def combine_localized(content_np, stylized_np, background_mask):
    """Steps :218-236 on arrays: ``background_mask`` [H,W] in {0,1}; the stylised image is nearest-resized to the mask when
    sizes differ; foreground = content * (1 - m), background = stylised * m; result = colour-matched foreground * (1 - m) +
    background, as uint8."""
    m = np.asarray(background_mask)
    if stylized_np.shape[:2] != m.shape[:2]:
        stylized_np = np.array(Image.fromarray(stylized_np).resize((m.shape[1], m.shape[0]), Image.NEAREST))
    fg_mask = 1 - m
    foreground = content_np * fg_mask[..., None]
    background = stylized_np * m[..., None]
    adjusted = color_transfer_foreground(foreground, background)
    return (adjusted * fg_mask[..., None] + background).astype(np.uint8)


def run_localized_style_transfer(content_img_path, style_img_path, output_path="../output", file_name="test", use_depth=False,
                                 depth_offset=0.5, depth_prominence=20, background_mask=None, **adain_kwargs):
    """Same parameters and return value (the saved file's path, a str) as the reference (:191-245) plus ``background_mask``
    ([1,H,W] uint8) to bypass the mask provider; extra keyword arguments go to ``adain_inference`` (checkpoint paths,
    ``depth_map=`` ...)."""
    from .AdaIN.test import adain_inference

    content_img = Image.open(content_img_path).convert("RGB")
    content_np = np.array(content_img)
    if background_mask is None:
        background_mask = extract_foreground_deeplab(content_img)
    background_mask = np.asarray(background_mask).astype(np.uint8)
    stylized_path = adain_inference(content_img=content_img_path, style_img=style_img_path, content_mask=background_mask,
                                    output=output_path, file_name=file_name, use_depth=use_depth, depth_offset=depth_offset,
                                    depth_prominence=depth_prominence, alpha=1, **adain_kwargs)
    stylized_np = np.array(Image.open(stylized_path).convert("RGB"))
    combined = combine_localized(content_np, stylized_np, background_mask[0])
    Path(output_path).mkdir(exist_ok=True, parents=True)
    save_path = f"{output_path}/localized_style_transfer_result.jpg"
    Image.fromarray(combined).save(save_path)
    return save_path
