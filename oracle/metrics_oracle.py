"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's quality metrics (evaluate/metrics.py:42-65).

`mse_loss` is torch's own F.mse_loss (what the reference calls).  `structural_similarity` restates
skimage.metrics.structural_similarity for the reference's call `ssim(a, b)` on float32 images with default arguments:
that call is only accepted by scikit-image <= 0.18 (later versions demand data_range for floats), where it means
  win_size 7, uniform filter (scipy.ndimage.uniform_filter, which is what scikit-image calls), float64 arithmetic,
  data_range = dtype range of floats = 2, K1 0.01, K2 0.03, use_sample_covariance=True, mean over the map cropped by 3.
scikit-image itself is NOT installed here and nothing of it is in the reference mount: this restatement is PARITY UNPINNED
(checked only against hand-computable properties); the product never imports this module."""
import numpy as np
import torch
from scipy.ndimage import uniform_filter


def mse_loss(y_input: torch.Tensor, y_target: torch.Tensor) -> torch.Tensor:
    return torch.nn.functional.mse_loss(y_input, y_target)


def ssim_image(im1: np.ndarray, im2: np.ndarray, data_range: float = 2.0) -> float:
    im1 = im1.astype(np.float64)
    im2 = im2.astype(np.float64)
    win, K1, K2 = 7, 0.01, 0.03
    NP = win ** 2
    cov_norm = NP / (NP - 1)
    ux, uy = uniform_filter(im1, size=win), uniform_filter(im2, size=win)
    uxx, uyy, uxy = uniform_filter(im1 * im1, size=win), uniform_filter(im2 * im2, size=win), uniform_filter(im1 * im2, size=win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win - 1) // 2
    return float(S[pad:-pad, pad:-pad].mean(dtype=np.float64))


def structural_similarity(y_input: torch.Tensor, y_target: torch.Tensor) -> float:
    a, b = y_input.cpu().numpy(), y_target.cpu().numpy()
    N, C, H, W = a.shape
    assert C == 1 or C == 3
    total = 0.0
    for i in range(N):
        total += float(np.mean([ssim_image(a[i, c], b[i, c]) for c in range(C)]))
    return total / float(N)
