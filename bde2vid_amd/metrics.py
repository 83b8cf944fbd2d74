"""Quality metrics of `eval_model`'s scoring loop (eval_models_seq.py:229-258) computed on the GPU.

Same names, arguments and return meaning as the reference's `evaluate/metrics.py`:
`mse_loss(y_input, y_target)` (:42-43) and `structural_similarity(y_input, y_target)` (:46-65, [N,C,H,W] with C in {1,3},
mean over the batch).  `perceptual_loss` (LPIPS, :69-97) needs the AlexNet + linear-layer weights that are not in the mount
and is not built.  SSIM follows scikit-image's published algorithm for the call the reference makes (float images, no
data_range -> scikit-image <= 0.18: float64, data_range 2); scikit-image is absent from this image, so its parity is
UNPINNED (see csrc/metrics.h); MSE is pinned by the CPU oracle.
"""
import ctypes as C

import torch

from . import _lib


def _prep(y_input, y_target):
    if y_input.shape != y_target.shape:
        raise ValueError(f'shape mismatch {tuple(y_input.shape)} vs {tuple(y_target.shape)}')
    if not (y_input.is_cuda and y_target.is_cuda):
        raise RuntimeError('bde2vid_amd.metrics runs on the GPU only')
    a = y_input.detach().to(torch.float32).contiguous()
    b = y_target.detach().to(device=a.device, dtype=torch.float32).contiguous()
    return a, b


def _run(fn_name, a, b, n_images, *dims):
    L = _lib.lib()
    dev = a.device
    scratch = torch.empty(int(L.bde_metric_scratch_doubles(n_images)), dtype=torch.float64, device=dev)
    out = torch.empty(n_images, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        st = C.c_void_p(int(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(getattr(L, fn_name)(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), *dims, C.c_void_p(scratch.data_ptr()),
                                       C.c_void_p(out.data_ptr()), st))
    return out


def mse_per_image(y_input: torch.Tensor, y_target: torch.Tensor) -> torch.Tensor:
    """[N, ...] -> float64 [N] on the GPU: mean squared error of every image."""
    a, b = _prep(y_input, y_target)
    n = a.shape[0]
    return _run('bde_metric_mse', a, b, n, a[0].numel(), n)


def mse_loss(y_input: torch.Tensor, y_target: torch.Tensor) -> torch.Tensor:
    """F.mse_loss(y_input, y_target) (evaluate/metrics.py:42-43): a 0-dim float32 tensor on the GPU."""
    return mse_per_image(y_input, y_target).mean().to(torch.float32)


def ssim_per_image(y_input: torch.Tensor, y_target: torch.Tensor, data_range: float = 2.0) -> torch.Tensor:
    """[N, C, H, W] -> float64 [N]: per image, mean over its channels (skimage `multichannel=True` averages them)."""
    a, b = _prep(y_input, y_target)
    N, Cc, H, W = a.shape
    assert Cc == 1 or Cc == 3                                              # evaluate/metrics.py:50
    per = _run('bde_metric_ssim', a, b, N * Cc, H, W, N * Cc, C.c_double(data_range))
    return per.view(N, Cc).mean(dim=1)


def structural_similarity(y_input: torch.Tensor, y_target: torch.Tensor) -> float:
    """evaluate/metrics.py:46-65: mean SSIM over the batch, as a Python float."""
    return float(ssim_per_image(y_input, y_target).mean().item())


def perceptual_loss(*args, **kwargs):
    raise NotImplementedError('LPIPS needs the AlexNet / linear-layer weights (LPIPS/weights), which are not in the mount')


def score_sequence(predictions, targets, metrics=('mse', 'ssim')):
    """The accumulation of eval_model's second loop (eval_models_seq.py:229-258, 264-270): per-frame metric values and
    their means over the sequence.  predictions / targets: sequences of [1,1,H,W] (already cropped) GPU tensors."""
    pred = torch.cat(list(predictions))
    gt = torch.cat([t.to(pred.device) for t in targets])
    detail = {}
    if 'mse' in metrics:
        detail['mse'] = mse_per_image(pred, gt).cpu().tolist()
    if 'ssim' in metrics:
        detail['ssim'] = ssim_per_image(pred, gt).cpu().tolist()
    return {k: sum(v) / len(v) for k, v in detail.items()}, detail
