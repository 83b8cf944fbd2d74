"""Caller-side glue of the hot path: the part of `eval_model` (eval_models_seq.py:183-222,242)
that assembles a sequence, pads it, chunks it and crops the reconstructions.

`Croper` mirrors `utils_func/inference_utils.py:69-114` (same method names and arithmetic).
"""
from math import ceil, floor
from typing import Iterable, List, Optional, Sequence

import torch
import torch.nn.functional as F


def optimal_crop_size(max_size: int, max_subsample_factor: int, safety_margin: int = 0) -> int:
    """Smallest multiple of 2**max_subsample_factor that is >= max_size (inference_utils.py:26-32)."""
    m = 2 ** max_subsample_factor
    return int(m * ceil(max_size / m))


class Croper:
    """Zero-pad to the network size (ceil on top/left) and centre-crop back."""

    def __init__(self, num_encoders: int):
        self.width = self.height = None
        self.height_crop_size = self.width_crop_size = None
        self.num_encoders = num_encoders

    def update_params(self, width: int, height: int):
        self.width, self.height = width, height
        n = self.num_encoders
        self.width_crop_size = optimal_crop_size(width, n)
        self.height_crop_size = optimal_crop_size(height, n)
        self.padding_top = ceil(0.5 * (self.height_crop_size - height))
        self.padding_bottom = floor(0.5 * (self.height_crop_size - height))
        self.padding_left = ceil(0.5 * (self.width_crop_size - width))
        self.padding_right = floor(0.5 * (self.width_crop_size - width))
        self.cx = floor(self.width_crop_size / 2)
        self.cy = floor(self.height_crop_size / 2)
        self.ix0 = self.cx - floor(width / 2)
        self.ix1 = self.cx + ceil(width / 2)
        self.iy0 = self.cy - floor(height / 2)
        self.iy1 = self.cy + ceil(height / 2)

    def pad(self, x: torch.Tensor) -> torch.Tensor:
        h, w = x.shape[-2:]
        if h != self.height_crop_size or w != self.width_crop_size:
            if h != self.height or w != self.width:
                self.update_params(w, h)
            x = F.pad(x, (self.padding_left, self.padding_right, self.padding_top, self.padding_bottom))
        return x

    def crop(self, img: torch.Tensor) -> torch.Tensor:
        return img[..., self.iy0:self.iy1, self.ix0:self.ix1] if self.num_encoders != -1 else img


def chunked(seq: Sequence, n: Optional[int]) -> Iterable[Sequence]:
    """more_itertools.chunked as used at eval_models_seq.py:218 (None = one chunk)."""
    if n is None:
        yield seq
        return
    for i in range(0, len(seq), n):
        yield seq[i:i + n]


def reconstruct_sequence(model, voxels: Sequence[torch.Tensor], subseq_L: Optional[int] = 1000) -> List[torch.Tensor]:
    """voxels: T tensors [B, num_bins, H, W] on the model's device -> T cropped frames [B,1,H,W].

    Pads with Croper(num_encoders) (eval_models_seq.py:195-207), runs the model on chunks of at
    most subseq_L frames with a fresh state each (:214-221) and crops (:242).
    """
    try:
        n_enc = model.num_encoders
    except Exception:
        n_enc = 3
    crop = Croper(n_enc)
    h, w = voxels[0].shape[-2:]
    crop.update_params(w, h)
    inputs = [{'events': crop.pad(v)} for v in voxels]
    preds: List[torch.Tensor] = []
    with torch.no_grad():
        for sub in chunked(inputs, subseq_L):
            preds += model(list(sub))
    return [crop.crop(p) for p in preds]
