"""Caller-side glue of the hot path: the part of `eval_model` (eval_models_seq.py:183-222,242)
that assembles a sequence, pads it, chunks it and crops the reconstructions.

`Croper` has the interface of `utils_func/inference_utils.py:69-114`; its results are pinned by a fixture of
the reference's outputs (tests/golden/croper.json).
"""
from typing import Iterable, List, NamedTuple, Optional, Sequence

import torch


def optimal_crop_size(max_size: int, max_subsample_factor: int, safety_margin: int = 0) -> int:
    """Smallest multiple of 2**max_subsample_factor that is >= max_size (inference_utils.py:26-32)."""
    step = 1 << max_subsample_factor
    return -(-max_size // step) * step


class _Axis(NamedTuple):
    """One image axis: `n` sensor pixels inside `full` network pixels."""
    n: int
    full: int
    before: int          # zero pixels in front of the data (the larger half of an odd surplus)
    after: int
    lo: int              # the centred window [lo, hi) that crop() returns
    hi: int

    @staticmethod
    def plan(n: int, levels: int) -> '_Axis':
        full = optimal_crop_size(n, levels)
        surplus = full - n
        lo = full // 2 - n // 2
        return _Axis(n, full, surplus - surplus // 2, surplus // 2, lo, lo + n)


class Croper:
    """`Croper(num_encoders).pad(x)` / `.crop(img)` with the reference's interface and results
    (utils_func/inference_utils.py:69-114, used at eval_models_seq.py:195-207,242): zero-pad H and W up to
    multiples of 2**num_encoders with the odd pixel in front, and cut the centred sensor window back out.
    The attribute names the reference exposes (`height_crop_size`, `padding_top`, `iy0`, ...) are kept as
    read-only views of the two axis plans; pinned by tests/golden/croper.json (reference outputs)."""

    def __init__(self, num_encoders: int):
        self.num_encoders = num_encoders
        self._y: Optional[_Axis] = None
        self._x: Optional[_Axis] = None

    def update_params(self, width: int, height: int):
        self._x = _Axis.plan(int(width), self.num_encoders)
        self._y = _Axis.plan(int(height), self.num_encoders)

    # the reference's attribute names
    width = property(lambda self: self._x.n if self._x else None)
    height = property(lambda self: self._y.n if self._y else None)
    width_crop_size = property(lambda self: self._x.full if self._x else None)
    height_crop_size = property(lambda self: self._y.full if self._y else None)
    padding_left = property(lambda self: self._x.before)
    padding_right = property(lambda self: self._x.after)
    padding_top = property(lambda self: self._y.before)
    padding_bottom = property(lambda self: self._y.after)
    ix0 = property(lambda self: self._x.lo)
    ix1 = property(lambda self: self._x.hi)
    iy0 = property(lambda self: self._y.lo)
    iy1 = property(lambda self: self._y.hi)

    def pad(self, x: torch.Tensor) -> torch.Tensor:
        h, w = int(x.shape[-2]), int(x.shape[-1])
        if self._y is not None and (h, w) == (self._y.full, self._x.full):
            return x                                    # already network-sized
        if self._y is None or (h, w) != (self._y.n, self._x.n):
            self.update_params(w, h)
        ay, ax = self._y, self._x
        out = x.new_zeros(x.shape[:-2] + (ay.full, ax.full))
        out[..., ay.before:ay.before + h, ax.before:ax.before + w] = x
        return out

    def crop(self, img: torch.Tensor) -> torch.Tensor:
        if self.num_encoders == -1:
            return img
        return img[..., self._y.lo:self._y.hi, self._x.lo:self._x.hi]


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
