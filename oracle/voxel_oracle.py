"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the event -> voxel-grid binning.

Restates `events_to_voxel_torch` (events_contrast_maximization/utils/event_utils.py:466-509)
with its call into the nearest-pixel branch of `events_to_image_torch` (:330-376, esp.
:360 zero image, :371-375 float->long truncation + index_put_(accumulate=True)).
numpy float32 throughout; accumulation is in event order per bin (np.add.at is
unbuffered and sequential), which is what the reference's CPU index_put_ does, so the
restatement is bit-exact against the reference on CPU (pinned by tests/golden/voxel_*.npz).
The product never imports this module.
"""
import numpy as np


def events_to_voxel(xs, ys, ts, ps, num_bins, sensor_size):
    """xs, ys, ts, ps: float32 [N].  Returns float32 [num_bins, H, W].

    t_norm = (t - t0) / (t_last - t0) * (B-1)            (:489-490)
    bin b += p * max(0, 1 - |t_norm - b|) at (long(y), long(x))   (:494-498, 371-375)
    No clipping (clip_out_of_range=False, :498); dt == 0 gives NaN weights like the reference.
    """
    xs = np.asarray(xs, dtype=np.float32)
    ys = np.asarray(ys, dtype=np.float32)
    ts = np.asarray(ts, dtype=np.float32)
    ps = np.asarray(ps, dtype=np.float32)
    H, W = sensor_size
    B = int(num_bins)
    with np.errstate(divide='ignore', invalid='ignore'):
        dt = np.float32(ts[-1] - ts[0])
        t_norm = ((ts - ts[0]) / dt * np.float32(B - 1)).astype(np.float32)
    xi = xs.astype(np.int64)   # C-style truncation, like Tensor.long()
    yi = ys.astype(np.int64)
    out = np.zeros((B, H, W), dtype=np.float32)
    for b in range(B):
        w = np.maximum(np.float32(0.0), np.float32(1.0) - np.abs(t_norm - np.float32(b))).astype(np.float32)
        np.add.at(out[b], (yi, xi), (ps * w).astype(np.float32))
    return out


def synthetic_events(n, height, width, seed):
    """Synthetic event packet of SURVEY.md §8(d): x~U{0..W-1}, y~U{0..H-1} as float32 integers,
    t = sorted U(0,1) float32 minus first, p in {-1,+1}."""
    rng = np.random.default_rng(seed)
    xs = rng.integers(0, width, n).astype(np.float32)
    ys = rng.integers(0, height, n).astype(np.float32)
    ts = np.sort(rng.random(n, dtype=np.float32))
    ts = (ts - ts[0]).astype(np.float32)
    ps = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    return xs, ys, ts, ps
