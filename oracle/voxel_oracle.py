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


def events_to_voxel_indexput(xs, ys, ts, ps, num_bins, sensor_size):
    """The same binning with the reference's own operation sequence -- per bin one full pass over the events and one
    `index_put_(accumulate=True)` into a zero image (event_utils.py:492-507 calling :360,371-375) -- on torch CPU
    tensors.  This is the form bench.py times as the host baseline of the voxel path; pinned to the
    same goldens as `events_to_voxel`."""
    import torch
    H, W = sensor_size
    B = int(num_bins)
    dt = ts[-1] - ts[0]                                                   # :489
    t_norm = (ts - ts[0]) / dt * (B - 1)                                  # :490
    xi, yi = xs.long(), ys.long()                                         # :371-372 (truncation)
    bins = []
    for b in range(B):
        w = ps * torch.clamp(1.0 - torch.abs(t_norm - b), min=0.0)        # :494-495  max(0, 1 - |t - b|)
        img = torch.zeros((H, W), dtype=torch.float32)                    # :360
        img.index_put_((yi, xi), w, accumulate=True)                      # :375
        bins.append(img)
    return torch.stack(bins)                                              # :508


def between_frames_voxels(xs, ys, ts, ps, event_idx, num_bins, sensor_size):
    """Voxel grids of consecutive windows from native event columns: the item assembly of
    BaseVoxelDataset.__getitem__ (data_loader/h5_dataset.py:213-226) on the columns DynamicH5Dataset.get_events
    returns (:410-415: xs, ys int16; ts float64; ps = bool * 2.0 - 1.0), then get_voxel_grid (:343-366,
    combined channels, all-ones hot-pixel mask).  Returns float32 [nwin, num_bins, H, W]."""
    xs = np.asarray(xs)
    ys = np.asarray(ys)
    ts = np.asarray(ts, dtype=np.float64)
    ps = np.asarray(ps).astype(np.float64) * 2.0 - 1.0                      # :414
    H, W = sensor_size
    out = np.zeros((len(event_idx) - 1, int(num_bins), H, W), dtype=np.float32)
    for w in range(len(event_idx) - 1):
        i0, i1 = int(event_idx[w]), int(event_idx[w + 1])
        if i1 - i0 < 3:                                                     # :219-220 empty voxel grid
            continue
        x = xs[i0:i1].astype(np.float32)                                    # :222
        y = ys[i0:i1].astype(np.float32)                                    # :223
        t = (ts[i0:i1] - ts[i0]).astype(np.float32)                         # :224
        p = ps[i0:i1].astype(np.float32)                                    # :225
        out[w] = events_to_voxel(x, y, t, p, num_bins, sensor_size)
    return out


# input generators live with the product's host code (bench.py uses them too); re-exported for the tests
from bde2vid_amd.synth import synthetic_events, synthetic_recording  # noqa: E402,F401
