"""The synthetic workload of bench.py and the check of its output against the reference's.

`bench_voxels` assembles the inputs the way `eval_model` does (eval_models_seq.py:183-207): one voxel grid per frame
from that frame's events (`events_to_voxel_torch`, here the HIP scatter), zero-padded to the network size.
`verify_against_fixture` compares reconstructed frames with `tests/golden/e2e_bench_T16.npz`, which holds what the
REFERENCE produced for exactly these events (oracle/gen_golden.py::gen_bench_fixture): pixels at a stride plus the
per-frame mean and standard deviation.
"""
import json
import os
import time
from typing import Optional, Tuple

import numpy as np
import torch

from .events import events_to_voxel_batch
from .harness import Croper
from .synth import synthetic_events

FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'e2e_bench_T16.npz')
TOLERANCE = 2e-4          # max-abs on the sigmoid output (north_star: 1e-3)


def bench_voxels(T: int, sensor_hw: Tuple[int, int], device, seed0: int = 1000, num_bins: int = 5, num_encoders: int = 3):
    """-> (voxels [T, 1, num_bins, Hp, Wp] on `device`, number of events, seconds spent binning incl. H2D)."""
    sh, sw = sensor_hw
    packs = [synthetic_events(sh * sw // 2, sh, sw, seed0 + t) for t in range(T)]
    off = np.cumsum([0] + [len(p[0]) for p in packs])
    cat = [torch.from_numpy(np.concatenate([p[k] for p in packs])) for k in range(4)]
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    grids = events_to_voxel_batch(*cat, off, num_bins, sensor_size=(sh, sw), device=device)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    vox = Croper(num_encoders).pad(grids[:, None])
    return vox.contiguous(), int(off[-1]), dt


def fixture_meta() -> Optional[dict]:
    if not os.path.exists(FIXTURE):
        return None
    return json.loads(str(np.load(FIXTURE)['meta']))


def verify_against_fixture(frames: torch.Tensor):
    """frames: [T, 1, 1, Hp, Wp] as returned by the model for bench_voxels(16, (180, 240), seed0=1000).
    -> (ok, max abs error over the stored pixels)."""
    z = np.load(FIXTURE)
    meta = json.loads(str(z['meta']))
    y = frames.detach().float().cpu().numpy()
    s = meta['stride']
    if y[..., ::s, ::s].shape != z['out'].shape:
        return False, float('inf')
    err = float(np.abs(y[..., ::s, ::s].astype(np.float64) - z['out']).max())
    stats = bool(np.allclose(y.mean(axis=(1, 2, 3, 4)), z['mean'], atol=1e-5) and
                 np.allclose(y.std(axis=(1, 2, 3, 4)), z['std'], atol=1e-5))
    return bool(err <= TOLERANCE and stats), err
