"""The synthetic workload of bench.py and the check of its output against the reference's.

`bench_voxels` assembles the inputs the way `eval_model` does (eval_models_seq.py:183-207): one voxel grid per frame
from that frame's events (`events_to_voxel_torch`, here the HIP scatter), zero-padded to the network size.
`verify_against_fixture` compares reconstructed frames with the committed fixture of the workload (`find_fixture`), which
holds what the REFERENCE produced for exactly these events (oracle/gen_golden.py::gen_bench_fixture / ::gen_bench_fullsize):
pixels at a stride plus the per-frame mean and standard deviation.
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

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
# what the REFERENCE produced for bench.py's synthetic workloads (oracle/gen_golden.py::gen_bench_fixture,
# ::gen_bench_fullsize): BASELINE configs 2, 3 and 5 at their full sizes
FIXTURES = ('e2e_bench_T16', 'e2e_bench_480x640_T32_B4', 'e2e_bench_720x1280_T64')
SEED0 = 1000              # event seed of frame 0 of batch element 0; the fixtures record theirs (meta['seed'])
TOLERANCE = 2e-4          # max-abs on the sigmoid output (north_star: 1e-3)


def bench_voxels(T: int, sensor_hw: Tuple[int, int], device, seed0: int = SEED0, num_bins: int = 5, num_encoders: int = 3,
                 batch: int = 1):
    """-> (voxels [T, batch, num_bins, Hp, Wp] on `device`, number of events, seconds spent binning incl. H2D).
    Batch element b of frame t is binned from the events with seed seed0 + b*T + t (every element a different stream)."""
    sh, sw = sensor_hw
    packs = [synthetic_events(sh * sw // 2, sh, sw, seed0 + b * T + t) for t in range(T) for b in range(batch)]
    off = np.cumsum([0] + [len(p[0]) for p in packs])
    cat = [torch.from_numpy(np.concatenate([p[k] for p in packs])) for k in range(4)]
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    grids = events_to_voxel_batch(*cat, off, num_bins, sensor_size=(sh, sw), device=device)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    vox = Croper(num_encoders).pad(grids.reshape(T, batch, num_bins, sh, sw))
    return vox.contiguous(), int(off[-1]), dt


def find_fixture(T: int, B: int, H: int, W: int, sensor_hw) -> Optional[Tuple[str, dict]]:
    """(path, meta) of the committed reference fixture for this workload, or None."""
    for name in FIXTURES:
        path = os.path.join(GOLDEN_DIR, name + '.npz')
        if not os.path.exists(path):
            continue
        meta = json.loads(str(np.load(path)['meta']))
        if (meta['T'], meta['B'], meta['H'], meta['W']) == (T, B, H, W) and list(meta['sensor']) == list(sensor_hw):
            return path, meta
    return None


def verify_against_fixture(frames: torch.Tensor, path: str):
    """frames: [T, B, 1, Hp, Wp] as returned by the model for bench_voxels(T, sensor, seed0=meta['seed'], batch=B).
    -> (ok, max abs error over the stored pixels)."""
    z = np.load(path)
    meta = json.loads(str(z['meta']))
    s = meta['stride']
    sub = frames[..., ::s, ::s].detach().float().cpu().numpy()
    if sub.shape != z['out'].shape:
        return False, float('inf')
    err = float(np.abs(sub.astype(np.float64) - z['out']).max())
    # per-frame statistics on the device (an HD sequence is 59 M pixels)
    f = frames.detach().double()
    mean = f.mean(dim=(1, 2, 3, 4)).cpu().numpy()
    std = f.std(dim=(1, 2, 3, 4), unbiased=False).cpu().numpy()
    stats = bool(np.allclose(mean, z['mean'], atol=1e-5) and np.allclose(std, z['std'], atol=1e-5))
    return bool(err <= TOLERANCE and stats), err
