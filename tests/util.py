"""Helpers shared by the CPU and GPU test files."""
import json
import os

import numpy as np
import torch

from bde2vid_amd.config import GeneratorConfig
from bde2vid_amd.weights import formula_state_dict
from oracle.gen_golden import (golden_inputs, voxel_like, dense_like, voxel_case, E2E_CASES, CFGA_FULL,  # noqa: F401
                               CFGA_SAMPLED, LONGT_CASES, BENCH_FIXTURE, VARIANT_CASES)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    meta = json.loads(str(z['meta'])) if 'meta' in z.files else {}
    return z, meta


def case_from_meta(meta):
    cfg = GeneratorConfig.from_dict(meta['cfg'])
    sd = formula_state_dict(cfg, meta['weight_seed'])
    xs = golden_inputs(meta['T'], meta['B'], cfg.num_bins, meta['H'], meta['W'], meta['seed'])
    return cfg, sd, xs


def maxabs(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())


def bench_fixture_inputs(meta):
    """Inputs of tests/golden/e2e_bench_T16.npz = bench.py's workload: synthetic events (bde2vid_amd/synth.py) binned by
    the CPU restatement of the voxel binning, zero-padded to the network size like Croper.pad.  CPU tensors."""
    from oracle import voxel_oracle
    from bde2vid_amd.harness import Croper
    sh, sw = meta['sensor']
    crop = Croper(3)
    out = []
    for t in range(meta['T']):
        xs, ys, ts, ps = voxel_oracle.synthetic_events(meta['events_per_frame'], sh, sw, meta['seed'] + t)
        g = voxel_oracle.events_to_voxel(xs, ys, ts, ps, 5, (sh, sw))
        out.append(crop.pad(torch.from_numpy(g)[None]))
    return out


def assert_sampled(y, z, meta, tol, mean_tol):
    """y: [T,B,1,H,W] (torch or numpy) against a fixture stored as strided pixels + per-frame mean/std."""
    y = y.detach().cpu().numpy() if isinstance(y, torch.Tensor) else np.asarray(y)
    s = meta['stride']
    assert y[..., ::s, ::s].shape == z['out'].shape
    err = maxabs(y[..., ::s, ::s], z['out'])
    assert err <= tol, f'sampled pixels differ by {err}'
    assert np.allclose(y.mean(axis=(1, 2, 3, 4)), z['mean'], atol=mean_tol)
    assert np.allclose(y.std(axis=(1, 2, 3, 4)), z['std'], atol=mean_tol)
