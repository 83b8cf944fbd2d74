"""Helpers shared by the CPU and GPU test files."""
import json
import os

import numpy as np
import torch

from bde2vid_amd.config import GeneratorConfig
from bde2vid_amd.weights import formula_state_dict
from oracle.gen_golden import golden_inputs, voxel_like, dense_like, voxel_case, E2E_CASES, CFGA_FULL  # noqa: F401

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
