import sys, os
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from bde2vid_amd import canonical, ops
from bde2vid_amd.model import build_model
from bde2vid_amd.weights import formula_state_dict
cfg = canonical()
m = build_model(cfg, formula_state_dict(cfg), 'cuda:0')
g = torch.Generator(device='cuda').manual_seed(1)
H, W = 14, 14
bufs = [torch.randn(1, 64, H, W, device='cuda', generator=g) for _ in range(3)]
y = ops.dframe_attention(m, 0, bufs, 0, 1)
m.set_tuning('winblock', 0)
r = ops.dframe_attention(m, 0, bufs, 0, 1)
e = (y - r).abs()[0]          # [64, H, W]
print('max err', float(e.max()), 'finite', bool(torch.isfinite(y).all()))
pe = e.amax(dim=0)            # per pixel
print('per-pixel max err (rows = y):')
np.set_printoptions(linewidth=200, precision=2, suppress=True)
print(pe.cpu().numpy())
ce = e.amax(dim=(1, 2))
print('per-channel max err:', ce.cpu().numpy())
