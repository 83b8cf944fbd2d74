import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from bde2vid_amd import canonical, ops
from bde2vid_amd.model import build_model
from bde2vid_amd.weights import formula_state_dict
cfg = canonical()
m = build_model(cfg, formula_state_dict(cfg), 'cuda:0')
g = torch.Generator(device='cuda').manual_seed(1)
H, W = 7, 7
bufs = [torch.randn(1, 64, H, W, device='cuda', generator=g) for _ in range(3)]
y = ops.dframe_attention(m, 0, bufs, 0, 1)
m.set_tuning('winblock', 0)
r = ops.dframe_attention(m, 0, bufs, 0, 1)
e = (y - r).abs()[0]
np.set_printoptions(linewidth=220, precision=3, suppress=True)
print('max err', float(e.max()))
print('per-pixel max err:'); print(e.amax(dim=0).cpu().numpy())
for (py, px) in [(6, 6), (0, 0), (2, 2)]:
    print('pixel', py, px, 'err per channel:', e[:, py, px].cpu().numpy())
    print('   y  ', y[0, :8, py, px].cpu().numpy())
    print('   ref', r[0, :8, py, px].cpu().numpy())
