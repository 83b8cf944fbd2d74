"""Phase timeline (s_memtime) of the fused window-block kernel at level 0 of config A (GPU box)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from bde2vid_amd import canonical, ops, _lib
from bde2vid_amd.model import build_model
from bde2vid_amd.weights import formula_state_dict
cfg = canonical()
m = build_model(cfg, formula_state_dict(cfg), 'cuda:0')
L = _lib.lib()
bufs = [torch.randn(1, 64, 92, 120, device='cuda') for _ in range(3)]
for _ in range(3): ops.dframe_attention(m, 0, bufs)
torch.cuda.synchronize()
L.bde_debug_token_stamps(m._h, None, 0)
ops.dframe_attention(m, 0, bufs, 0, 1)
torch.cuda.synchronize()
out = (C.c_int64 * 2048)()
L.bde_debug_token_stamps(m._h, out, 2048)
a = np.array(out[:], dtype=np.int64).reshape(64, 4, 8)
d = a[:, :, 1:7] - a[:, :, 0:6]
print('phase cycles (gather, qkv, attention, proj, fc1, fc2) median over waves:', np.median(d.reshape(-1, 6), axis=0))
print('block 0 wave 0:', d[0, 0], ' total', a[0, 0, 6] - a[0, 0, 0])
print('start skew across blocks (cycles):', (a[:, 0, 0] - a[:, 0, 0].min())[:16])
