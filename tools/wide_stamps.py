"""s_memtime phase timeline of the level-2 chain's kernels (tok_debug 21: mlp_fused_kernel, 22: attn core).  GPU box.
The counter ticks at 100 MHz x ... (s_memtime: shader clock); differences in cycles, converted with the measured clock."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from bde2vid_amd import canonical, ops, _lib
from bde2vid_amd.model import build_model
from bde2vid_amd.weights import formula_state_dict
which = int(sys.argv[1]) if len(sys.argv) > 1 else 21
cfg = canonical()
m = build_model(cfg, formula_state_dict(cfg), 'cuda:0')
L = _lib.lib()
bufs = [torch.randn(1, 256, 23, 30, device='cuda') for _ in range(3)]
for _ in range(3):
    ops.dframe_attention(m, 2, bufs)
torch.cuda.synchronize()
m.set_tuning('tok_debug', which)
L.bde_debug_token_stamps(m._h, None, 0)
ops.dframe_attention(m, 2, bufs, 2, 1)          # one block (block 2)
torch.cuda.synchronize()
out = (C.c_int64 * 2048)()
L.bde_debug_token_stamps(m._h, out, 2048)
a = np.array(out[:], dtype=np.int64).reshape(64, 4, 8)
names = {21: ['early loads issued .. proj + x1 + LN', 'fc1 + GELU', 'fc2', 'partial stores drained', 'barrier + counter add', 'last arriver: loads + sum + store'],
         22: ['loads issued + weights landed', 'q|k|v GEMM', 'K/V staged', 'scores', 'softmax + p.v', 'store']}[which]
n = len(names)
valid = a[:, :, 0] > 0
d = (a[:, :, 1:n + 1] - a[:, :, 0:n]).astype(np.float64)
d[~valid] = np.nan
for i, nm in enumerate(names):
    col = d[:, :, i]
    col = col[(a[:, :, i + 1] > 0) & valid]
    if col.size:
        print(f'{nm:45s} median {np.median(col):8.0f} cycles   min {col.min():8.0f}  max {col.max():8.0f}   ({col.size} waves)')
tot = (a[:, :, 5 if which == 21 else n] - a[:, :, 0])[valid]
print('start .. common end: median', np.median(tot), 'cycles; start skew across workgroups', int((a[:, 0, 0][valid[:, 0]]).max() - (a[:, 0, 0][valid[:, 0]]).min()))
