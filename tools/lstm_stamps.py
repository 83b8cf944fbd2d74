import sys, os, ctypes as C
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from bde2vid_amd import canonical, ops, _lib
from bde2vid_amd.model import build_model
from bde2vid_amd.weights import formula_state_dict
cfg = canonical()
m = build_model(cfg, formula_state_dict(cfg), 'cuda:0')
L = _lib.lib()
if os.environ.get('PADLDS'): m.set_tuning('lstm_padlds', int(os.environ['PADLDS']))
x = torch.randn(4, 1, 32, 184, 240, device='cuda')
ops.recurrent_conv(m, 0, 0, x); torch.cuda.synchronize()
L.bde_debug_token_stamps(m._h, None, 0)
ops.recurrent_conv(m, 0, 0, x); torch.cuda.synchronize()
out = (C.c_int64 * 2048)()
L.bde_debug_token_stamps(m._h, out, 2048)
a = np.array(out[:], dtype=np.int64).reshape(16, 4, 8, 4)   # block, wave, stage, stamp
bar1 = a[..., 1] - a[..., 0]; store = a[..., 2] - a[..., 1]; load = a[..., 3] - a[..., 2]
mfma = np.roll(a[..., 0], -1, axis=2) - a[..., 3]
print('per stage (cycles), median over blocks/waves, stages 0..7')
print(' barrier wait :', np.median(bar1, axis=(0, 1)))
print(' LDS store+bar:', np.median(store, axis=(0, 1)))
print(' issue loads  :', np.median(load, axis=(0, 1)))
print(' MFMA loop    :', np.median(mfma[:, :, :7], axis=(0, 1)))
print(' whole stage  :', np.median((np.roll(a[..., 0], -1, axis=2) - a[..., 0])[:, :, :7], axis=(0, 1)))
