"""Level-2 attention chain of config A (23x30 map, 256 ch): HIP-event time per launch of every kernel of the chain, for each
setting of the chain's switches, and the difference of each setting's result from the all-fp32 split path.
GPU box:  python tools/wide_bench.py [H W]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bde2vid_amd import canonical, ops, _lib
from bde2vid_amd.model import build_model
from bde2vid_amd.weights import formula_state_dict
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (23, 30)
cfg = canonical()
m = build_model(cfg, formula_state_dict(cfg), 'cuda:0')
L = _lib.lib()
g = torch.Generator(device='cuda').manual_seed(1)
bufs = [torch.randn(1, 256, H, W, device='cuda', generator=g) for _ in range(3)]
for _ in range(3):
    y = ops.dframe_attention(m, 2, bufs)
torch.cuda.synchronize()
m.set_tuning('wide', 0)
y_old = ops.dframe_attention(m, 2, bufs)
m.set_tuning('wide', 1)
print('max |wide - split path| over 6 blocks:', float((y - y_old).abs().max()))
SETTINGS = [dict(wide_fuse_fc2=0, wide_core2=0, wide_spl=0), dict(wide_fuse_fc2=1, wide_core2=1, wide_spl=0), dict(wide_fuse_fc2=1, wide_core2=1, wide_spl=1)]
for st in SETTINGS:
    for k, v in st.items():
        m.set_tuning(k, v)
    yf = ops.dframe_attention(m, 2, bufs)
    yf2 = ops.dframe_attention(m, 2, bufs)
    print(f'{st}: max |y - split path| {float((yf - y_old).abs().max()):.3e}  repeatable {bool(torch.equal(yf, yf2))}')
    L.bde_profile_reset(m._h, 1)
    for _ in range(20):
        ops.dframe_attention(m, 2, bufs)
    torch.cuda.synchronize()
    buf = C.create_string_buffer(4096)
    L.bde_profile_names(m._h, buf, len(buf))
    tot = 0.0
    for nm in buf.value.decode().split():
        ms, cnt = C.c_double(), C.c_int64()
        L.bde_profile_get(m._h, nm.encode(), C.byref(ms), C.byref(cnt))
        per_frame = ms.value / 20 * 1e3
        tot += per_frame
        print(f'  {nm:16s} {cnt.value // 20:3d} launches/frame  {ms.value / cnt.value * 1e3:7.2f} us each  {per_frame:8.1f} us/frame')
    print(f'  sum of spans per frame: {tot:.1f} us (eager, HIP events incl. their overhead)')
    L.bde_profile_reset(m._h, 0)
