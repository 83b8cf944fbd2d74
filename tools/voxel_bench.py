"""Event binning: the three kernels (bde_voxel_method 3 streaming, 2 bucketed, 1 scatter) on one large recording, HIP-event time per
call; run under rocprofv3 --kernel-trace --stats for the per-kernel split.   GPU box:  python tools/voxel_bench.py [n_events] [nwin]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bde2vid_amd import _lib
from bde2vid_amd.events import events_to_voxel_windows
from bde2vid_amd.synth import synthetic_recording
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24_000_000
nwin = int(sys.argv[2]) if len(sys.argv) > 2 else 64
L = _lib.lib()
for (H, W) in ((180, 240), (480, 640), (720, 1280)):
    xs, ys, ts, ps, _ = synthetic_recording(n, H, W, 4, 77)
    idx = torch.from_numpy(np.arange(nwin + 1, dtype=np.int64) * (n // nwin))
    cols = [torch.from_numpy(a).cuda() for a in (xs, ys, ts, ps)]
    for method, name in ((3, 'streaming'), (2, 'bucketed'), (1, 'scatter')):
        _lib.check(L.bde_voxel_method(method))
        g = events_to_voxel_windows(*cols, idx, 5, sensor_size=(H, W), check_bounds=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            events_to_voxel_windows(*cols, idx, 5, sensor_size=(H, W), check_bounds=False)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        alg = n * 13 + nwin * 5 * H * W * 4
        print(f'{H}x{W} {name:9s}: {ms:.3f} ms per call, {n / ms / 1e6:.1f} G events/s, {alg / ms / 1e6:.0f} GB/s algorithmic')
    del cols
_lib.check(L.bde_voxel_method(0))
