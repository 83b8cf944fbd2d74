"""Event -> voxel-grid binning on the GPU (mirror of the reference's data-path helper).

`events_to_voxel_torch(xs, ys, ts, ps, B, device=None, sensor_size=(180, 240), temporal_bilinear=True)`
has the signature and argument meaning of the reference function of the same name
(events_contrast_maximization/utils/event_utils.py:466-509, called from
data_loader/h5_dataset.py:357).  The work happens in libbde2vid.so (bde_voxelize).
"""
import ctypes as C
from typing import Sequence

import torch

from . import _lib


def _dev_f32(t, device):
    return torch.as_tensor(t).to(device=device, dtype=torch.float32).contiguous()


def events_to_voxel_torch(xs, ys, ts, ps, B, device=None, sensor_size=(180, 240), temporal_bilinear=True,
                          check_bounds=True):
    if not temporal_bilinear:
        raise NotImplementedError('temporal_bilinear=False is not on the eval path (and is broken in the reference)')
    if device is None:
        device = xs.device if isinstance(xs, torch.Tensor) and xs.is_cuda else torch.device('cuda')
    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError('bde2vid_amd.events runs on the GPU only')
    xs, ys, ts, ps = (_dev_f32(a, device) for a in (xs, ys, ts, ps))
    assert len(xs) == len(ys) == len(ts) == len(ps)      # event_utils.py:487
    H, W = sensor_size
    grid = torch.empty((B, H, W), dtype=torch.float32, device=device)
    oob = torch.zeros(1, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        st = C.c_void_p(int(torch.cuda.current_stream(device).cuda_stream))
        _lib.check(_lib.lib().bde_voxelize(C.c_void_p(xs.data_ptr()), C.c_void_p(ys.data_ptr()),
                                           C.c_void_p(ts.data_ptr()), C.c_void_p(ps.data_ptr()), xs.numel(),
                                           B, H, W, C.c_void_p(grid.data_ptr()), C.c_void_p(oob.data_ptr()), st))
    if check_bounds and int(oob.item()) != 0:
        raise IndexError(f'{int(oob.item())} events fall outside the {H}x{W} sensor '
                         '(the reference index_put_ raises here as well)')
    return grid


def events_to_voxel_batch(xs, ys, ts, ps, offsets: Sequence[int], B, sensor_size=(180, 240), device=None,
                          check_bounds=True):
    """Many event packets in one launch.  offsets: nseg+1 boundaries into the concatenated arrays."""
    device = torch.device(device) if device is not None else torch.device('cuda')
    xs, ys, ts, ps = (_dev_f32(a, device) for a in (xs, ys, ts, ps))
    off = torch.as_tensor(offsets, dtype=torch.int64)
    nseg = off.numel() - 1
    max_n = int((off[1:] - off[:-1]).max()) if nseg > 0 else 0
    off_d = off.to(device)
    H, W = sensor_size
    grids = torch.empty((nseg, B, H, W), dtype=torch.float32, device=device)
    oob = torch.zeros(1, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        st = C.c_void_p(int(torch.cuda.current_stream(device).cuda_stream))
        _lib.check(_lib.lib().bde_voxelize_batch(C.c_void_p(xs.data_ptr()), C.c_void_p(ys.data_ptr()),
                                                 C.c_void_p(ts.data_ptr()), C.c_void_p(ps.data_ptr()),
                                                 C.c_void_p(off_d.data_ptr()), nseg, max_n, B, H, W,
                                                 C.c_void_p(grids.data_ptr()), C.c_void_p(oob.data_ptr()), st))
    if check_bounds and int(oob.item()) != 0:
        raise IndexError(f'{int(oob.item())} events fall outside the {H}x{W} sensor')
    return grids


def events_to_voxel_windows(xs, ys, ts, ps, event_idx: Sequence[int], B, sensor_size=(180, 240), device=None,
                            check_bounds=True):
    """Voxel grids of consecutive between-frames windows straight from a recording's native event columns.

    The data path of `DynamicH5Dataset` + `BaseVoxelDataset.__getitem__` (data_loader/h5_dataset.py:204-259,
    343-366, 410-415) for `voxel_method = between_frames`, without the per-item host work:
      xs, ys : int16 [N]   (`events/xs`, `events/ys`)          ts : float64 [N] seconds (`events/ts`)
      ps     : bool / uint8 [N] (`events/ps`)                  event_idx : nwin+1 boundaries into the columns
                                                               (the frames' `event_idx` attributes, :262-275)
    Window w covers events [event_idx[w], event_idx[w+1]); fewer than 3 events give a zero grid (:219-220).
    Returns float32 [nwin, B, H, W] on the GPU (numpy or torch inputs, host or device)."""
    device = torch.device(device) if device is not None else torch.device('cuda')
    if device.type != 'cuda':
        raise RuntimeError('bde2vid_amd.events runs on the GPU only')

    def col(a, dtype):
        t = torch.as_tensor(a)
        if t.dtype == torch.bool:
            t = t.to(torch.uint8)
        return t.to(device=device, dtype=dtype).contiguous()

    xs_d, ys_d = col(xs, torch.int16), col(ys, torch.int16)
    ts_d, ps_d = col(ts, torch.float64), col(ps, torch.uint8)
    assert xs_d.numel() == ys_d.numel() == ts_d.numel() == ps_d.numel()
    off = torch.as_tensor(event_idx, dtype=torch.int64)
    nwin = off.numel() - 1
    if nwin < 1:
        raise ValueError('event_idx needs at least two boundaries')
    if int(off.min()) < 0 or int(off.max()) > xs_d.numel() or bool((off[1:] < off[:-1]).any()):
        raise IndexError('event_idx must be non-decreasing and inside the event columns')
    max_n = int((off[1:] - off[:-1]).max())
    off_d = off.to(device)
    H, W = sensor_size
    grids = torch.empty((nwin, B, H, W), dtype=torch.float32, device=device)
    oob = torch.zeros(1, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        st = C.c_void_p(int(torch.cuda.current_stream(device).cuda_stream))
        _lib.check(_lib.lib().bde_voxelize_events(C.c_void_p(xs_d.data_ptr()), C.c_void_p(ys_d.data_ptr()),
                                                  C.c_void_p(ts_d.data_ptr()), C.c_void_p(ps_d.data_ptr()),
                                                  C.c_void_p(off_d.data_ptr()), nwin, max_n, B, H, W,
                                                  C.c_void_p(grids.data_ptr()), C.c_void_p(oob.data_ptr()), st))
    if check_bounds and int(oob.item()) != 0:
        raise IndexError(f'{int(oob.item())} events fall outside the {H}x{W} sensor '
                         '(the reference index_put_ raises here as well)')
    return grids
