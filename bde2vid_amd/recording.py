"""A recording in the Monash HDF5 event format with its event columns resident in HBM: the file side of the data path.

`Recording` is the counterpart of the reference's `DynamicH5Dataset` (data_loader/h5_dataset.py:398-455) on top of
`BaseVoxelDataset` (:45-396) for what `eval_model` uses (eval_models_seq.py:148-165): it exposes the same members
(`sensor_resolution`, `t0`, `tk`, `num_events`, `num_frames`, `frame_ts`, `has_flow`, `duration`, `event_indices`,
`__len__`, `__getitem__`) and the same three voxel methods (`between_frames`, `t_seconds`, `k_events`, :261-317), but

* the four event columns are uploaded ONCE in their native types (int16, int16, float64, bool: 13 B per event, schema
  events_contrast_maximization/tools/event_packagers.py:44-47) and stay on the GPU,
* `find_ts_index` (:444-446 -> event_utils.py:10-28) is the same bisection run on the device for all timestamps at once
  (`bde_find_ts_index`),
* voxel grids come from `bde_voxelize_event_ranges`, many windows per launch, instead of one CPU
  `events_to_voxel_torch` call per item in a DataLoader worker (:204-226,343-366).

The HDF5 container is read through `h5py` when it is installed (`open_recording`); any object with h5py's mapping
interface works (`Recording(file_like)`), which is how the tests exercise the reader in this image (h5py is absent).
On-disk schema: events/{xs,ys,ts,ps}; images/image%09d (uint8) with attrs `timestamp`, `event_idx`; file attrs
`sensor_resolution`, `num_events`, `num_imgs` (event_packagers.py:62-67,98-108).
"""
import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib

DATA_SOURCES = ('esim', 'ijrr', 'mvsec', 'eccd', 'hqfd', 'unknown')      # h5_dataset.py:16


def open_recording(path: str, **kwargs) -> 'Recording':
    """Open an .h5 recording (needs h5py; the reference's loader needs it too, h5_dataset.py:5,417-421)."""
    try:
        import h5py
    except ImportError as e:
        raise RuntimeError('reading .h5 recordings needs h5py, which is not installed; pass an h5py-like object or '
                           'arrays to Recording(...) instead') from e
    return Recording(h5py.File(path, 'r'), **kwargs)


class ItemIndexError(IndexError, AssertionError):
    """An item index outside the recording.  The reference asserts here (h5_dataset.py:210); iteration protocols want an
    IndexError: this is both."""


class Recording:
    def __init__(self, h5_file=None, *, arrays: Optional[dict] = None, sensor_resolution=None, num_bins: int = 5,
                 voxel_method: Optional[dict] = None, max_length: Optional[int] = None, device='cuda'):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('bde2vid_amd.recording keeps the event columns on the GPU; there is no CPU path')
        self.num_bins = num_bins
        self.data_source_idx = -1
        self.frames = None
        self._frame_names: List[str] = []
        self._frame_cache: Dict[int, np.ndarray] = {}
        self._event_idx = None
        if (h5_file is None) == (arrays is None):
            raise ValueError('give either an h5py-like file object or arrays=dict(xs, ys, ts, ps, ...)')
        if h5_file is not None:
            self._load_h5(h5_file, sensor_resolution)
        else:
            self._load_arrays(arrays, sensor_resolution)
        self.duration = self.tk - self.t0                                    # :167
        if voxel_method is None:
            voxel_method = {'method': 'between_frames'}                      # :180-181
        self.set_voxel_method(voxel_method)
        if max_length is not None:
            self.length = min(self.length, max_length + 1)                   # :201-202

    # ---- load_data (h5_dataset.py:417-443) ------------------------------------------------------------------------
    def _load_h5(self, f, sensor_resolution):
        self.h5_file = f
        res = sensor_resolution if sensor_resolution is not None else f.attrs['sensor_resolution']
        self.sensor_resolution = tuple(int(v) for v in res[0:2])
        self.has_flow = 'flow' in f.keys() and len(f['flow']) > 0
        cols = dict(xs=f['events/xs'][:], ys=f['events/ys'][:], ts=f['events/ts'][:], ps=f['events/ps'][:])
        self.num_events = int(f.attrs['num_events'])
        self.num_frames = int(f.attrs['num_imgs'])
        names = list(f['images']) if 'images' in f.keys() else []
        self.frame_ts = [f['images/{}'.format(n)].attrs['timestamp'] for n in names]
        if names and all('event_idx' in f['images/{}'.format(n)].attrs for n in names):
            self._event_idx = [int(f['images/{}'.format(n)].attrs['event_idx']) for n in names]
        self._frame_names = names                  # images are decoded on demand (get_frame), as the reference does (:404-405)
        src = f.attrs.get('source', 'unknown') if hasattr(f.attrs, 'get') else 'unknown'
        self.data_source_idx = DATA_SOURCES.index(src) if src in DATA_SOURCES else -1
        self._upload(cols)

    def _load_arrays(self, a, sensor_resolution):
        self.h5_file = None
        res = sensor_resolution if sensor_resolution is not None else a['sensor_resolution']
        self.sensor_resolution = tuple(int(v) for v in res[0:2])
        self.has_flow = False
        self.num_events = int(a.get('num_events', len(a['ts'])))
        self.frame_ts = [float(t) for t in a.get('frame_ts', [])]
        self.num_frames = int(a.get('num_imgs', len(self.frame_ts)))
        if a.get('event_idx') is not None:
            self._event_idx = [int(v) for v in a['event_idx']]
        if a.get('frames') is not None:
            self.frames = np.asarray(a['frames'])
        self._upload(a)

    def _upload(self, cols):
        def dev(x, dtype):
            t = torch.as_tensor(np.ascontiguousarray(x))
            if t.dtype == torch.bool:
                t = t.to(torch.uint8)
            return t.to(device=self.device, dtype=dtype).contiguous()
        self.xs, self.ys = dev(cols['xs'], torch.int16), dev(cols['ys'], torch.int16)
        self.ts, self.ps = dev(cols['ts'], torch.float64), dev(cols['ps'], torch.uint8)
        n = self.ts.numel()
        if not (self.xs.numel() == self.ys.numel() == self.ps.numel() == n):
            raise ValueError('event columns differ in length')
        if self.num_events > n:
            # a file whose num_events attribute overstates its datasets: an h5py slice would stop at the data, and so do we
            # (event windows are checked against this value, and the binning kernel clamps to the column length as well)
            self.num_events = n
        ts_h = np.asarray(cols['ts'])
        self.t0 = ts_h[0] if n else 0.0                                      # :429-430 (numpy float64, like the reference)
        self.tk = ts_h[-1] if n else 0.0

    # ---- find_ts_index (:444-446) ---------------------------------------------------------------------------------
    def find_ts_index(self, timestamp):
        """Index into the events/ts column for one timestamp (-> int) or a sequence (-> int64 numpy array)."""
        scalar = np.ndim(timestamp) == 0
        q = torch.as_tensor(np.atleast_1d(np.asarray(timestamp, dtype=np.float64))).to(self.device)
        out = torch.empty(q.numel(), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            st = C.c_void_p(int(torch.cuda.current_stream(self.device).cuda_stream))
            _lib.check(_lib.lib().bde_find_ts_index(C.c_void_p(self.ts.data_ptr()), self.ts.numel(), C.c_void_p(q.data_ptr()),
                                                    q.numel(), C.c_void_p(out.data_ptr()), st))
        res = out.cpu().numpy()
        return int(res[0]) if scalar else res

    # ---- event index tables (:261-302, 447-455) -------------------------------------------------------------------
    def compute_frame_indices(self) -> List[List[int]]:
        """[start, end] per frame: the images' `event_idx` attributes when the file has them (DynamicH5Dataset,
        :447-455), else find_ts_index of the frame timestamps (BaseVoxelDataset, :261-275)."""
        ends = self._event_idx if self._event_idx is not None else [int(v) for v in self.find_ts_index(self.frame_ts)]
        out, start = [], 0
        for e in ends:
            out.append([start, int(e)])
            start = int(e)
        return out

    def compute_timeblock_indices(self) -> List[List[int]]:
        vm = self.voxel_method
        end_times = []
        for i in range(len(self)):
            start_time = ((vm['t'] - vm['sliding_window_t']) * i) + self.t0            # :284
            end_times.append(start_time + vm['t'])                                       # :285
        ends = self.find_ts_index(np.asarray(end_times, dtype=np.float64)) if end_times else []
        out, start = [], 0
        for e in ends:
            out.append([start, int(e)])
            start = int(e)
        return out

    def compute_k_indices(self) -> List[List[int]]:
        vm = self.voxel_method
        out = []
        for i in range(len(self)):
            idx0 = (vm['k'] - vm['sliding_window_w']) * i                                # :298
            out.append([idx0, idx0 + vm['k']])
        return out

    def set_voxel_method(self, voxel_method: dict):
        """Select how the event stream is cut into items (:303-317): the item count and the [start, end) table.
        Counts keep the reference's arithmetic: floor of a float quotient, never negative; `between_frames` has one item
        fewer than there are images."""
        self.voxel_method = voxel_method
        kind = voxel_method.get('method')

        def strided_count(total, span_key, overlap_key):
            return max(int(total / (voxel_method[span_key] - voxel_method[overlap_key])), 0)
        planners = {
            'k_events': (lambda: strided_count(self.num_events, 'k', 'sliding_window_w'), self.compute_k_indices),
            't_seconds': (lambda: strided_count(self.duration, 't', 'sliding_window_t'), self.compute_timeblock_indices),
            'between_frames': (lambda: self.num_frames - 1, self.compute_frame_indices),
        }
        if kind not in planners:
            raise ValueError(f'unknown voxel method {kind!r}: expected one of {sorted(planners)}')
        count, table = planners[kind]
        self.length = count()                      # (the index builders iterate over len(self))
        self.event_indices = table()
        if self.length == 0:
            raise ValueError(f'voxel method {voxel_method} yields a sequence length of zero items for this recording '
                             f'({self.num_events} events, {self.num_frames} images, {self.duration} s)')

    def __len__(self):
        return self.length

    def get_event_indices(self, index):
        """[start, end) of item `index` in the event columns; a window that leaves the recording is an error (:329-334)."""
        first, last = self.event_indices[index]
        if first < 0 or last > self.num_events:
            raise IndexError(f'item {index}: event window [{first}, {last}) is out of bounds for a recording of '
                             f'{self.num_events} events')
        return first, last

    # ---- items ----------------------------------------------------------------------------------------------------
    def voxels(self, indices: Sequence[int], check_bounds: bool = True) -> torch.Tensor:
        """Voxel grids of the items `indices` in one launch: float32 [len(indices), num_bins, H, W] on the GPU
        (BaseVoxelDataset.__getitem__ :213-226 + get_voxel_grid :343-366, default all-ones hot-pixel mask)."""
        idx = [self.get_event_indices(int(i)) for i in indices]
        H, W = self.sensor_resolution
        n = len(idx)
        grids = torch.empty((n, self.num_bins, H, W), dtype=torch.float32, device=self.device)
        if n == 0:
            return grids
        se = torch.tensor(idx, dtype=torch.int64).t().contiguous().to(self.device)       # [2][n]
        oob = torch.zeros(1, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            st = C.c_void_p(int(torch.cuda.current_stream(self.device).cuda_stream))
            _lib.check(_lib.lib().bde_voxelize_event_ranges(
                C.c_void_p(self.xs.data_ptr()), C.c_void_p(self.ys.data_ptr()), C.c_void_p(self.ts.data_ptr()),
                C.c_void_p(self.ps.data_ptr()), self.ts.numel(), C.c_void_p(se[0].data_ptr()), C.c_void_p(se[1].data_ptr()), n,
                max(b - a for a, b in idx),
                self.num_bins, H, W, C.c_void_p(grids.data_ptr()), C.c_void_p(oob.data_ptr()), st))
        if check_bounds and int(oob.item()) != 0:
            raise IndexError(f'{int(oob.item())} events fall outside the {H}x{W} sensor '
                             '(the reference index_put_ raises here as well)')
        return grids

    def get_frame(self, index) -> np.ndarray:
        """Image `index` (uint8).  Files are read one image at a time, the way DynamicH5Dataset.get_frame does (:404-405);
        the most recent ones are kept.  The arrays= path holds its frames already."""
        if self.frames is not None:
            return self.frames[index]
        if index not in self._frame_cache:
            if len(self._frame_cache) >= 64:
                self._frame_cache.pop(next(iter(self._frame_cache)))
            self._frame_cache[index] = np.asarray(self.h5_file['images/{}'.format(self._frame_names[index])][:])
        return self._frame_cache[index]

    def item_times(self, index):
        """(ts_0, ts_k, dt) of item `index` as the reference computes them (:214-218,228-231)."""
        idx0, idx1 = self.get_event_indices(index)
        if idx1 - idx0 > 0:
            pair = self.ts[[idx0, idx1 - 1]].cpu().numpy()
            ts_0, ts_k = pair[0], pair[1]
        else:
            ts_0, ts_k = 0, 0
        return ts_0, ts_k, ts_k - ts_0

    def __getitem__(self, index) -> Dict[str, torch.Tensor]:
        if not 0 <= index < len(self):
            raise ItemIndexError(f'item {index} requested from a recording of {len(self)} items')
        voxel = self.voxels([index])[0]
        ts_0, ts_k, dt = self.item_times(index)
        if self.voxel_method['method'] == 'between_frames':
            # uint8 -> float / 255 on the host exactly as transform_frame does (:370; the GPU's division rounds 1 ulp apart)
            frame = (torch.from_numpy(self.get_frame(index)).float().unsqueeze(0) / 255).to(self.device)
            flow = torch.zeros((2, frame.shape[-2], frame.shape[-1]), dtype=frame.dtype, device=self.device)
            return {'frame': frame, 'flow': flow, 'events': voxel,
                    'timestamp': torch.tensor(self.frame_ts[index], dtype=torch.float64),
                    'data_source_idx': self.data_source_idx, 'dt': torch.tensor(dt, dtype=torch.float64)}
        return {'events': voxel, 'timestamp': torch.tensor(ts_k, dtype=torch.float64),
                'data_source_idx': self.data_source_idx, 'dt': torch.tensor(dt, dtype=torch.float64)}
