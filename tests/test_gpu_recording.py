"""GPU parity of the file side of the data path (bde2vid_amd/recording.py, bde_find_ts_index, bde_voxelize_event_ranges)
against what the reference's own DynamicH5Dataset returned for the same synthetic recording (tests/golden/rec_dataset.npz,
oracle/gen_golden.py::gen_recording_dataset).  Index tables are exact; voxel grids differ only by the float summation order."""
import json
import os

import numpy as np
import pytest
import torch

from tests.util import GOLDEN, maxabs
from bde2vid_amd.synth import synthetic_recording_with_frames

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def golden():
    z = np.load(os.path.join(GOLDEN, 'rec_dataset.npz'))
    meta = json.loads(str(z['meta']))
    return z, meta, synthetic_recording_with_frames(**meta['recording'])


@pytest.mark.parametrize('method', ['between_frames', 't_seconds', 'k_events'])
def test_recording_matches_reference_dataset(golden, method):
    from bde2vid_amd.recording import Recording
    from oracle import fake_h5                         # an h5py.File stand-in (h5py is not installed in this image)
    z, meta, rec = golden
    ds = Recording(fake_h5.File(rec), num_bins=5, voxel_method=dict(meta['methods'][method]))
    assert len(ds) == int(z[method + '_len'])
    assert np.asarray(ds.event_indices).tolist() == z[method + '_indices'].tolist()
    assert tuple(ds.sensor_resolution) == (36, 48) and ds.num_events == 6000 and ds.num_frames == 12
    n = z[method + '_events'].shape[0]
    grids = ds.voxels(range(n))
    assert maxabs(grids, z[method + '_events']) <= 1e-4
    for i in range(n):
        it = ds[i]
        assert maxabs(it['events'], z[method + '_events'][i]) <= 1e-4
        assert float(it['dt']) == float(z[method + '_dt'][i])
        assert float(it['timestamp']) == float(z[method + '_timestamp'][i])
        if method == 'between_frames':
            assert torch.equal(it['frame'].cpu(), torch.from_numpy(z['between_frames_frame'][i]))
            assert it['flow'].shape == (2, 36, 48) and float(it['flow'].abs().max()) == 0.0


def test_find_ts_index_is_the_reference_bisection(golden):
    from bde2vid_amd.recording import Recording
    z, meta, rec = golden
    arrays = {k: v for k, v in rec.items() if k != 'event_idx'}      # a file without event_idx attributes
    ds = Recording(arrays=arrays)
    assert ds.find_ts_index(z['probes']).tolist() == z['find_ts_index'].tolist()
    assert ds.find_ts_index(float(z['probes'][3])) == int(z['find_ts_index'][3])
    # BaseVoxelDataset.compute_frame_indices (h5_dataset.py:261-275): frame windows from the timestamps alone
    assert np.asarray(ds.event_indices).tolist() == z['base_frame_indices'].tolist()


def test_out_of_range_item_raises_like_the_reference(golden):
    from bde2vid_amd.recording import Recording
    z, meta, rec = golden
    ds = Recording(arrays=rec, voxel_method={'method': 'k_events', 'k': 700, 'sliding_window_w': 200})
    last = len(ds) - 1
    assert ds.event_indices[last][1] > ds.num_events        # the reference's table runs past the recording too
    with pytest.raises(Exception, match='out of bounds'):
        ds[last]
    with pytest.raises(Exception, match='length of zero'):
        Recording(arrays=rec, voxel_method={'method': 'k_events', 'k': 10 ** 7, 'sliding_window_w': 0})
    with pytest.raises(AssertionError):                      # the reference asserts (h5_dataset.py:210) ...
        ds[len(ds)]
    with pytest.raises(IndexError):                          # ... and Python's iteration protocol wants an IndexError
        ds[-1]
    with pytest.raises(ValueError, match='unknown voxel method'):
        Recording(arrays=rec, voxel_method={'method': 'every_other_tuesday'})


def test_attributes_that_overstate_the_columns_cannot_reach_past_them(golden):
    """A truncated / inconsistent file: `num_events` and the images' `event_idx` claim more events than the datasets hold
    (ADVICE r2).  num_events is clamped to the columns, the host check rejects windows past them, and a window handed to the
    binning entry point directly is clamped on the device instead of read out of bounds."""
    import ctypes as C
    from bde2vid_amd import _lib
    from bde2vid_amd.recording import Recording
    z, meta, rec = golden
    lying = dict(rec)
    lying['num_events'] = int(rec['num_events']) + 5000
    # (between_frames: item i is the window that ends at image i's event_idx; there are num_imgs - 1 items)
    lying['event_idx'] = [int(v) for v in rec['event_idx'][:-2]] + [int(rec['num_events']) + 4000] * 2
    ds = Recording(arrays=lying)
    assert ds.num_events == len(rec['ts'])
    with pytest.raises(IndexError, match='out of bounds'):
        ds.voxels([len(ds) - 1])
    with pytest.raises(IndexError, match='out of bounds'):
        ds[len(ds) - 1]
    ok = ds.voxels(range(len(ds) - 1))
    ref = Recording(arrays=rec).voxels(range(len(ds) - 1))
    assert maxabs(ok, ref) <= 1e-5                          # (LDS atomics: the summation order inside a pixel varies)
    # straight through the C ABI: [n - 100, n + 10^6) must behave as [n - 100, n)
    n = ds.ts.numel()
    H, W = ds.sensor_resolution

    def run(end):
        se = torch.tensor([[n - 100], [end]], dtype=torch.int64, device='cuda')
        g = torch.empty((1, 5, H, W), dtype=torch.float32, device='cuda')
        _lib.check(_lib.lib().bde_voxelize_event_ranges(
            C.c_void_p(ds.xs.data_ptr()), C.c_void_p(ds.ys.data_ptr()), C.c_void_p(ds.ts.data_ptr()), C.c_void_p(ds.ps.data_ptr()),
            n, C.c_void_p(se[0].data_ptr()), C.c_void_p(se[1].data_ptr()), 1, 0, 5, H, W, C.c_void_p(g.data_ptr()), None, None))
        torch.cuda.synchronize()
        return g
    assert maxabs(run(n + 10 ** 6), run(n)) <= 1e-5


@pytest.mark.parametrize('sensor', [(180, 240), (480, 640), (720, 1280)])
def test_bucketed_binning_equals_streaming_and_scatter(sensor):
    """The three binning kernels on the same recording (native columns and float columns): windows with 0 and 2 events, a
    window that is one partial chunk, windows of many chunks; 8 / 50 / 150 pixel tiles per grid."""
    from bde2vid_amd import _lib
    from bde2vid_amd.events import events_to_voxel_windows, events_to_voxel_batch
    from bde2vid_amd.synth import synthetic_recording
    H, W = sensor
    xs, ys, ts, ps, idx = synthetic_recording(400000, H, W, 9, 11)
    L = _lib.lib()
    out, outf = {}, {}
    off = [int(v) for v in idx]
    fcols = (xs.astype(np.float32), ys.astype(np.float32), (ts - ts[0]).astype(np.float32), np.where(ps, 0.75, -1.5).astype(np.float32))
    try:
        for method in (3, 2, 1):
            _lib.check(L.bde_voxel_method(method))
            out[method] = events_to_voxel_windows(xs, ys, ts, ps, idx, 5, sensor_size=(H, W))
            outf[method] = events_to_voxel_batch(*fcols, off, 5, sensor_size=(H, W))       # arbitrary per-event weights
    finally:
        _lib.check(L.bde_voxel_method(0))
    for o in (out, outf):
        assert maxabs(o[2], o[3]) <= 1e-4 and maxabs(o[2], o[1]) <= 1e-4
        assert float(o[2][-1].abs().max()) == 0.0 or o is outf          # empty window (native: < 3 events give a zero grid)
        assert float(o[2].abs().sum()) > 0
    assert float(out[2][-2].abs().max()) == 0.0                          # the 2-event window of a recording is a zero grid


def test_zero_duration_window_is_nan_on_every_kernel():
    """dt == 0 makes every weight of the window NaN in the reference (event_utils.py:489-495): pixels that received an event
    are NaN in all bins, the others stay 0 -- on the streaming kernel, the bucketed one (fixed-point tile: a flag, not
    arithmetic) and the scatter."""
    from bde2vid_amd import _lib
    from bde2vid_amd.events import events_to_voxel_windows
    from bde2vid_amd.synth import synthetic_recording
    xs, ys, ts, ps, _ = synthetic_recording(90000, 180, 240, 4, 13)
    idx = np.array([0, 30000, 60000, 90000], dtype=np.int64)
    ts = ts.copy()
    ts[30000:60000] = ts[30000]                                # the middle window has zero duration
    L = _lib.lib()
    out = {}
    try:
        for method in (3, 2, 1):
            _lib.check(L.bde_voxel_method(method))
            out[method] = events_to_voxel_windows(xs, ys, ts, ps, idx, 5, sensor_size=(180, 240)).cpu()
    finally:
        _lib.check(L.bde_voxel_method(0))
    hit = torch.zeros(180, 240, dtype=torch.bool)
    hit[torch.from_numpy(ys[30000:60000].astype(np.int64)), torch.from_numpy(xs[30000:60000].astype(np.int64))] = True
    for method, g in out.items():
        assert torch.equal(torch.isnan(g[1]), hit[None].expand(5, -1, -1)), method
        assert float(g[1][~torch.isnan(g[1])].abs().max()) == 0.0
        assert not torch.isnan(g[0]).any() and not torch.isnan(g[2]).any()
        assert maxabs(g[0], out[3][0]) <= 1e-4 and maxabs(g[2], out[3][2]) <= 1e-4


def test_tile_binning_equals_atomic_scatter():
    """The tile-privatised kernel (default) against the global-atomic scatter on overlapping-free windows of a larger
    recording, HD-sized sensor (several column tiles), windows with 0 and 2 events included."""
    from bde2vid_amd import _lib
    from bde2vid_amd.events import events_to_voxel_windows
    from bde2vid_amd.synth import synthetic_recording
    xs, ys, ts, ps, idx = synthetic_recording(300000, 720, 1280, 9, 5)
    L = _lib.lib()
    a = events_to_voxel_windows(xs, ys, ts, ps, idx, 5, sensor_size=(720, 1280))
    _lib.check(L.bde_voxel_method(3))
    a = events_to_voxel_windows(xs, ys, ts, ps, idx, 5, sensor_size=(720, 1280))
    _lib.check(L.bde_voxel_method(1))
    try:
        b = events_to_voxel_windows(xs, ys, ts, ps, idx, 5, sensor_size=(720, 1280))
    finally:
        _lib.check(L.bde_voxel_method(0))
    assert maxabs(a, b) <= 1e-4
    assert float(a[-1].abs().max()) == 0.0 and float(a[-2].abs().max()) == 0.0      # 2-event and empty windows
    assert float(a.abs().sum()) > 0


def test_bucketed_binning_from_two_threads_on_two_streams():
    """The scratch of the bucketed binning (records + run tables) is stream-ordered (hipMallocAsync on the caller's stream in
    front of the call's two launches, hipFreeAsync behind them): two host threads binning different recordings on two streams
    at the same time get the grids each of them gets alone -- with one shared buffer per device they overwrote each other's
    records.  A NaN time stamp inside an ordinary window marks its pixel NaN in every bin on the fixed-point tile path too."""
    import threading
    from bde2vid_amd import _lib
    from bde2vid_amd.events import events_to_voxel_windows
    from bde2vid_amd.synth import synthetic_recording
    H, W = 180, 240
    recs = [synthetic_recording(600000 + 50000 * i, H, W, 8, 21 + i) for i in range(2)]
    L = _lib.lib()
    _lib.check(L.bde_voxel_method(2))                      # bucketed always
    try:
        ref = [events_to_voxel_windows(*r[:4], r[4], 5, sensor_size=(H, W)).clone() for r in recs]
        torch.cuda.synchronize()
        out = [[None] * 6 for _ in recs]
        errs = []

        def work(i):
            try:
                st = torch.cuda.Stream()
                with torch.cuda.stream(st):
                    for k in range(6):
                        out[i][k] = events_to_voxel_windows(*recs[i][:4], recs[i][4], 5, sensor_size=(H, W))
                st.synchronize()
            except Exception as e:                          # noqa: BLE001 (reported below)
                errs.append(e)
        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for i in range(2):
            for k in range(6):
                assert torch.equal(out[i][k], ref[i]), (i, k)
        # a NaN time stamp in window 2 of recording 0
        xs, ys, ts, ps, idx = recs[0]
        ts = ts.copy()
        j = int(idx[2]) + 1234
        ts[j] = np.nan
        g = events_to_voxel_windows(xs, ys, ts, ps, idx, 5, sensor_size=(H, W))
        y, x = int(ys[j]), int(xs[j])
        assert torch.isnan(g[2, :, y, x]).all() and int(torch.isnan(g).sum()) == 5
    finally:
        _lib.check(L.bde_voxel_method(0))
