"""Synthetic event data (numpy only): the input generators of bench.py and of the tests."""
import numpy as np


def synthetic_events(n, height, width, seed):
    """Synthetic event packet of SURVEY.md §8(d): x~U{0..W-1}, y~U{0..H-1} as float32 integers,
    t = sorted U(0,1) float32 minus first, p in {-1,+1}."""
    rng = np.random.default_rng(seed)
    xs = rng.integers(0, width, n).astype(np.float32)
    ys = rng.integers(0, height, n).astype(np.float32)
    ts = np.sort(rng.random(n, dtype=np.float32))
    ts = (ts - ts[0]).astype(np.float32)
    ps = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    return xs, ys, ts, ps


def synthetic_recording(n, height, width, nwin, seed, t_start=1.6e9):
    """Native-typed event columns of a synthetic recording: int16 pixel coordinates, float64 absolute
    timestamps (seconds, epoch-sized so that the float64 -> float32 order of operations matters), bool
    polarity, and nwin+1 window boundaries, one window with fewer than 3 events and one empty."""
    rng = np.random.default_rng(seed)
    xs = rng.integers(0, width, n).astype(np.int16)
    ys = rng.integers(0, height, n).astype(np.int16)
    ts = t_start + np.sort(rng.random(n)) * 0.5
    ps = rng.integers(0, 2, n).astype(bool)
    cuts = np.sort(rng.integers(0, n, nwin - 3))
    idx = np.concatenate([[0], cuts, [n - 2, n - 2, n]]).astype(np.int64)   # ..., empty window, 2-event window
    return xs, ys, ts, ps, idx


def packager_event_idx(ts, frame_ts):
    """The `event_idx` attribute a recording's images carry, as the converter writes it
    (events_contrast_maximization/tools/event_packagers.py:88-108, one buffer: max_buffer_size >= len(ts)):
    searchsorted(ts, image timestamp) - 1 clamped at 0; an image later than every event gets len(ts)."""
    ts = np.asarray(ts)
    out = np.empty(len(frame_ts), dtype=np.int64)
    added, cur = 0, ts
    for k, t in enumerate(frame_ts):
        idx = int(np.searchsorted(cur, t))
        if idx == len(cur):                       # (:100-104) move to the next buffer, which is empty here
            added += len(cur)
            cur = ts[:0]
            idx = int(np.searchsorted(cur, t))
        out[k] = max(0, idx - 1) + added
    return out


def synthetic_recording_with_frames(n, height, width, nframes, seed, t_start=1.6e9, duration=0.5):
    """A synthetic recording in the Monash HDF5 schema (event_packagers.py:44-47,62-67,98-108): native-typed event
    columns with DUPLICATE timestamps (quantised clock), `nframes` uint8 images whose timestamps are a mix of exact event
    timestamps and in-between values, their `event_idx` attributes, and the file attributes `DynamicH5Dataset.load_data`
    reads (data_loader/h5_dataset.py:417-443)."""
    rng = np.random.default_rng(seed)
    xs = rng.integers(0, width, n).astype(np.int16)
    ys = rng.integers(0, height, n).astype(np.int16)
    ts = t_start + np.sort(np.round(rng.random(n) * duration, 4))
    ps = rng.integers(0, 2, n).astype(bool)
    ft = np.sort(rng.random(nframes)) * duration * 0.98 + 0.01 * duration + t_start
    hit = rng.integers(0, n, nframes)
    exact = rng.random(nframes) < 0.4
    ft[exact] = ts[hit[exact]]                    # timestamps that ARE event timestamps (the bisection's == branch)
    ft = np.sort(ft)
    frames = rng.integers(0, 256, (nframes, height, width)).astype(np.uint8)
    return dict(xs=xs, ys=ys, ts=ts, ps=ps, frame_ts=ft, frames=frames, event_idx=packager_event_idx(ts, ft),
                sensor_resolution=np.array([height, width]), num_events=n, num_imgs=nframes)
