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
