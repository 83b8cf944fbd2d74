"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's event-index logic for recordings.

Restates `binary_search_h5_dset` (events_contrast_maximization/utils/event_utils.py:10-28), which is
`DynamicH5Dataset.find_ts_index` (data_loader/h5_dataset.py:444-446), and the three index tables of
`BaseVoxelDataset.set_voxel_method` (:303-321) with `compute_frame_indices` (:261-275, and the H5 override :447-455),
`compute_timeblock_indices` (:277-290) and `compute_k_indices` (:292-302).  Pinned to tests/golden/rec_dataset.npz, which
holds what the reference's own DynamicH5Dataset returned.  The product never imports this module."""
import numpy as np


def find_ts_index(ts, x):
    """event_utils.py:10-28 with side='left': the bisection's own landing index on an exact hit, else the insertion point."""
    l, r = 0, len(ts) - 1
    while l <= r:
        mid = l + (r - l) // 2
        v = ts[mid]
        if v == x:
            return mid
        if v < x:
            l = mid + 1
        else:
            r = mid - 1
    return l


def chain(ends):
    out, start = [], 0
    for e in ends:
        out.append([start, int(e)])
        start = int(e)
    return out


def frame_indices_from_attrs(event_idx):                      # h5_dataset.py:447-455
    return chain(event_idx)


def frame_indices_from_timestamps(ts, frame_ts):              # h5_dataset.py:261-275
    return chain(find_ts_index(ts, t) for t in frame_ts)


def dataset_length(method, num_events, num_frames, duration):  # h5_dataset.py:308-321
    m = method['method']
    if m == 'k_events':
        return max(int(num_events / (method['k'] - method['sliding_window_w'])), 0)
    if m == 't_seconds':
        return max(int(duration / (method['t'] - method['sliding_window_t'])), 0)
    return num_frames - 1


def timeblock_indices(ts, method, length):                    # h5_dataset.py:277-290
    t0 = ts[0]
    ends = []
    for i in range(length):
        start_time = ((method['t'] - method['sliding_window_t']) * i) + t0
        ends.append(find_ts_index(ts, start_time + method['t']))
    return chain(ends)


def k_indices(method, length):                                # h5_dataset.py:292-302
    return [[(method['k'] - method['sliding_window_w']) * i, (method['k'] - method['sliding_window_w']) * i + method['k']]
            for i in range(length)]
