"""TEST INFRASTRUCTURE ONLY -- a minimal in-memory stand-in for `h5py.File` holding a recording in the Monash HDF5
schema (events/{xs,ys,ts,ps}, images/image%09d with attrs timestamp / event_idx, file attrs sensor_resolution /
num_events / num_imgs; events_contrast_maximization/tools/event_packagers.py:44-47,62-67,98-108).  h5py is not installed
in this image: the reference's own DynamicH5Dataset runs on this object in oracle/gen_golden.py (build container), and
the product's `Recording.from_h5` is tested on it on the GPU box.  It implements only what those two readers touch."""
import numpy as np


class _Node:
    def __init__(self, data=None, attrs=None, children=None):
        self.data, self.attrs, self.children = data, dict(attrs or {}), children

    def __getitem__(self, key):
        if self.children is not None:
            if isinstance(key, str):
                node = self
                for part in key.split('/'):
                    node = node.children[part]
                return node
            raise TypeError(key)
        return self.data[key]

    def __len__(self):
        return len(self.children) if self.children is not None else len(self.data)

    def __iter__(self):
        return iter(sorted(self.children))

    def keys(self):
        return self.children.keys()

    @property
    def shape(self):
        return self.data.shape

    @property
    def dtype(self):
        return self.data.dtype


class File(_Node):
    """File(recording_dict): `recording_dict` as returned by bde2vid_amd.synth.synthetic_recording_with_frames."""

    def __init__(self, rec, mode='r'):
        images = {f'image{i:09d}': _Node(rec['frames'][i], dict(timestamp=float(rec['frame_ts'][i]),
                                                                  event_idx=int(rec['event_idx'][i]),
                                                                  size=rec['frames'][i].shape, type='greyscale'))
                  for i in range(len(rec['frame_ts']))}
        events = {k: _Node(np.asarray(rec[k])) for k in ('xs', 'ys', 'ts', 'ps')}
        super().__init__(children={'events': _Node(children=events), 'images': _Node(children=images)},
                         attrs=dict(sensor_resolution=np.asarray(rec['sensor_resolution']), num_events=int(rec['num_events']),
                                    num_imgs=int(rec['num_imgs']), source='unknown'))

    def close(self):
        pass
