#!/usr/bin/env python3
"""End to end on one GPU, the way the reference's eval_models_seq.py drives its model, with this build's
pieces: a recording's native event columns -> one voxel grid per between-frames window (on the device) ->
padded to a multiple of 2**num_encoders -> the model on chunks of subseq_L frames -> cropped frames.

The recording here is synthetic (the HDF5 container needs h5py, which is not part of this build):

    python examples/reconstruct_recording.py [--frames 32] [--height 180] [--width 240] [--checkpoint model.pth]

With a real file the four columns are `f['events/xs'][:]`, `ys`, `ts`, `ps` and the window boundaries the
frames' `event_idx` attributes (data_loader/h5_dataset.py:262-275,410-415 in the reference)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np   # noqa: E402
import torch         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=32)
    ap.add_argument('--height', type=int, default=180)
    ap.add_argument('--width', type=int, default=240)
    ap.add_argument('--subseq', type=int, default=16, help='frames per model call (eval_models_seq.py: subseq_L)')
    ap.add_argument('--checkpoint', default=None, help="a reference checkpoint {'state_dict','meta':{'cfg'}}")
    args = ap.parse_args()

    from bde2vid_amd import canonical
    from bde2vid_amd.events import events_to_voxel_windows
    from bde2vid_amd.harness import reconstruct_sequence
    from bde2vid_amd.synth import synthetic_recording

    device = torch.device('cuda:0')
    if args.checkpoint:
        from bde2vid_amd.checkpoint import load_model
        model = load_model(args.checkpoint, device)
        cfg = model.cfg
    else:
        from bde2vid_amd.model import build_model
        from bde2vid_amd.weights import formula_state_dict
        cfg = canonical()
        model = build_model(cfg, formula_state_dict(cfg), device)   # random-init weights of the canonical architecture

    H, W, T = args.height, args.width, args.frames
    n = T * (H * W // 2)
    xs, ys, ts, ps, _ = synthetic_recording(n, H, W, 4, seed=1)
    event_idx = np.arange(T + 1, dtype=np.int64) * (n // T)          # one window per frame

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    grids = events_to_voxel_windows(xs, ys, ts, ps, event_idx, cfg.num_bins, sensor_size=(H, W), device=device)
    voxels = [grids[t:t + 1] for t in range(T)]                      # T tensors [1, num_bins, H, W]
    frames = reconstruct_sequence(model, voxels, subseq_L=args.subseq)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = torch.cat(frames)                                          # [T, 1, H, W] in (0, 1)
    print(f'{T} frames of {H}x{W} from {n} events in {dt * 1e3:.1f} ms (first call: includes workspace setup); '
          f'output range [{float(out.min()):.3f}, {float(out.max()):.3f}], mean {float(out.mean()):.3f}')
    return out


if __name__ == '__main__':
    main()
