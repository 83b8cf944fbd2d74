#!/usr/bin/env python3
"""End to end on one GPU, the way the reference's eval_model (eval_models_seq.py:147-282) drives its model, with this
build's pieces: a recording (event columns resident in HBM, bde2vid_amd.recording.Recording = the reference's
DynamicH5Dataset) -> one voxel grid per between-frames window (on the device) -> padded to a multiple of 2**num_encoders
-> the model on chunks of subseq_L frames -> cropped frames -> MSE / SSIM against the recording's images (on the device).

The recording here is synthetic; with h5py installed a real file opens with `bde2vid_amd.recording.open_recording(path)`:

    python examples/reconstruct_recording.py [--frames 32] [--height 180] [--width 240] [--checkpoint model.pth]"""
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

    from bde2vid_amd import canonical, metrics
    from bde2vid_amd.harness import reconstruct_sequence
    from bde2vid_amd.recording import Recording
    from bde2vid_amd.synth import synthetic_recording_with_frames

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
    rec = Recording(arrays=synthetic_recording_with_frames(n, H, W, T + 1, seed=1), num_bins=cfg.num_bins,
                    voxel_method={'method': 'between_frames'}, device=device)       # len(rec) == T items

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    grids = rec.voxels(range(len(rec)))                              # [T, num_bins, H, W], one launch
    voxels = [grids[t:t + 1] for t in range(len(rec))]
    frames = reconstruct_sequence(model, voxels, subseq_L=args.subseq)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = torch.cat(frames)                                          # [T, 1, H, W] in (0, 1)
    gts = [torch.from_numpy(rec.get_frame(i)).float()[None, None] / 255 for i in range(len(rec))]
    mean, _ = metrics.score_sequence(frames, gts)
    print(f'{len(rec)} frames of {H}x{W} from {rec.num_events} events in {dt * 1e3:.1f} ms (first call: includes workspace setup); '
          f'output range [{float(out.min()):.3f}, {float(out.max()):.3f}], mean {float(out.mean()):.3f}; against the (random) '
          f'images of the synthetic recording: MSE {mean["mse"]:.4f}, SSIM {mean["ssim"]:.4f}')
    return out


if __name__ == '__main__':
    main()
